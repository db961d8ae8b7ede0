!> Minimal stand-in for the HOST MODEL's `prognostics` module (the part time_stepping.f90 uses), to build and test the drop-in
!  inside this repository.  In a real integration it is the model's own (source/prognostics.f90) and this file is not compiled.
module prognostics
    use types, only: p
    use params
    implicit none
    complex(p) :: vor(mx,nx,kx,2), div(mx,nx,kx,2), t(mx,nx,kx,2), ps(mx,nx,2), tr(mx,nx,kx,2,ntr)
    complex(p) :: phi(mx,nx,kx), phis(mx,nx)
end module
