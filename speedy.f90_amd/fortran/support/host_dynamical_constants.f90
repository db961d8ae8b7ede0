!> Minimal stand-in for the HOST MODEL's `dynamical_constants` module (the part time_stepping.f90 uses), to build and test the drop-in
!  inside this repository.  In a real integration it is the model's own (source/dynamical_constants.f90) and this file is not compiled.
module dynamical_constants
    use types, only: p
    implicit none
    real(p), parameter :: tdrs = 24.0*30.0    ! damping time (hours) of the zonal-mean wind drag in the stratosphere
end module
