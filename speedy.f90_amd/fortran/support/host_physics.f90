!> Stand-in for the HOST MODEL's `physics` module, to build and test the physics hook of the time_stepping drop-in inside
!  this repository (-DSPDY_WITH_PHYSICS).  Same interface as the model's get_physical_tendencies (source/physics.f90:43-72):
!  spectral prognostics of one time level, phi and ln(ps) in; the four grid tendencies inout.  The body is NOT the model's
!  column physics (out of scope): a few linear terms that read every argument through the drop-in `spectral` module, so the
!  test can restate them exactly.
module physics
    use types, only: p
    use params
    implicit none
    private
    public get_physical_tendencies
contains
    subroutine get_physical_tendencies(vor, div, t, q, phi, psl, utend, vtend, ttend, qtend)
        use spectral, only: spec_to_grid
        complex(p), intent(in) :: vor(mx,nx,kx), div(mx,nx,kx), t(mx,nx,kx), q(mx,nx,kx), phi(mx,nx,kx), psl(mx,nx)
        real(p), intent(inout) :: utend(ix,il,kx), vtend(ix,il,kx), ttend(ix,il,kx), qtend(ix,il,kx)
        real(p) :: g(ix,il), pslg(ix,il)
        integer :: k

        pslg = spec_to_grid(psl, 1)
        do k = 1, kx
            g = spec_to_grid(phi(:,:,k), 1)
            utend(:,:,k) = utend(:,:,k) + 1.0e-12_p*g                   ! a "push" built from the geopotential (1e-9 until round 5:
                                                                        ! 0.5 m/s per step on the timing loop's state)
            g = spec_to_grid(vor(:,:,k), 1)
            vtend(:,:,k) = 0.999_p*vtend(:,:,k) + 1.0e-3_p*g
            g = spec_to_grid(t(:,:,k), 1)
            ttend(:,:,k) = ttend(:,:,k) - 1.0e-6_p*(g - 250.0_p)        ! relaxation towards 250 K
            g = spec_to_grid(q(:,:,k), 1) + spec_to_grid(div(:,:,k), 1)
            qtend(:,:,k) = qtend(:,:,k) + 1.0e-7_p*pslg - 1.0e-6_p*g
        end do
    end subroutine
end module
