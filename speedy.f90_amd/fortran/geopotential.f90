!> Drop-in replacement for the reference's `geopotential` module (source/geopotential.f90): same module name and
!  public set (:11: initialize_geopotential, get_geopotential), so initialization.f90:19, prognostics.f90:42 and
!  tendencies.f90:54,246 compile unchanged.  The hydrostatic coefficients live in the plan.
module geopotential
    use iso_c_binding
    use types, only: p
    use params
    use spdy_c
    use spectral, only: spectral_plan, initialize_spectral

    implicit none

    private
    public initialize_geopotential, get_geopotential

contains
    !> geopotential.f90:18 -- the coefficients xgeop1/xgeop2 are plan tables; nothing left to do but make sure it exists
    subroutine initialize_geopotential
        call initialize_spectral
    end subroutine

    !> geopotential.f90:33
    function get_geopotential(t, phis) result(phi)
        complex(p), intent(in) :: t(mx,nx,kx)
        complex(p), intent(in) :: phis(mx,nx)
        complex(p) :: phi(mx,nx,kx)
        call spdy_check(spdy_geopotential(spectral_plan, t, phis, phi), 'get_geopotential')
    end function
end module
