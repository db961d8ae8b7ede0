!> Drop-in replacement for the reference's `implicit` module (source/implicit.f90): same module name and public set
!  (:10-11: initialize_implicit, implicit_terms, tref, tref2, tref3), so tendencies.f90:12,53,247 and
!  time_stepping.f90:13 compile unchanged.  initialize_implicit(dt) rebuilds the plan's semi-implicit tables
!  (implicit.f90:36-165, host side of libspdy, uploaded to the GPU), refreshes the public reference-temperature
!  profiles and -- like the reference (implicit.f90:50-56) -- the dmp1* tables of module horizontal_diffusion.
module implicit
    use iso_c_binding
    use types, only: p
    use params
    use spdy_c
    use spectral, only: spectral_plan

    implicit none

    private
    public initialize_implicit, implicit_terms
    public tref, tref2, tref3

    real(p), dimension(kx) :: tref  !! Temperature profile for the semi-implicit scheme
    real(p), dimension(kx) :: tref2 !! akap * tref
    real(p), dimension(kx) :: tref3 !! fsgr * tref

contains
    !> implicit.f90:36
    subroutine initialize_implicit(dt)
        use horizontal_diffusion, only: dmp1, dmp1d, dmp1s
        real(p), intent(in) :: dt
        call spdy_flush_pending()    ! leapfrog steps time_stepping has collected but not launched yet belong to the OLD tables
        call spdy_check(spdy_implicit_init(spectral_plan, real(dt, c_double)), 'initialize_implicit')
        call spdy_check(spdy_get_table(spectral_plan, 'dmp1'//c_null_char, dmp1, int(mx*nx, c_int)), 'dmp1')
        call spdy_check(spdy_get_table(spectral_plan, 'dmp1d'//c_null_char, dmp1d, int(mx*nx, c_int)), 'dmp1d')
        call spdy_check(spdy_get_table(spectral_plan, 'dmp1s'//c_null_char, dmp1s, int(mx*nx, c_int)), 'dmp1s')
        call spdy_check(spdy_get_table(spectral_plan, 'tref'//c_null_char, tref, int(kx, c_int)), 'tref')
        call spdy_check(spdy_get_table(spectral_plan, 'tref2'//c_null_char, tref2, int(kx, c_int)), 'tref2')
        call spdy_check(spdy_get_table(spectral_plan, 'tref3'//c_null_char, tref3, int(kx, c_int)), 'tref3')
    end subroutine

    !> implicit.f90:168
    subroutine implicit_terms(divdt, tdt, psdt)
        complex(p), intent(inout) :: divdt(mx,nx,kx), tdt(mx,nx,kx), psdt(mx,nx)
        call spdy_check(spdy_implicit_terms(spectral_plan, divdt, tdt, psdt), 'implicit_terms')
    end subroutine
end module
