!> HIP-backed bodies for the two spectral-space tail routines of the reference, to be called from
!  its own modules (see INTEGRATION.md for the three-line patches):
!    horizontal_diffusion.f90:86-105  do_horizontal_diffusion_2d / _3d  -> spdy_do_horizontal_diffusion
!    implicit.f90:36,168              initialize_implicit / implicit_terms -> spdy_initialize_implicit, spdy_implicit_terms_f
!  The damping tables stay where the reference keeps them (module horizontal_diffusion publics);
!  they are passed in exactly as the reference passes them.
module spdy_tail
    use iso_c_binding
    use types, only: p
    use params
    use spdy_c
    use spectral, only: spectral_plan

    implicit none

    private
    public spdy_do_horizontal_diffusion, spdy_initialize_implicit, spdy_implicit_terms_f

    interface spdy_do_horizontal_diffusion
        module procedure hdiff_2d
        module procedure hdiff_3d
    end interface

contains
    function hdiff_2d(field, fdt_in, dmp, dmp1) result(fdt_out)
        complex(p), intent(in) :: field(mx,nx), fdt_in(mx,nx)
        complex(p) :: fdt_out(mx,nx)
        real(p), intent(in) :: dmp(mx,nx), dmp1(mx,nx)
        call spdy_check(spdy_hdiff(spectral_plan, 1_c_int, field, fdt_in, dmp, dmp1, fdt_out), 'do_horizontal_diffusion_2d')
    end function

    function hdiff_3d(field, fdt_in, dmp, dmp1) result(fdt_out)
        complex(p), intent(in) :: field(mx,nx,kx), fdt_in(mx,nx,kx)
        complex(p) :: fdt_out(mx,nx,kx)
        real(p), intent(in) :: dmp(mx,nx), dmp1(mx,nx)
        call spdy_check(spdy_hdiff(spectral_plan, int(kx, c_int), field, fdt_in, dmp, dmp1, fdt_out), 'do_horizontal_diffusion_3d')
    end function

    subroutine spdy_initialize_implicit(dt)
        real(p), intent(in) :: dt
        call spdy_check(spdy_implicit_init(spectral_plan, real(dt, c_double)), 'initialize_implicit')
    end subroutine

    subroutine spdy_implicit_terms_f(divdt, tdt, psdt)
        complex(p), intent(inout) :: divdt(mx,nx,kx), tdt(mx,nx,kx), psdt(mx,nx)
        call spdy_check(spdy_implicit_terms(spectral_plan, divdt, tdt, psdt), 'implicit_terms')
    end subroutine
end module
