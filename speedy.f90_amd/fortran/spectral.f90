!> Drop-in replacement for the reference's `spectral` module (source/spectral.f90): same module
!  name, same public names, same argument kinds/shapes and result shapes -- every caller in
!  tendencies.f90, physics.f90, input_output.f90, prognostics.f90, boundaries.f90, forcing.f90,
!  sppt.f90, diagnostics.f90 and time_stepping.f90 compiles unchanged.  The bodies only forward
!  to the MI355X HIP path through the C ABI (include/spdy.h, module spdy_c); the module state
!  the reference keeps in private tables is one opaque plan handle here.
!
!  Array dimensions come from the host model's own `params` module, exactly like the reference
!  (spectral.f90:3).  Supported builds: T30 (trunc=30, ix=96, iy=24) and T63 (63, 192, 48).
!
!  Besides the reference API the module offers batched variants (`*_levels`) that transform a
!  whole (.., kx) stack in one call -- the per-level loops of tendencies.f90:89-107 / physics.f90:95-104
!  can switch to them for one PCIe round trip per stack instead of one per level.
module spectral
    use iso_c_binding
    use types, only: p
    use params
    use spdy_c

    implicit none

    private
    public el2
    public initialize_spectral
    public laplacian, inverse_laplacian, spec_to_grid, grid_to_spec
    public grad, vds, uvspec, vdspec, trunct
    ! extensions (not in the reference)
    public spec_to_grid_levels, grid_to_spec_levels, finalize_spectral, spectral_plan
    public uvspec_to_grid_levels, grad_to_grid_levels, vdspec_levels

    real(p), dimension(mx,nx) :: el2            !! l(l+1)/a^2, public in the reference (spectral.f90:8)
    type(c_ptr) :: spectral_plan = c_null_ptr   !! opaque spdy_plan*; replaces the private tables

contains
    !> spectral.f90:20 -- builds the plan (Legendre/FFT/operator tables, device upload).
    subroutine initialize_spectral
        integer(c_int) :: rc
        if (c_associated(spectral_plan)) return
        ! one process per GPU: the device is $SPDY_DEVICE, else the launcher's local rank, else 0 (SPDY_DEVICE_AUTO)
        rc = spdy_plan_create(int(trunc, c_int), int(ix, c_int), int(iy, c_int), int(kx, c_int), &
                            & int(max(8*kx, 64), c_int), SPDY_DEVICE_AUTO, spectral_plan)
        call spdy_check(rc, 'spdy_plan_create')
        rc = spdy_get_table(spectral_plan, 'el2'//c_null_char, el2, int(mx*nx, c_int))
        call spdy_check(rc, 'spdy_get_table(el2)')
    end subroutine

    subroutine finalize_spectral
        integer(c_int) :: rc
        if (c_associated(spectral_plan)) rc = spdy_plan_destroy(spectral_plan)
        spectral_plan = c_null_ptr
    end subroutine

    !> spectral.f90:84
    function laplacian(input) result(output)
        complex(p), intent(in) :: input(mx,nx)
        complex(p) :: output(mx,nx)
        call spdy_check(spdy_laplacian(spectral_plan, 1_c_int, input, output), 'laplacian')
    end function

    !> spectral.f90:91
    function inverse_laplacian(input) result(output)
        complex(p), intent(in) :: input(mx,nx)
        complex(p) :: output(mx,nx)
        call spdy_check(spdy_inverse_laplacian(spectral_plan, 1_c_int, input, output), 'inverse_laplacian')
    end function

    !> spectral.f90:98
    function spec_to_grid(vorm, kcos) result(vorg)
        complex(p), intent(in) :: vorm(mx,nx)
        integer, intent(in) :: kcos
        real(p) :: vorg(ix,il)
        call spdy_check(spdy_spec_to_grid(spectral_plan, vorm, int(kcos, c_int), vorg), 'spec_to_grid')
    end function

    !> spectral.f90:112
    function grid_to_spec(vorg) result(vorm)
        real(p), intent(in) :: vorg(ix,il)
        complex(p) :: vorm(mx,nx)
        call spdy_check(spdy_grid_to_spec(spectral_plan, vorg, vorm), 'grid_to_spec')
    end function

    !> spectral.f90:124
    subroutine grad(psi,psdx,psdy)
        complex(p), dimension(mx,nx), intent(inout) :: psi
        complex(p), dimension(mx,nx), intent(inout) :: psdx, psdy
        call spdy_check(spdy_grad(spectral_plan, 1_c_int, psi, psdx, psdy), 'grad')
    end

    !> spectral.f90:146
    subroutine vds(ucosm,vcosm,vorm,divm)
        complex(p), dimension(mx,nx) :: ucosm, vcosm
        complex(p), dimension(mx,nx), intent(inout) :: vorm, divm
        call spdy_check(spdy_vds(spectral_plan, 1_c_int, ucosm, vcosm, vorm, divm), 'vds')
    end

    !> spectral.f90:173
    subroutine uvspec(vorm,divm,ucosm,vcosm)
        complex(p), dimension(mx,nx), intent(in) :: vorm,divm
        complex(p), dimension(mx,nx), intent(inout) :: ucosm,vcosm
        call spdy_check(spdy_uvspec(spectral_plan, 1_c_int, vorm, divm, ucosm, vcosm), 'uvspec')
    end

    !> spectral.f90:198
    subroutine vdspec(ug,vg,vorm,divm,kcos)
        real(p), intent(in) :: ug(ix,il), vg(ix,il)
        complex(p), intent(out) :: vorm(mx,nx), divm(mx,nx)
        integer, intent(in) :: kcos
        call spdy_check(spdy_vdspec(spectral_plan, 1_c_int, ug, vg, vorm, divm, int(kcos, c_int)), 'vdspec')
    end

    !> spectral.f90:229
    subroutine trunct(vor)
        complex(p), intent(inout) :: vor(mx,nx)
        call spdy_check(spdy_trunct(spectral_plan, 1_c_int, vor), 'trunct')
    end

    !> Extension: nlev independent inverse transforms in one call (kcos per level).
    subroutine spec_to_grid_levels(nlev, vorm, kcos, vorg)
        integer, intent(in) :: nlev
        complex(p), intent(in) :: vorm(mx,nx,nlev)
        integer, intent(in) :: kcos(nlev)
        real(p), intent(out) :: vorg(ix,il,nlev)
        integer(c_int) :: kc(nlev)
        kc = int(kcos, c_int)
        call spdy_check(spdy_spec_to_grid_batch(spectral_plan, int(nlev, c_int), vorm, kc, vorg), 'spec_to_grid_levels')
    end subroutine

    !> Extension: nlev independent direct transforms in one call.
    subroutine grid_to_spec_levels(nlev, vorg, vorm)
        integer, intent(in) :: nlev
        real(p), intent(in) :: vorg(ix,il,nlev)
        complex(p), intent(out) :: vorm(mx,nx,nlev)
        call spdy_check(spdy_grid_to_spec_batch(spectral_plan, int(nlev, c_int), vorg, vorm), 'grid_to_spec_levels')
    end subroutine

    !> Extension: the loop body of tendencies.f90:98-100 for a whole level stack in one call --
    !  uvspec(vorm(:,:,k), divm(:,:,k), u, v); ug(:,:,k) = spec_to_grid(u, 2); vg(:,:,k) = spec_to_grid(v, 2)
    subroutine uvspec_to_grid_levels(nlev, vorm, divm, ug, vg)
        integer, intent(in) :: nlev
        complex(p), intent(in) :: vorm(mx,nx,nlev), divm(mx,nx,nlev)
        real(p), intent(out) :: ug(ix,il,nlev), vg(ix,il,nlev)
        call spdy_check(spdy_uvspec_to_grid(spectral_plan, int(nlev, c_int), vorm, divm, ug, vg, 2_c_int), 'uvspec_to_grid_levels')
    end subroutine

    !> Extension: grad(psi(:,:,k), dx, dy) followed by spec_to_grid(dx, 2), spec_to_grid(dy, 2) (tendencies.f90:121-123)
    subroutine grad_to_grid_levels(nlev, psi, gx, gy)
        integer, intent(in) :: nlev
        complex(p), intent(in) :: psi(mx,nx,nlev)
        real(p), intent(out) :: gx(ix,il,nlev), gy(ix,il,nlev)
        call spdy_check(spdy_grad_to_grid(spectral_plan, int(nlev, c_int), psi, gx, gy, 2_c_int), 'grad_to_grid_levels')
    end subroutine

    !> Extension: vdspec for a whole level stack (tendencies.f90:216-232)
    subroutine vdspec_levels(nlev, ug, vg, vorm, divm, kcos)
        integer, intent(in) :: nlev, kcos
        real(p), intent(in) :: ug(ix,il,nlev), vg(ix,il,nlev)
        complex(p), intent(out) :: vorm(mx,nx,nlev), divm(mx,nx,nlev)
        call spdy_check(spdy_vdspec(spectral_plan, int(nlev, c_int), ug, vg, vorm, divm, int(kcos, c_int)), 'vdspec_levels')
    end subroutine
end module
