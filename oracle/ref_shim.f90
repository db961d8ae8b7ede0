! TEST INFRASTRUCTURE ONLY -- never linked into, imported by or called from the product path.
!
! bind(C) shim over the *public* API of the reference's hot-path modules
! (/root/reference/source/{geometry,fourier,legendre,spectral,horizontal_diffusion,implicit}.f90).
! oracle/build_ref.sh compiles the reference sources where they lie, together with this
! file, into oracle/_ref/libspeedy_ref_<res>.so.  Nothing of the reference is copied here:
! every routine below only forwards to a reference procedure and moves data across the C ABI.
!
! Array shapes come from the reference's compile-time `params` module, so the same shim
! serves the stock T30 build and the T63 build (params.f90 patched in a scratch dir).

subroutine ref_dims(d) bind(C, name="ref_dims")
    use iso_c_binding
    use params, only: trunc, ix, iy, il, kx, nx, mx
    integer(c_int), intent(out) :: d(7)
    d = (/ trunc, ix, iy, il, kx, nx, mx /)
end subroutine

! geometry.f90:35 initialize_geometry ; spectral.f90:20 initialize_spectral
subroutine ref_init() bind(C, name="ref_init")
    use geometry, only: initialize_geometry
    use spectral, only: initialize_spectral
    call initialize_geometry
    call initialize_spectral
end subroutine

! geometry.f90:11-31 public latitude tables
subroutine ref_get_geometry(o_sia_half, o_coa_half, o_cosgr, o_cosgr2) bind(C, name="ref_get_geometry")
    use iso_c_binding
    use params, only: iy, il
    use geometry, only: sia_half, coa_half, cosgr, cosgr2
    real(c_double), intent(out) :: o_sia_half(iy), o_coa_half(il), o_cosgr(il), o_cosgr2(il)
    o_sia_half = sia_half
    o_coa_half = coa_half
    o_cosgr = cosgr
    o_cosgr2 = cosgr2
end subroutine

! geometry.f90:14-18 sigma-level tables
subroutine ref_get_sigma(o_hsg, o_dhs, o_fsg, o_dhsr, o_fsgr) bind(C, name="ref_get_sigma")
    use iso_c_binding
    use params, only: kx
    use geometry, only: hsg, dhs, fsg, dhsr, fsgr
    real(c_double), intent(out) :: o_hsg(kx+1), o_dhs(kx), o_fsg(kx), o_dhsr(kx), o_fsgr(kx)
    o_hsg = hsg
    o_dhs = dhs
    o_fsg = fsg
    o_dhsr = dhsr
    o_fsgr = fsgr
end subroutine

! fftpack.f90:1 rffti1 (external subroutine)
subroutine ref_rffti1(n, wa, ifac) bind(C, name="ref_rffti1")
    use iso_c_binding
    integer(c_int), value :: n
    real(c_double), intent(inout) :: wa(n)
    integer(c_int), intent(inout) :: ifac(15)
    external rffti1
    call rffti1(n, wa, ifac)
end subroutine

! fftpack.f90:69 rfftb1 / :136 rfftf1 on one length-n vector (c in place, ch scratch)
subroutine ref_rfftb1(n, c, wa, ifac) bind(C, name="ref_rfftb1")
    use iso_c_binding
    integer(c_int), value :: n
    real(c_double), intent(inout) :: c(n), wa(n)
    integer(c_int), intent(in) :: ifac(15)
    real(c_double) :: ch(n)
    external rfftb1
    call rfftb1(n, c, ch, wa, ifac)
end subroutine

subroutine ref_rfftf1(n, c, wa, ifac) bind(C, name="ref_rfftf1")
    use iso_c_binding
    integer(c_int), value :: n
    real(c_double), intent(inout) :: c(n), wa(n)
    integer(c_int), intent(in) :: ifac(15)
    real(c_double) :: ch(n)
    external rfftf1
    call rfftf1(n, c, ch, wa, ifac)
end subroutine

! legendre.f90:12 public epsi ; spectral.f90:8 public el2
subroutine ref_get_epsi(o) bind(C, name="ref_get_epsi")
    use iso_c_binding
    use params, only: mx, nx
    use legendre, only: epsi
    real(c_double), intent(out) :: o(mx+1,nx+1)
    o = epsi
end subroutine

subroutine ref_get_el2(o) bind(C, name="ref_get_el2")
    use iso_c_binding
    use params, only: mx, nx
    use spectral, only: el2
    real(c_double), intent(out) :: o(mx,nx)
    o = el2
end subroutine

! fourier.f90:23 fourier_inv ; fourier.f90:56 fourier_dir
subroutine ref_fourier_inv(f, kcos, g) bind(C, name="ref_fourier_inv")
    use iso_c_binding
    use params, only: mx, ix, il
    use fourier, only: fourier_inv
    real(c_double), intent(in) :: f(2*mx,il)
    integer(c_int), value :: kcos
    real(c_double), intent(out) :: g(ix,il)
    g = fourier_inv(f, kcos)
end subroutine

subroutine ref_fourier_dir(g, f) bind(C, name="ref_fourier_dir")
    use iso_c_binding
    use params, only: mx, ix, il
    use fourier, only: fourier_dir
    real(c_double), intent(in) :: g(ix,il)
    real(c_double), intent(out) :: f(2*mx,il)
    f = fourier_dir(g)
end subroutine

! legendre.f90:74 legendre_inv ; legendre.f90:114 legendre_dir
subroutine ref_legendre_inv(s, f) bind(C, name="ref_legendre_inv")
    use iso_c_binding
    use params, only: mx, nx, il
    use legendre, only: legendre_inv
    real(c_double), intent(in) :: s(2*mx,nx)
    real(c_double), intent(out) :: f(2*mx,il)
    f = legendre_inv(s)
end subroutine

subroutine ref_legendre_dir(f, s) bind(C, name="ref_legendre_dir")
    use iso_c_binding
    use params, only: mx, nx, il
    use legendre, only: legendre_dir
    real(c_double), intent(in) :: f(2*mx,il)
    real(c_double), intent(out) :: s(2*mx,nx)
    s = legendre_dir(f)
end subroutine

! spectral.f90:98 spec_to_grid ; spectral.f90:112 grid_to_spec
subroutine ref_spec_to_grid(s, kcos, g) bind(C, name="ref_spec_to_grid")
    use iso_c_binding
    use params, only: mx, nx, ix, il
    use spectral, only: spec_to_grid
    complex(c_double_complex), intent(in) :: s(mx,nx)
    integer(c_int), value :: kcos
    real(c_double), intent(out) :: g(ix,il)
    g = spec_to_grid(s, kcos)
end subroutine

subroutine ref_grid_to_spec(g, s) bind(C, name="ref_grid_to_spec")
    use iso_c_binding
    use params, only: mx, nx, ix, il
    use spectral, only: grid_to_spec
    real(c_double), intent(in) :: g(ix,il)
    complex(c_double_complex), intent(out) :: s(mx,nx)
    s = grid_to_spec(g)
end subroutine

! spectral.f90:84 laplacian ; :91 inverse_laplacian ; :229 trunct
subroutine ref_laplacian(a, o) bind(C, name="ref_laplacian")
    use iso_c_binding
    use params, only: mx, nx
    use spectral, only: laplacian
    complex(c_double_complex), intent(in) :: a(mx,nx)
    complex(c_double_complex), intent(out) :: o(mx,nx)
    o = laplacian(a)
end subroutine

subroutine ref_inverse_laplacian(a, o) bind(C, name="ref_inverse_laplacian")
    use iso_c_binding
    use params, only: mx, nx
    use spectral, only: inverse_laplacian
    complex(c_double_complex), intent(in) :: a(mx,nx)
    complex(c_double_complex), intent(out) :: o(mx,nx)
    o = inverse_laplacian(a)
end subroutine

subroutine ref_trunct(a) bind(C, name="ref_trunct")
    use iso_c_binding
    use params, only: mx, nx
    use spectral, only: trunct
    complex(c_double_complex), intent(inout) :: a(mx,nx)
    call trunct(a)
end subroutine

! spectral.f90:124 grad ; :146 vds ; :173 uvspec ; :198 vdspec
subroutine ref_grad(psi, psdx, psdy) bind(C, name="ref_grad")
    use iso_c_binding
    use params, only: mx, nx
    use spectral, only: grad
    complex(c_double_complex), intent(inout) :: psi(mx,nx), psdx(mx,nx), psdy(mx,nx)
    call grad(psi, psdx, psdy)
end subroutine

subroutine ref_vds(ucosm, vcosm, vorm, divm) bind(C, name="ref_vds")
    use iso_c_binding
    use params, only: mx, nx
    use spectral, only: vds
    complex(c_double_complex), intent(inout) :: ucosm(mx,nx), vcosm(mx,nx), vorm(mx,nx), divm(mx,nx)
    call vds(ucosm, vcosm, vorm, divm)
end subroutine

subroutine ref_uvspec(vorm, divm, ucosm, vcosm) bind(C, name="ref_uvspec")
    use iso_c_binding
    use params, only: mx, nx
    use spectral, only: uvspec
    complex(c_double_complex), intent(in) :: vorm(mx,nx), divm(mx,nx)
    complex(c_double_complex), intent(inout) :: ucosm(mx,nx), vcosm(mx,nx)
    call uvspec(vorm, divm, ucosm, vcosm)
end subroutine

subroutine ref_vdspec(ug, vg, vorm, divm, kcos) bind(C, name="ref_vdspec")
    use iso_c_binding
    use params, only: mx, nx, ix, il
    use spectral, only: vdspec
    real(c_double), intent(in) :: ug(ix,il), vg(ix,il)
    complex(c_double_complex), intent(out) :: vorm(mx,nx), divm(mx,nx)
    integer(c_int), value :: kcos
    call vdspec(ug, vg, vorm, divm, kcos)
end subroutine

! horizontal_diffusion.f90:36 initialize_horizontal_diffusion ; implicit.f90:36 initialize_implicit
subroutine ref_tail_init(dt) bind(C, name="ref_tail_init")
    use iso_c_binding
    use horizontal_diffusion, only: initialize_horizontal_diffusion
    use implicit, only: initialize_implicit
    real(c_double), value :: dt
    call initialize_horizontal_diffusion
    call initialize_implicit(dt)
end subroutine

! horizontal_diffusion.f90:12 public damping tables
subroutine ref_get_dmp(o_dmp, o_dmpd, o_dmps, o_dmp1, o_dmp1d, o_dmp1s) bind(C, name="ref_get_dmp")
    use iso_c_binding
    use params, only: mx, nx
    use horizontal_diffusion, only: dmp, dmpd, dmps, dmp1, dmp1d, dmp1s
    real(c_double), intent(out), dimension(mx,nx) :: o_dmp, o_dmpd, o_dmps, o_dmp1, o_dmp1d, o_dmp1s
    o_dmp = dmp
    o_dmpd = dmpd
    o_dmps = dmps
    o_dmp1 = dmp1
    o_dmp1d = dmp1d
    o_dmp1s = dmp1s
end subroutine

! implicit.f90:11 public reference-temperature profiles
subroutine ref_get_tref(o_tref, o_tref2, o_tref3) bind(C, name="ref_get_tref")
    use iso_c_binding
    use params, only: kx
    use implicit, only: tref, tref2, tref3
    real(c_double), intent(out), dimension(kx) :: o_tref, o_tref2, o_tref3
    o_tref = tref
    o_tref2 = tref2
    o_tref3 = tref3
end subroutine

! horizontal_diffusion.f90:86 / :96 do_horizontal_diffusion (2-D and 3-D specifics)
subroutine ref_hdiff_2d(field, fdt_in, dmp, dmp1, fdt_out) bind(C, name="ref_hdiff_2d")
    use iso_c_binding
    use params, only: mx, nx
    use horizontal_diffusion, only: do_horizontal_diffusion
    complex(c_double_complex), intent(in) :: field(mx,nx), fdt_in(mx,nx)
    real(c_double), intent(in) :: dmp(mx,nx), dmp1(mx,nx)
    complex(c_double_complex), intent(out) :: fdt_out(mx,nx)
    fdt_out = do_horizontal_diffusion(field, fdt_in, dmp, dmp1)
end subroutine

subroutine ref_hdiff_3d(field, fdt_in, dmp, dmp1, fdt_out) bind(C, name="ref_hdiff_3d")
    use iso_c_binding
    use params, only: mx, nx, kx
    use horizontal_diffusion, only: do_horizontal_diffusion
    complex(c_double_complex), intent(in) :: field(mx,nx,kx), fdt_in(mx,nx,kx)
    real(c_double), intent(in) :: dmp(mx,nx), dmp1(mx,nx)
    complex(c_double_complex), intent(out) :: fdt_out(mx,nx,kx)
    fdt_out = do_horizontal_diffusion(field, fdt_in, dmp, dmp1)
end subroutine

! implicit.f90:168 implicit_terms
subroutine ref_implicit_terms(divdt, tdt, psdt) bind(C, name="ref_implicit_terms")
    use iso_c_binding
    use params, only: mx, nx, kx
    use implicit, only: implicit_terms
    complex(c_double_complex), intent(inout) :: divdt(mx,nx,kx), tdt(mx,nx,kx), psdt(mx,nx)
    call implicit_terms(divdt, tdt, psdt)
end subroutine

! CPU baseline (bench.py cpu_baseline.kind = "reference"): nrep passes of
! grid_to_spec followed by spec_to_grid(.,1) over nf independent 2-D fields,
! one field at a time exactly as the reference executes them (spectral.f90:98-122).
subroutine ref_roundtrip_loop(nf, nrep, g_in, g_out) bind(C, name="ref_roundtrip_loop")
    use iso_c_binding
    use params, only: mx, nx, ix, il
    use spectral, only: grid_to_spec, spec_to_grid
    integer(c_int), value :: nf, nrep
    real(c_double), intent(in) :: g_in(ix,il,nf)
    real(c_double), intent(out) :: g_out(ix,il,nf)
    complex(c_double_complex) :: s(mx,nx)
    integer :: f, r
    do r = 1, nrep
        do f = 1, nf
            s = grid_to_spec(g_in(:,:,f))
            g_out(:,:,f) = spec_to_grid(s, 1)
        end do
    end do
end subroutine

! geometry.f90:14-18: the sigma-level tables are public module data.  For a level count the reference has no set
! for (geometry.f90:42-48 knows kx = 5, 7, 8; this build may be compiled with another kx), the test supplies them;
! nothing of the reference is re-implemented here -- initialize_implicit / initialize_geopotential /
! initialize_horizontal_diffusion then run unchanged on these values.
subroutine ref_set_sigma(i_hsg, i_dhs, i_fsg, i_dhsr, i_fsgr) bind(C, name="ref_set_sigma")
    use iso_c_binding
    use params, only: kx
    use geometry, only: hsg, dhs, fsg, dhsr, fsgr
    real(c_double), intent(in) :: i_hsg(kx+1), i_dhs(kx), i_fsg(kx), i_dhsr(kx), i_fsgr(kx)
    hsg = i_hsg
    dhs = i_dhs
    fsg = i_fsg
    dhsr = i_dhsr
    fsgr = i_fsgr
end subroutine

! geometry.f90:31 public coriol
subroutine ref_get_coriol(o) bind(C, name="ref_get_coriol")
    use iso_c_binding
    use params, only: il
    use geometry, only: coriol
    real(c_double), intent(out) :: o(il)
    o = coriol
end subroutine

! horizontal_diffusion.f90:27-28 public tcorv, qcorv (after ref_tail_init)
subroutine ref_get_corv(o_tcorv, o_qcorv) bind(C, name="ref_get_corv")
    use iso_c_binding
    use params, only: kx
    use horizontal_diffusion, only: tcorv, qcorv
    real(c_double), intent(out) :: o_tcorv(kx), o_qcorv(kx)
    o_tcorv = tcorv
    o_qcorv = qcorv
end subroutine

! geopotential.f90:18 initialize_geopotential ; :33 get_geopotential
subroutine ref_geopotential(t, phis, phi) bind(C, name="ref_geopotential")
    use iso_c_binding
    use params, only: mx, nx, kx
    use geopotential, only: initialize_geopotential, get_geopotential
    complex(c_double_complex), intent(in) :: t(mx,nx,kx), phis(mx,nx)
    complex(c_double_complex), intent(out) :: phi(mx,nx,kx)
    call initialize_geopotential
    phi = get_geopotential(t, phis)
end subroutine

! time_stepping.f90:126-167 step_field_3d / step_field_2d, compiled from the reference file itself (build_ref.sh cuts the two
! functions into the scratch module step_field_ref).  field = both time levels (mx,nx,nlev,2) in and out; fdt is truncated
! in place as the reference does (ix == 4*iy).  nlev = kx -> step_field_3d, nlev = 1 -> step_field_2d.
subroutine ref_step_field_3d(j1, dt, eps, field, fdt) bind(C, name="ref_step_field_3d")
    use iso_c_binding
    use params, only: mx, nx, kx
    use step_field_ref, only: step_field_3d
    integer(c_int), value :: j1
    real(c_double), value :: dt, eps
    complex(c_double_complex), intent(inout) :: field(mx,nx,kx,2), fdt(mx,nx,kx)
    field = step_field_3d(int(j1), dt, eps, field, fdt)
end subroutine

subroutine ref_step_field_2d(j1, dt, eps, field, fdt) bind(C, name="ref_step_field_2d")
    use iso_c_binding
    use params, only: mx, nx
    use step_field_ref, only: step_field_2d
    integer(c_int), value :: j1
    real(c_double), value :: dt, eps
    complex(c_double_complex), intent(inout) :: field(mx,nx,2), fdt(mx,nx)
    field = step_field_2d(int(j1), dt, eps, field, fdt)
end subroutine

subroutine ref_get_wil_rob(o_wil, o_rob) bind(C, name="ref_get_wil_rob")
    use iso_c_binding
    use params, only: wil, rob
    real(c_double), intent(out) :: o_wil, o_rob
    o_wil = wil
    o_rob = rob
end subroutine

! tendencies.f90:241-293 get_spectral_tendencies, compiled from the reference file itself (build_ref.sh cuts the subroutine
! into the scratch module spectral_tendencies_ref and the declaration part of prognostics.f90 into module prognostics).  The
! time-level-j2 slabs of div, t, ps and phis are assigned to the reference's own module arrays; phi is read back from them.
! Needs ref_tail_init (tref, tref2, tref3 of implicit) first, like the model (time_stepping.f90:14).
subroutine ref_spectral_tendencies(j2, i_div, i_t, i_ps, i_phis, divdt, tdt, psdt, o_phi) bind(C, name="ref_spectral_tendencies")
    use iso_c_binding
    use params, only: mx, nx, kx
    use prognostics, only: div, t, ps, phis, phi
    use geopotential, only: initialize_geopotential
    use spectral_tendencies_ref, only: get_spectral_tendencies
    integer(c_int), value :: j2
    complex(c_double_complex), intent(in) :: i_div(mx,nx,kx), i_t(mx,nx,kx), i_ps(mx,nx), i_phis(mx,nx)
    complex(c_double_complex), intent(inout) :: divdt(mx,nx,kx), tdt(mx,nx,kx), psdt(mx,nx)
    complex(c_double_complex), intent(out) :: o_phi(mx,nx,kx)
    call initialize_geopotential
    div(:,:,:,j2) = i_div
    t(:,:,:,j2) = i_t
    ps(:,:,j2) = i_ps
    phis = i_phis
    call get_spectral_tendencies(divdt, tdt, psdt, int(j2))
    o_phi = phi
end subroutine

! time_stepping.f90:35-118 step(j1, j2, dt) -- the reference file itself, compiled unchanged -- on top of the reference's
! tendencies.f90 minus its three physics lines (build_ref.sh): one ADIABATIC time step of the reference.  The prognostics
! (both time levels), phis and the orographic-correction fields tcorh / qcorh are assigned to the reference's own public
! module variables; the stepped prognostics and phi are read back from them.  Needs ref_tail_init(dt) first
! (initialize_horizontal_diffusion + initialize_implicit, as time_stepping.f90:12-24 / initialization.f90:20 do).
subroutine ref_step(j1, j2, dt, io_vor, io_div, io_t, io_tr, io_ps, i_phis, i_tcorh, i_qcorh, o_phi) bind(C, name="ref_step")
    use iso_c_binding
    use params, only: mx, nx, kx
    use prognostics, only: vor, div, t, tr, ps, phis, phi
    use horizontal_diffusion, only: tcorh, qcorh
    use geopotential, only: initialize_geopotential
    use time_stepping, only: step
    integer(c_int), value :: j1, j2
    real(c_double), value :: dt
    complex(c_double_complex), intent(inout) :: io_vor(mx,nx,kx,2), io_div(mx,nx,kx,2), io_t(mx,nx,kx,2), io_tr(mx,nx,kx,2), io_ps(mx,nx,2)
    complex(c_double_complex), intent(in) :: i_phis(mx,nx), i_tcorh(mx,nx), i_qcorh(mx,nx)
    complex(c_double_complex), intent(out) :: o_phi(mx,nx,kx)
    call initialize_geopotential
    vor = io_vor
    div = io_div
    t = io_t
    tr(:,:,:,:,1) = io_tr
    ps = io_ps
    phis = i_phis
    tcorh = i_tcorh
    qcorh = i_qcorh
    call step(int(j1), int(j2), dt)
    io_vor = vor
    io_div = div
    io_t = t
    io_tr = tr(:,:,:,:,1)
    io_ps = ps
    o_phi = phi
end subroutine

! tendencies.f90:11-41 get_tendencies (same adiabatic build): the spectral tendencies of one step before the diffusion and the
! time integration -- transforms of time level j2, grid-space dynamical tendencies, direct transforms, get_spectral_tendencies
! and implicit_terms (alph = 0.5).
subroutine ref_get_tendencies(j2, i_vor, i_div, i_t, i_tr, i_ps, i_phis, vordt, divdt, tdt, psdt, trdt) bind(C, name="ref_get_tendencies")
    use iso_c_binding
    use params, only: mx, nx, kx
    use prognostics, only: vor, div, t, tr, ps, phis
    use geopotential, only: initialize_geopotential
    use tendencies, only: get_tendencies
    integer(c_int), value :: j2
    complex(c_double_complex), intent(in) :: i_vor(mx,nx,kx,2), i_div(mx,nx,kx,2), i_t(mx,nx,kx,2), i_tr(mx,nx,kx,2), i_ps(mx,nx,2), i_phis(mx,nx)
    complex(c_double_complex), intent(out) :: vordt(mx,nx,kx), divdt(mx,nx,kx), tdt(mx,nx,kx), psdt(mx,nx), trdt(mx,nx,kx,1)
    call initialize_geopotential
    vor = i_vor
    div = i_div
    t = i_t
    tr(:,:,:,:,1) = i_tr
    ps = i_ps
    phis = i_phis
    vordt = (0.0d0, 0.0d0); divdt = (0.0d0, 0.0d0); tdt = (0.0d0, 0.0d0); psdt = (0.0d0, 0.0d0); trdt = (0.0d0, 0.0d0)
    call get_tendencies(vordt, divdt, tdt, psdt, trdt, int(j2))
end subroutine

! input_output.f90:183-205, the computing lines of subroutine output, compiled from the reference file itself (build_ref.sh
! cuts them, with the subroutine's own declarations, into the scratch module output_fields_ref).  Inputs: the time-level-1 slabs
! of vor, div, t, tr(:,:,:,1,1) and ps, and phi; the reference subroutine reads nothing else of its arguments.
subroutine ref_output_fields(i_vor, i_div, i_t, i_q, i_phi, i_ps, u_out, v_out, t_out, q_out, phi_out, ps_out) bind(C, name="ref_output_fields")
    use iso_c_binding
    use params, only: mx, nx, kx, ix, il, ntr
    use output_fields_ref, only: output_fields
    complex(c_double_complex), intent(in) :: i_vor(mx,nx,kx), i_div(mx,nx,kx), i_t(mx,nx,kx), i_q(mx,nx,kx), i_phi(mx,nx,kx), i_ps(mx,nx)
    real(c_float), intent(out) :: u_out(ix,il,kx), v_out(ix,il,kx), t_out(ix,il,kx), q_out(ix,il,kx), phi_out(ix,il,kx), ps_out(ix,il)
    complex(c_double_complex), allocatable :: vor(:,:,:,:), div(:,:,:,:), t(:,:,:,:), ps(:,:,:), tr(:,:,:,:,:)
    allocate(vor(mx,nx,kx,2), div(mx,nx,kx,2), t(mx,nx,kx,2), ps(mx,nx,2), tr(mx,nx,kx,2,ntr))
    vor = (0.0d0, 0.0d0); div = vor; t = vor; ps = (0.0d0, 0.0d0); tr = (0.0d0, 0.0d0)
    vor(:,:,:,1) = i_vor
    div(:,:,:,1) = i_div
    t(:,:,:,1) = i_t
    tr(:,:,:,1,1) = i_q
    ps(:,:,1) = i_ps
    call output_fields(vor, div, t, ps, tr, i_phi, u_out, v_out, t_out, q_out, phi_out, ps_out)
end subroutine
