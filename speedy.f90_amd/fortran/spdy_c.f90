!> ISO_C_BINDING interfaces to the C ABI of include/spdy.h (libspdy.so).
!  One-to-one with the header; complex(c_double_complex) arrays are passed where the C side
!  takes `double*` of interleaved (re,im) pairs -- the storage is identical.
module spdy_c
    use iso_c_binding
    implicit none
    public

    !> spdy_spec_seg (include/spdy.h): one source array of plain spectra of spdy_inverse_batch_segs_dev
    type, bind(C) :: spdy_spec_seg
        integer(c_int) :: nb
        type(c_ptr) :: d_spec
    end type

    !> spdy_hdiff_op, spdy_step_op (include/spdy.h): one array of spdy_hdiff_multi_dev / spdy_step_fields_dev
    type, bind(C) :: spdy_hdiff_op
        integer(c_int) :: nlev
        type(c_ptr) :: field, fdt_in, d_dmp, d_dmp1, fdt_out
    end type
    type, bind(C) :: spdy_step_op
        integer(c_int) :: nlev
        type(c_ptr) :: field, fdt
    end type

    interface
        function spdy_plan_create(trunc, ix, iy, kx, max_batch, device, plan) bind(C, name="spdy_plan_create") result(rc)
            import :: c_int, c_ptr
            integer(c_int), value :: trunc, ix, iy, kx, max_batch, device
            type(c_ptr), intent(out) :: plan
            integer(c_int) :: rc
        end function
        function spdy_plan_destroy(plan) bind(C, name="spdy_plan_destroy") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int) :: rc
        end function
        function spdy_last_error() bind(C, name="spdy_last_error") result(msg)
            import :: c_ptr
            type(c_ptr) :: msg
        end function
        function spdy_get_table(plan, name, buf, cap) bind(C, name="spdy_get_table") result(rc)
            import :: c_int, c_ptr, c_char, c_double
            type(c_ptr), value :: plan
            character(kind=c_char), intent(in) :: name(*)
            real(c_double), intent(out) :: buf(*)
            integer(c_int), value :: cap
            integer(c_int) :: rc
        end function
        function spdy_spec_to_grid(plan, spec, kcos, grid) bind(C, name="spdy_spec_to_grid") result(rc)
            import :: c_int, c_ptr, c_double, c_double_complex
            type(c_ptr), value :: plan
            complex(c_double_complex), intent(in) :: spec(*)
            integer(c_int), value :: kcos
            real(c_double), intent(out) :: grid(*)
            integer(c_int) :: rc
        end function
        function spdy_grid_to_spec(plan, grid, spec) bind(C, name="spdy_grid_to_spec") result(rc)
            import :: c_int, c_ptr, c_double, c_double_complex
            type(c_ptr), value :: plan
            real(c_double), intent(in) :: grid(*)
            complex(c_double_complex), intent(out) :: spec(*)
            integer(c_int) :: rc
        end function
        function spdy_spec_to_grid_batch(plan, nb, spec, kcos, grid) bind(C, name="spdy_spec_to_grid_batch") result(rc)
            import :: c_int, c_ptr, c_double, c_double_complex
            type(c_ptr), value :: plan
            integer(c_int), value :: nb
            complex(c_double_complex), intent(in) :: spec(*)
            integer(c_int), intent(in) :: kcos(*)
            real(c_double), intent(out) :: grid(*)
            integer(c_int) :: rc
        end function
        function spdy_grid_to_spec_batch(plan, nb, grid, spec) bind(C, name="spdy_grid_to_spec_batch") result(rc)
            import :: c_int, c_ptr, c_double, c_double_complex
            type(c_ptr), value :: plan
            integer(c_int), value :: nb
            real(c_double), intent(in) :: grid(*)
            complex(c_double_complex), intent(out) :: spec(*)
            integer(c_int) :: rc
        end function
        function spdy_laplacian(plan, nb, a, o) bind(C, name="spdy_laplacian") result(rc)
            import :: c_int, c_ptr, c_double_complex
            type(c_ptr), value :: plan
            integer(c_int), value :: nb
            complex(c_double_complex), intent(in) :: a(*)
            complex(c_double_complex), intent(out) :: o(*)
            integer(c_int) :: rc
        end function
        function spdy_inverse_laplacian(plan, nb, a, o) bind(C, name="spdy_inverse_laplacian") result(rc)
            import :: c_int, c_ptr, c_double_complex
            type(c_ptr), value :: plan
            integer(c_int), value :: nb
            complex(c_double_complex), intent(in) :: a(*)
            complex(c_double_complex), intent(out) :: o(*)
            integer(c_int) :: rc
        end function
        function spdy_trunct(plan, nb, a) bind(C, name="spdy_trunct") result(rc)
            import :: c_int, c_ptr, c_double_complex
            type(c_ptr), value :: plan
            integer(c_int), value :: nb
            complex(c_double_complex), intent(inout) :: a(*)
            integer(c_int) :: rc
        end function
        function spdy_grad(plan, nb, psi, psdx, psdy) bind(C, name="spdy_grad") result(rc)
            import :: c_int, c_ptr, c_double_complex
            type(c_ptr), value :: plan
            integer(c_int), value :: nb
            complex(c_double_complex), intent(in) :: psi(*)
            complex(c_double_complex), intent(inout) :: psdx(*), psdy(*)
            integer(c_int) :: rc
        end function
        function spdy_vds(plan, nb, ucosm, vcosm, vorm, divm) bind(C, name="spdy_vds") result(rc)
            import :: c_int, c_ptr, c_double_complex
            type(c_ptr), value :: plan
            integer(c_int), value :: nb
            complex(c_double_complex), intent(in) :: ucosm(*), vcosm(*)
            complex(c_double_complex), intent(inout) :: vorm(*), divm(*)
            integer(c_int) :: rc
        end function
        function spdy_uvspec(plan, nb, vorm, divm, ucosm, vcosm) bind(C, name="spdy_uvspec") result(rc)
            import :: c_int, c_ptr, c_double_complex
            type(c_ptr), value :: plan
            integer(c_int), value :: nb
            complex(c_double_complex), intent(in) :: vorm(*), divm(*)
            complex(c_double_complex), intent(inout) :: ucosm(*), vcosm(*)
            integer(c_int) :: rc
        end function
        function spdy_vdspec(plan, nb, ug, vg, vorm, divm, kcos) bind(C, name="spdy_vdspec") result(rc)
            import :: c_int, c_ptr, c_double, c_double_complex
            type(c_ptr), value :: plan
            integer(c_int), value :: nb, kcos
            real(c_double), intent(in) :: ug(*), vg(*)
            complex(c_double_complex), intent(out) :: vorm(*), divm(*)
            integer(c_int) :: rc
        end function
        function spdy_uvspec_to_grid(plan, nb, vorm, divm, ug, vg, kcos) bind(C, name="spdy_uvspec_to_grid") result(rc)
            import :: c_ptr, c_int, c_double, c_double_complex
            type(c_ptr), value :: plan
            integer(c_int), value :: nb, kcos
            complex(c_double_complex), intent(in) :: vorm(*), divm(*)
            real(c_double), intent(out) :: ug(*), vg(*)
            integer(c_int) :: rc
        end function
        function spdy_grad_to_grid(plan, nb, psi, gx, gy, kcos) bind(C, name="spdy_grad_to_grid") result(rc)
            import :: c_ptr, c_int, c_double, c_double_complex
            type(c_ptr), value :: plan
            integer(c_int), value :: nb, kcos
            complex(c_double_complex), intent(in) :: psi(*)
            real(c_double), intent(out) :: gx(*), gy(*)
            integer(c_int) :: rc
        end function
        function spdy_hdiff(plan, nlev, field, fdt_in, dmp, dmp1, fdt_out) bind(C, name="spdy_hdiff") result(rc)
            import :: c_int, c_ptr, c_double, c_double_complex
            type(c_ptr), value :: plan
            integer(c_int), value :: nlev
            complex(c_double_complex), intent(in) :: field(*), fdt_in(*)
            real(c_double), intent(in) :: dmp(*), dmp1(*)
            complex(c_double_complex), intent(out) :: fdt_out(*)
            integer(c_int) :: rc
        end function
        function spdy_implicit_init(plan, dt) bind(C, name="spdy_implicit_init") result(rc)
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: plan
            real(c_double), value :: dt
            integer(c_int) :: rc
        end function
        function spdy_implicit_terms(plan, divdt, tdt, psdt) bind(C, name="spdy_implicit_terms") result(rc)
            import :: c_int, c_ptr, c_double_complex
            type(c_ptr), value :: plan
            complex(c_double_complex), intent(inout) :: divdt(*), tdt(*), psdt(*)
            integer(c_int) :: rc
        end function
        function spdy_plan_set_sigma(plan, hsg) bind(C, name="spdy_plan_set_sigma") result(rc)
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: plan
            real(c_double), intent(in) :: hsg(*)
            integer(c_int) :: rc
        end function
        function spdy_geopotential(plan, t, phis, phi) bind(C, name="spdy_geopotential") result(rc)
            import :: c_int, c_ptr, c_double_complex
            type(c_ptr), value :: plan
            complex(c_double_complex), intent(in) :: t(*), phis(*)
            complex(c_double_complex), intent(out) :: phi(*)
            integer(c_int) :: rc
        end function
        function spdy_step_field(plan, nlev, j1, dt, eps, wil, field, fdt) bind(C, name="spdy_step_field") result(rc)
            import :: c_int, c_ptr, c_double, c_double_complex
            type(c_ptr), value :: plan
            integer(c_int), value :: nlev, j1
            real(c_double), value :: dt, eps, wil
            complex(c_double_complex), intent(inout) :: field(*), fdt(*)
            integer(c_int) :: rc
        end function
        ! ---- multi-GPU (one process per GPU): level all-gather over RCCL for the level-sharded implicit solve
        function spdy_comm_unique_id(id) bind(C, name="spdy_comm_unique_id") result(rc)
            import :: c_int, c_char
            character(kind=c_char), intent(out) :: id(128)
            integer(c_int) :: rc
        end function
        function spdy_comm_create(plan, nranks, rank, id, comm) bind(C, name="spdy_comm_create") result(rc)
            import :: c_int, c_ptr, c_char
            type(c_ptr), value :: plan
            integer(c_int), value :: nranks, rank
            character(kind=c_char), intent(in) :: id(128)
            type(c_ptr), intent(out) :: comm
            integer(c_int) :: rc
        end function
        function spdy_comm_destroy(comm) bind(C, name="spdy_comm_destroy") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: comm
            integer(c_int) :: rc
        end function
        function spdy_comm_level_range(comm, nlev, lo, hi) bind(C, name="spdy_comm_level_range") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: comm
            integer(c_int), value :: nlev
            integer(c_int), intent(out) :: lo, hi
            integer(c_int) :: rc
        end function
        ! device pointers (type(c_ptr) values obtained from the host's own HIP allocations)
        function spdy_implicit_terms_sharded_dev(comm, divdt, tdt, psdt) bind(C, name="spdy_implicit_terms_sharded_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: comm, divdt, tdt, psdt
            integer(c_int) :: rc
        end function
        ! ---- device-resident state: memory, the step's entry points, graph capture (time_stepping.f90) ----------
        function spdy_plan_synchronize(plan) bind(C, name="spdy_plan_synchronize") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int) :: rc
        end function
        function spdy_dev_alloc(plan, bytes, d_ptr) bind(C, name="spdy_dev_alloc") result(rc)
            import :: c_int, c_ptr, c_size_t
            type(c_ptr), value :: plan
            integer(c_size_t), value :: bytes
            type(c_ptr), intent(out) :: d_ptr
            integer(c_int) :: rc
        end function
        function spdy_dev_free(plan, d_ptr) bind(C, name="spdy_dev_free") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, d_ptr
            integer(c_int) :: rc
        end function
        ! host side as type(*): any contiguous array (complex or real) goes up or comes down as bytes
        function spdy_dev_upload(plan, d_dst, src, bytes) bind(C, name="spdy_dev_upload") result(rc)
            import :: c_int, c_ptr, c_size_t
            type(c_ptr), value :: plan, d_dst
            type(*), intent(in) :: src(*)
            integer(c_size_t), value :: bytes
            integer(c_int) :: rc
        end function
        function spdy_dev_download(plan, dst, d_src, bytes) bind(C, name="spdy_dev_download") result(rc)
            import :: c_int, c_ptr, c_size_t
            type(c_ptr), value :: plan, d_src
            type(*) :: dst(*)
            integer(c_size_t), value :: bytes
            integer(c_int) :: rc
        end function
        function spdy_graph_begin(plan) bind(C, name="spdy_graph_begin") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int) :: rc
        end function
        function spdy_graph_end(plan, graph) bind(C, name="spdy_graph_end") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            type(c_ptr), intent(out) :: graph
            integer(c_int) :: rc
        end function
        function spdy_graph_launch(graph) bind(C, name="spdy_graph_launch") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: graph
            integer(c_int) :: rc
        end function
        function spdy_comm_set_option(comm, name, value) bind(C, name="spdy_comm_set_option") result(rc)
            import :: c_char, c_int, c_ptr
            type(c_ptr), value :: comm
            character(kind=c_char), intent(in) :: name(*)     ! NUL-terminated
            integer(c_int), value :: value
            integer(c_int) :: rc
        end function
        function spdy_sharded_state_gather_dev(comm, vor, div, t, tr, ps) bind(C, name="spdy_sharded_state_gather_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: comm, vor, div, t, tr, ps
            integer(c_int) :: rc
        end function
        function spdy_sharded_gather_ranges_dev(comm, narr, arrays, nrows) bind(C, name="spdy_sharded_gather_ranges_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: comm
            integer(c_int), value :: narr
            type(c_ptr), intent(in) :: arrays(*)
            integer(c_int), intent(in) :: nrows(*)
            integer(c_int) :: rc
        end function
        function spdy_comm_describe(comm, buf, len) bind(C, name="spdy_comm_describe") result(rc)
            import :: c_char, c_int, c_ptr
            type(c_ptr), value :: comm
            character(kind=c_char), intent(out) :: buf(*)
            integer(c_int), value :: len
            integer(c_int) :: rc
        end function
        function spdy_graph_num_nodes(graph, nodes) bind(C, name="spdy_graph_num_nodes") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: graph
            integer(c_int), intent(out) :: nodes
            integer(c_int) :: rc
        end function
        function spdy_graph_destroy(graph) bind(C, name="spdy_graph_destroy") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: graph
            integer(c_int) :: rc
        end function
        ! every array argument below is a device pointer
        function spdy_inverse_batch_segs_dev(plan, npairs, d_vor, d_div, d_ug, d_vg, kcos_pairs, nseg, segs, d_kcos, kcos_all, &
                & d_grid, ngrad, d_psi, d_gx, d_gy, kcos_grad) bind(C, name="spdy_inverse_batch_segs_dev") result(rc)
            import :: c_int, c_ptr, spdy_spec_seg
            type(c_ptr), value :: plan, d_vor, d_div, d_ug, d_vg, d_kcos, d_grid, d_psi, d_gx, d_gy
            integer(c_int), value :: npairs, kcos_pairs, nseg, kcos_all, ngrad, kcos_grad
            type(spdy_spec_seg), intent(in) :: segs(*)
            integer(c_int) :: rc
        end function
        function spdy_grid_tendencies_dev(plan, ug, vg, tg, vorg, divg, trg, px, py, u_out, v_out, plain_out) &
                & bind(C, name="spdy_grid_tendencies_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, ug, vg, tg, vorg, divg, trg, px, py, u_out, v_out, plain_out
            integer(c_int) :: rc
        end function
        function spdy_direct_batch_spectral_step_dev(plan, d_ug, d_vg, d_grid, kcos, pvor, pdiv, pspec, vor, div, t, tr, ps, &
                & phis, d_tcorh, d_qcorh, sdrag, j1, dt, eps, wil, phi) bind(C, name="spdy_direct_batch_spectral_step_dev") result(rc)
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: plan, d_ug, d_vg, d_grid, pvor, pdiv, pspec, vor, div, t, tr, ps, phis, d_tcorh, d_qcorh, phi
            integer(c_int), value :: kcos, j1
            real(c_double), value :: sdrag, dt, eps, wil
            integer(c_int) :: rc
        end function
        ! ---- the rest of include/spdy.h, one interface per entry point (device pointers are type(c_ptr) values) ----
        function spdy_plan_set_stream(plan, hip_stream) bind(C, name="spdy_plan_set_stream") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, hip_stream
            integer(c_int) :: rc
        end function
        function spdy_plan_set_profiling(plan, on) bind(C, name="spdy_plan_set_profiling") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int), value :: on
            integer(c_int) :: rc
        end function
        function spdy_plan_set_fused(plan, mode) bind(C, name="spdy_plan_set_fused") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int), value :: mode
            integer(c_int) :: rc
        end function
        function spdy_plan_set_option(plan, name, value) bind(C, name="spdy_plan_set_option") result(rc)
            import :: c_char, c_int, c_ptr
            type(c_ptr), value :: plan
            character(kind=c_char), intent(in) :: name(*)     ! NUL-terminated
            integer(c_int), value :: value
            integer(c_int) :: rc
        end function
        function spdy_plan_get_profile(plan, ms, launches) bind(C, name="spdy_plan_get_profile") result(rc)
            import :: c_double, c_int, c_ptr
            type(c_ptr), value :: plan
            real(c_double), intent(out) :: ms(*)
            integer(c_int), intent(out) :: launches(*)
            integer(c_int) :: rc
        end function
        function spdy_wave_placement(plan, simd_of_wave, violations) bind(C, name="spdy_wave_placement") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int), intent(out) :: simd_of_wave(8), violations
            integer(c_int) :: rc
        end function
        function spdy_plan_dims(plan, dims) bind(C, name="spdy_plan_dims") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int), intent(out) :: dims(*)
            integer(c_int) :: rc
        end function
        function spdy_spec_to_grid_dev(plan, nb, d_spec, d_kcos, kcos_all, d_grid) &
                & bind(C, name="spdy_spec_to_grid_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, d_spec, d_kcos, d_grid
            integer(c_int), value :: nb, kcos_all
            integer(c_int) :: rc
        end function
        function spdy_grid_to_spec_dev(plan, nb, d_grid, d_spec) bind(C, name="spdy_grid_to_spec_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, d_grid, d_spec
            integer(c_int), value :: nb
            integer(c_int) :: rc
        end function
        function spdy_legendre_inv(plan, nb, spec, four) bind(C, name="spdy_legendre_inv") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int), value :: nb
            type(*), intent(in) :: spec(*)
            type(*) :: four(*)
            integer(c_int) :: rc
        end function
        function spdy_legendre_dir(plan, nb, four, spec) bind(C, name="spdy_legendre_dir") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int), value :: nb
            type(*), intent(in) :: four(*)
            type(*) :: spec(*)
            integer(c_int) :: rc
        end function
        function spdy_fourier_inv(plan, nb, four, kcos, grid) bind(C, name="spdy_fourier_inv") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int), value :: nb, kcos
            type(*), intent(in) :: four(*)
            type(*) :: grid(*)
            integer(c_int) :: rc
        end function
        function spdy_fourier_dir(plan, nb, grid, four) bind(C, name="spdy_fourier_dir") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int), value :: nb
            type(*), intent(in) :: grid(*)
            type(*) :: four(*)
            integer(c_int) :: rc
        end function
        function spdy_laplacian_dev(plan, nb, in, out) bind(C, name="spdy_laplacian_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, in, out
            integer(c_int), value :: nb
            integer(c_int) :: rc
        end function
        function spdy_inverse_laplacian_dev(plan, nb, in, out) bind(C, name="spdy_inverse_laplacian_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, in, out
            integer(c_int), value :: nb
            integer(c_int) :: rc
        end function
        function spdy_trunct_dev(plan, nb, inout) bind(C, name="spdy_trunct_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, inout
            integer(c_int), value :: nb
            integer(c_int) :: rc
        end function
        function spdy_grad_dev(plan, nb, psi, psdx, psdy) bind(C, name="spdy_grad_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, psi, psdx, psdy
            integer(c_int), value :: nb
            integer(c_int) :: rc
        end function
        function spdy_vds_dev(plan, nb, ucosm, vcosm, vorm, divm) bind(C, name="spdy_vds_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, ucosm, vcosm, vorm, divm
            integer(c_int), value :: nb
            integer(c_int) :: rc
        end function
        function spdy_uvspec_dev(plan, nb, vorm, divm, ucosm, vcosm) bind(C, name="spdy_uvspec_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, vorm, divm, ucosm, vcosm
            integer(c_int), value :: nb
            integer(c_int) :: rc
        end function
        function spdy_vdspec_dev(plan, nb, ug, vg, vorm, divm, kcos) bind(C, name="spdy_vdspec_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, ug, vg, vorm, divm
            integer(c_int), value :: nb, kcos
            integer(c_int) :: rc
        end function
        function spdy_hdiff_dev(plan, nlev, field, fdt_in, d_dmp, d_dmp1, fdt_out) bind(C, name="spdy_hdiff_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, field, fdt_in, d_dmp, d_dmp1, fdt_out
            integer(c_int), value :: nlev
            integer(c_int) :: rc
        end function
        function spdy_hdiff_multi_dev(plan, nops, ops) bind(C, name="spdy_hdiff_multi_dev") result(rc)
            import :: c_int, c_ptr, spdy_hdiff_op
            type(c_ptr), value :: plan
            integer(c_int), value :: nops
            type(spdy_hdiff_op), intent(in) :: ops(*)
            integer(c_int) :: rc
        end function
        function spdy_implicit_terms_dev(plan, divdt, tdt, psdt) bind(C, name="spdy_implicit_terms_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, divdt, tdt, psdt
            integer(c_int) :: rc
        end function
        function spdy_device_table(plan, name, d_ptr) bind(C, name="spdy_device_table") result(rc)
            import :: c_char, c_int, c_ptr
            type(c_ptr), value :: plan
            character(kind=c_char), intent(in) :: name(*)
            type(c_ptr), intent(out) :: d_ptr
            integer(c_int) :: rc
        end function
        function spdy_geopotential_dev(plan, t, phis, phi) bind(C, name="spdy_geopotential_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, t, phis, phi
            integer(c_int) :: rc
        end function
        function spdy_spectral_tendencies_dev(plan, div, t, ps, phis, divdt, tdt, psdt, phi) &
                & bind(C, name="spdy_spectral_tendencies_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, div, t, ps, phis, divdt, tdt, psdt, phi
            integer(c_int) :: rc
        end function
        function spdy_hdiff_step_dev(plan, vor, div, t, tr, d_tcorh, d_qcorh, sdrag, vordt, divdt, tdt, trdt) &
                & bind(C, name="spdy_hdiff_step_dev") result(rc)
            import :: c_double, c_int, c_ptr
            type(c_ptr), value :: plan, vor, div, t, tr, d_tcorh, d_qcorh, vordt, divdt, tdt, trdt
            real(c_double), value :: sdrag
            integer(c_int) :: rc
        end function
        function spdy_step_fields_dev(plan, nops, ops, j1, dt, eps, wil) bind(C, name="spdy_step_fields_dev") result(rc)
            import :: c_double, c_int, c_ptr, spdy_step_op
            type(c_ptr), value :: plan
            integer(c_int), value :: nops, j1
            type(spdy_step_op), intent(in) :: ops(*)
            real(c_double), value :: dt, eps, wil
            integer(c_int) :: rc
        end function
        function spdy_tendency_combine_dev(plan, pdiv, pspec) bind(C, name="spdy_tendency_combine_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, pdiv, pspec
            integer(c_int) :: rc
        end function
        function spdy_spectral_step_dev(plan, pvor, pdiv, pspec, vor, div, t, tr, ps, phis, d_tcorh, d_qcorh, sdrag, j1, &
                & dt, eps, wil, phi) &
                & bind(C, name="spdy_spectral_step_dev") result(rc)
            import :: c_double, c_int, c_ptr
            type(c_ptr), value :: plan, pvor, pdiv, pspec, vor, div, t, tr, ps, phis, d_tcorh, d_qcorh, phi
            real(c_double), value :: sdrag, dt, eps, wil
            integer(c_int), value :: j1
            integer(c_int) :: rc
        end function
        function spdy_allgather_levels_dev(comm, nlev, narr, d_full) bind(C, name="spdy_allgather_levels_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: comm
            integer(c_int), value :: nlev, narr
            type(c_ptr), intent(in) :: d_full(*)
            integer(c_int) :: rc
        end function
        ! ranks inside one process (one host thread + one plan per GPU) and the complete level-sharded step
        function spdy_comm_group_create(nranks, group) bind(C, name="spdy_comm_group_create") result(rc)
            import :: c_int, c_ptr
            integer(c_int), value :: nranks
            type(c_ptr), intent(out) :: group
            integer(c_int) :: rc
        end function
        function spdy_comm_group_destroy(group) bind(C, name="spdy_comm_group_destroy") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: group
            integer(c_int) :: rc
        end function
        function spdy_comm_create_local(plan, group, rank, comm) bind(C, name="spdy_comm_create_local") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, group
            integer(c_int), value :: rank
            type(c_ptr), intent(out) :: comm
            integer(c_int) :: rc
        end function
        function spdy_sharded_step_workspace(comm) bind(C, name="spdy_sharded_step_workspace") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: comm
            integer(c_int) :: rc
        end function
        function spdy_sharded_step_dev(comm, vor, div, t, tr, ps, phis, d_tcorh, d_qcorh, sdrag, j1, j2, dt, eps, wil, phi, tend_out) &
                & bind(C, name="spdy_sharded_step_dev") result(rc)
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: comm, vor, div, t, tr, ps, phis, d_tcorh, d_qcorh, phi, tend_out
            real(c_double), value :: sdrag, dt, eps, wil
            integer(c_int), value :: j1, j2
            integer(c_int) :: rc
        end function
        function spdy_sharded_step_grid_dev(comm, vor, div, t, tr, ps, j2) bind(C, name="spdy_sharded_step_grid_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: comm, vor, div, t, tr, ps
            integer(c_int), value :: j2
            integer(c_int) :: rc
        end function
        function spdy_sharded_step_operands(comm, u, v, plain, lo, hi) bind(C, name="spdy_sharded_step_operands") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: comm
            type(c_ptr), intent(out) :: u, v, plain
            integer(c_int), intent(out) :: lo, hi
            integer(c_int) :: rc
        end function
        function spdy_sharded_step_spectral_dev(comm, vor, div, t, tr, ps, phis, d_tcorh, d_qcorh, sdrag, j1, dt, eps, wil, phi, &
                & tend_out) bind(C, name="spdy_sharded_step_spectral_dev") result(rc)
            import :: c_int, c_ptr, c_double
            type(c_ptr), value :: comm, vor, div, t, tr, ps, phis, d_tcorh, d_qcorh, phi, tend_out
            real(c_double), value :: sdrag, dt, eps, wil
            integer(c_int), value :: j1
            integer(c_int) :: rc
        end function
        function spdy_sharded_step_stacks(comm, grid_stack, grid_doubles, spec_stack, spec_doubles) &
                & bind(C, name="spdy_sharded_step_stacks") result(rc)
            import :: c_int, c_ptr, c_size_t
            type(c_ptr), value :: comm
            type(c_ptr), intent(out) :: grid_stack, spec_stack
            integer(c_size_t), intent(out) :: grid_doubles, spec_doubles
            integer(c_int) :: rc
        end function
        function spdy_uvspec_to_grid_dev(plan, nb, d_vor, d_div, d_ug, d_vg, kcos) &
                & bind(C, name="spdy_uvspec_to_grid_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, d_vor, d_div, d_ug, d_vg
            integer(c_int), value :: nb, kcos
            integer(c_int) :: rc
        end function
        function spdy_grad_to_grid_dev(plan, nb, d_psi, d_gx, d_gy, kcos) bind(C, name="spdy_grad_to_grid_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, d_psi, d_gx, d_gy
            integer(c_int), value :: nb, kcos
            integer(c_int) :: rc
        end function
        function spdy_inverse_batch_dev(plan, npairs, d_vor, d_div, d_ug, d_vg, kcos_pairs, nplain, d_spec, d_kcos, &
                & kcos_all, d_grid) &
                & bind(C, name="spdy_inverse_batch_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, d_vor, d_div, d_ug, d_vg, d_spec, d_kcos, d_grid
            integer(c_int), value :: npairs, kcos_pairs, nplain, kcos_all
            integer(c_int) :: rc
        end function
        function spdy_direct_batch_dev(plan, npairs, d_ug, d_vg, d_vorm, d_divm, kcos, nplain, d_grid, d_spec) &
                & bind(C, name="spdy_direct_batch_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, d_ug, d_vg, d_vorm, d_divm, d_grid, d_spec
            integer(c_int), value :: npairs, kcos, nplain
            integer(c_int) :: rc
        end function
        function spdy_inverse_batch_grad_dev(plan, npairs, d_vor, d_div, d_ug, d_vg, kcos_pairs, nplain, d_spec, d_kcos, &
                & kcos_all, d_grid, ngrad, d_psi, d_gx, d_gy, kcos_grad) &
                & bind(C, name="spdy_inverse_batch_grad_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, d_vor, d_div, d_ug, d_vg, d_spec, d_kcos, d_grid, d_psi, d_gx, d_gy
            integer(c_int), value :: npairs, kcos_pairs, nplain, kcos_all, ngrad, kcos_grad
            integer(c_int) :: rc
        end function
        function spdy_output_workspace(plan) bind(C, name="spdy_output_workspace") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan
            integer(c_int) :: rc
        end function
        function spdy_output_batch_dev(plan, vor, div, t, q, phi, ps, u_out, v_out, t_out, q_out, phi_out, ps_out) &
                & bind(C, name="spdy_output_batch_dev") result(rc)
            import :: c_int, c_ptr
            type(c_ptr), value :: plan, vor, div, t, q, phi, ps, u_out, v_out, t_out, q_out, phi_out, ps_out
            integer(c_int) :: rc
        end function
    end interface

    integer(c_int), parameter :: SPDY_DEVICE_AUTO = -2_c_int   !! $SPDY_DEVICE, else the launcher's local rank (include/spdy.h)

    !> Work a drop-in module has deferred on the device and that must be issued before the plan's tables change (time_stepping's
    !  collected leapfrog steps under steps_per_launch > 1): the module registers its flush routine here, and every wrapper that
    !  re-uploads tables (implicit%initialize_implicit) calls spdy_flush_pending first -- no circular module dependency.
    abstract interface
        subroutine spdy_flush_iface()
        end subroutine
    end interface
    procedure(spdy_flush_iface), pointer, save :: spdy_pending_flush => null()

contains
    subroutine spdy_flush_pending()
        if (associated(spdy_pending_flush)) call spdy_pending_flush()
    end subroutine

    !> The reference has no status returns (it `stop`s on fatal errors, e.g. matrix_inversion.f90:26);
    !  the drop-in keeps the signatures and turns a non-zero C status into `error stop`.
    subroutine spdy_check(rc, what)
        integer(c_int), intent(in) :: rc
        character(*), intent(in) :: what
        character(kind=c_char), pointer :: msg(:)
        type(c_ptr) :: cmsg
        integer :: n
        if (rc >= 0) return
        cmsg = spdy_last_error()
        call c_f_pointer(cmsg, msg, [512])
        n = 0
        do while (n < 512)
            if (msg(n + 1) == c_null_char) exit
            n = n + 1
        end do
        write (*, '(3A,I0,2A)') 'spdy: ', what, ' failed (', rc, '): ', transfer(msg(1:n), repeat(' ', n))
        error stop 'spdy: HIP spectral-transform path failed'
    end subroutine
end module
