!> Drop-in replacement for the reference's `time_stepping` module (source/time_stepping.f90): same module name and the
!  same public subroutines, first_step and step(j1, j2, dt), for the ADIABATIC dynamical core -- everything step() does
!  except get_physical_tendencies (column physics, outside this library's scope, SURVEY s8).
!
!  The reference's step() (time_stepping.f90:35-118) walks the host arrays of `prognostics` through get_tendencies
!  (tendencies.f90:11-41: 6 kx + 2 inverse and 9 kx + 1 direct transforms, the grid-space tendencies, the spectral
!  tendencies, the implicit correction), the diffusion block and step_field_2d/3d.  Here the prognostics live in HBM and
!  the whole step is three calls on device pointers,
!      spdy_inverse_batch_segs_dev          everything that goes to the grid           tendencies.f90:89-107, :121-123
!      spdy_grid_tendencies_dev             the grid-space dynamical tendencies        tendencies.f90:105-197
!      spdy_direct_batch_spectral_step_dev  direct transforms and all the rest         tendencies.f90:212-293, implicit.f90:168-217,
!                                                                                      time_stepping.f90:62-167
!  4 kernel launches at T30, 6 at T63, no host arithmetic, no PCIe traffic -- issued as plain launches, or captured into a
!  graph of one or several steps and replayed (steps_per_launch below).  Results agree with the reference's call-by-call sequence to 1e-12
!  (tests/test_fortran_dropin.py, the time_stepping case).
!
!  With -DSPDY_WITH_PHYSICS the module is a drop-in for the FULL step(): between the grid-space dynamical tendencies and the
!  direct transforms it calls the model's own physics%get_physical_tendencies exactly as tendencies.f90:203-206 does -- phi of
!  time level 1, the level-1 prognostics and the four grid tendencies utend, vtend, ttend, trtend come down to the host arrays,
!  the physics adds to the tendencies, they go back up -- so the column physics stays the model's Fortran on the host and
!  everything spectral stays on the GPU.  A step is then plain launches with one host section (no graph).
!
!  The host arrays of `prognostics` are uploaded at the first step (or by prognostics_to_device, after the model changed
!  them) and are by default NOT kept current: call prognostics_from_device before reading vor, div, t, ps, tr, phi on the host
!  (output, diagnostics, restart files) -- or set host_refresh_interval / $SPDY_HOST_REFRESH (below) and leave the host's
!  main loop as it is.  tcorh/qcorh (horizontal_diffusion, filled by forcing.f90) and phis travel with
!  prognostics_to_device.
module time_stepping
    use iso_c_binding
    use types, only: p, sp
    use params
    use spdy_c
    use spectral, only: spectral_plan, initialize_spectral

    implicit none

    private
    public first_step, step
    ! extensions (not in the reference)
    public prognostics_to_device, prognostics_from_device, tendencies_from_device, finalize_time_stepping
    public output_fields_from_device, host_refresh_interval, steps_per_launch, flush_steps

    !> Coherent mode for an UNMODIFIED host.  The reference's main loop reads the host arrays of `prognostics` right after
    !  step() (speedy.f90:41 check_diagnostics, :44-50 output and the coupler); with the state in HBM those arrays go stale unless
    !  the host calls prognostics_from_device.  host_refresh_interval = N > 0 makes step() itself refresh them (both time levels
    !  and phi) after every N-th step: N = 1 is exactly the reference's semantics (at the price of 1.3 MB over PCIe
    !  and a synchronisation per step), N = nsteps_out is enough for a host that only looks at the state when it writes
    !  output.  0 (default): never -- the host calls prognostics_from_device where it needs the state.  The environment
    !  variable SPDY_HOST_REFRESH sets the initial value, so that not even a recompile of the host is needed.
    integer :: host_refresh_interval = -1
    integer :: steps_since_refresh = 0

    !> How the leapfrog steps reach the GPU.  steps_per_launch = 0 (default): plain launches, three calls (four or six kernels)
    !  per step on the plan's stream -- the host thread runs ahead of the GPU and the kernels of consecutive steps follow each
    !  other without a gap (T30 L8: 29.7 us per step, 33.7 k steps/s).  1: one captured graph of the step, launched once per
    !  step -- a graph launch has a start-up latency of its own (4-7 us here) that back-to-back replays do not hide (36.7 us per
    !  step), but the host thread issues one call per step instead of three.  K > 1: step() collects K identical leapfrog
    !  steps and sends them as ONE graph of K captured steps (the same kernels in the same order: the same bits; 29.7 us per
    !  step at K = 8 with an eighth of the launch calls) -- deferred: whatever reads or replaces the device state first launches
    !  what is pending (prognostics_from_device, output_fields_from_device, tendencies_from_device, prognostics_to_device, a
    !  start-up step, a change of dt); a host that changes the model's tables itself between two steps (initialize_implicit)
    !  calls flush_steps before.  K > 1 is ignored while host_refresh_interval is in use.  The environment variable
    !  SPDY_STEPS_PER_LAUNCH sets the initial value.
    integer :: steps_per_launch = -1
    integer :: pending_steps = 0
    integer :: pending_under = 0                      ! the steps_per_launch value the pending steps were collected under
    type(c_ptr) :: graph_multi = c_null_ptr           ! steps_per_launch captured steps
    integer :: graph_multi_steps = 0

    integer(c_size_t), parameter :: spec_bytes = 16_c_size_t*mx*nx, grid_bytes = 8_c_size_t*ix*il

    logical :: resident = .false.
    ! model state: vor, div, t, tr (mx,nx,kx,2); ps (mx,nx,2); phi (mx,nx,kx); phis, tcorh, qcorh (mx,nx)
    type(c_ptr) :: d_vor = c_null_ptr, d_div = c_null_ptr, d_t = c_null_ptr, d_tr = c_null_ptr, d_ps = c_null_ptr
    type(c_ptr) :: d_phi = c_null_ptr, d_phis = c_null_ptr, d_tcorh = c_null_ptr, d_qcorh = c_null_ptr
    ! one step's intermediates: grids of time level j2 (ug, vg [kx]; vorg | divg | tg | trg [4 kx]; px, py), the grid
    ! tendencies as operands of the direct batch (U, V [3 kx]; PL [3 kx + 1]) and their spectra / the spectral tendencies
    type(c_ptr) :: d_ug = c_null_ptr, d_vg = c_null_ptr, d_plain = c_null_ptr, d_px = c_null_ptr, d_py = c_null_ptr
    type(c_ptr) :: d_u = c_null_ptr, d_v = c_null_ptr, d_pl = c_null_ptr
    type(c_ptr) :: d_pvor = c_null_ptr, d_pdiv = c_null_ptr, d_pspec = c_null_ptr
    type(c_ptr) :: d_out = c_null_ptr                 ! float32 snapshot fields: u, v, t, q, phi (ix,il,kx) | ps (ix,il)
    ! the captured leapfrog step and what it was captured for
    type(c_ptr) :: graph = c_null_ptr
    integer :: graph_j1 = 0, graph_j2 = 0
    real(p) :: graph_dt = 0

contains
    !> time_stepping.f90:11-24
    subroutine first_step
        use implicit, only: initialize_implicit

        call initialize_implicit(0.5*delt)

        call step(1, 1, 0.5*delt)

        call initialize_implicit(delt)

        call step(1, 2, delt)

        call initialize_implicit(2*delt)
    end subroutine

    !> time_stepping.f90:35-118 without the physics: Fnew = F(1) + dt * T_dyn(F(j2)); F(1) = (1-2 eps) F(j1) + eps (F(1) + Fnew)
    !  (with the Williams correction of :159-163); F(2) = Fnew.  j1 == 1: eps = 0; j1 == 2: eps = rob.
    subroutine step(j1, j2, dt)
        integer, intent(in) :: j1, j2
        real(p), intent(in) :: dt
        real(p) :: eps

        integer :: i

        if (j1 < 1 .or. j1 > 2 .or. j2 < 1 .or. j2 > 2) error stop 'time_stepping%step: j1, j2 must be 1 or 2'
        if (.not. resident) call prognostics_to_device
        call read_launch_policy
        if (j1 == 1) then
            eps = 0.0
        else
            eps = rob
        end if

#ifdef SPDY_WITH_PHYSICS
        call enqueue_to_grid(j2)
        call host_physics
        call enqueue_from_grid(j1, dt, eps)
#else
        if (j1 /= 2 .or. steps_per_launch == 0) then   ! the two start-up steps run once: plain launches (steps_per_launch = 0: every step)
            call flush_steps
            call enqueue_step(j1, j2, dt, eps)
        else
            if (c_associated(graph) .and. (graph_j1 /= j1 .or. graph_j2 /= j2 .or. graph_dt /= dt)) then
                call flush_steps
                call spdy_check(spdy_graph_destroy(graph), 'graph_destroy')
                graph = c_null_ptr
                if (c_associated(graph_multi)) call spdy_check(spdy_graph_destroy(graph_multi), 'graph_destroy')
                graph_multi = c_null_ptr
            end if
            if (.not. c_associated(graph)) then
                ! The implicit and damping tables are refreshed in place by initialize_implicit: the captured step sees them.
                call spdy_check(spdy_graph_begin(spectral_plan), 'graph_begin')
                call enqueue_step(j1, j2, dt, eps)
                call spdy_check(spdy_graph_end(spectral_plan, graph), 'graph_end')
                graph_j1 = j1; graph_j2 = j2; graph_dt = dt
            end if
            ! a host that changes steps_per_launch between steps: what was collected under the old value goes out first, in order
            if (pending_steps > 0 .and. steps_per_launch /= pending_under) call flush_steps
            if (steps_per_launch > 1 .and. host_refresh_interval <= 0) then
                spdy_pending_flush => flush_steps      ! (table re-uploads of the other drop-in modules flush through this hook)
                pending_under = steps_per_launch
                pending_steps = pending_steps + 1
                if (pending_steps >= steps_per_launch) then
                    if (c_associated(graph_multi) .and. graph_multi_steps /= steps_per_launch) then
                        call spdy_check(spdy_graph_destroy(graph_multi), 'graph_destroy')
                        graph_multi = c_null_ptr
                    end if
                    if (.not. c_associated(graph_multi)) then
                        call spdy_check(spdy_graph_begin(spectral_plan), 'graph_begin')
                        do i = 1, steps_per_launch
                            call enqueue_step(j1, j2, dt, eps)
                        end do
                        call spdy_check(spdy_graph_end(spectral_plan, graph_multi), 'graph_end')
                        graph_multi_steps = steps_per_launch
                    end if
                    call spdy_check(spdy_graph_launch(graph_multi), 'graph_launch')
                    pending_steps = 0
                end if
            else
                call spdy_check(spdy_graph_launch(graph), 'graph_launch')
            end if
        end if
#endif
        call refresh_host_if_due(j1)
    end subroutine

    !> Launch the leapfrog steps step() has collected but not sent yet (steps_per_launch), one by one.
    subroutine flush_steps
        integer :: i
        do i = 1, pending_steps
            call spdy_check(spdy_graph_launch(graph), 'graph_launch')
        end do
        pending_steps = 0
    end subroutine

    subroutine read_launch_policy
        character(len=16) :: env
        integer :: stat, n
        if (steps_per_launch >= 0) return
        steps_per_launch = 0
        call get_environment_variable('SPDY_STEPS_PER_LAUNCH', env, status=stat)
        if (stat == 0) then
            read (env, *, iostat=stat) n
            if (stat == 0 .and. n >= 0) steps_per_launch = n
        end if
    end subroutine

    !> host_refresh_interval (above): the host arrays follow the device state every N-th leapfrog step.
    subroutine refresh_host_if_due(j1)
        integer, intent(in) :: j1          ! (unused: start-up steps count like leapfrog steps)
        character(len=16) :: env
        integer :: stat, n
        if (host_refresh_interval < 0) then
            host_refresh_interval = 0
            call get_environment_variable('SPDY_HOST_REFRESH', env, status=stat)
            if (stat == 0) then
                read (env, *, iostat=stat) n
                if (stat == 0 .and. n > 0) host_refresh_interval = n
            end if
        end if
        if (host_refresh_interval <= 0) return
        steps_since_refresh = steps_since_refresh + 1
        if (steps_since_refresh >= host_refresh_interval) then
            call prognostics_from_device
            steps_since_refresh = 0
        end if
    end subroutine

    !> One adiabatic step on the plan's stream (returns when it is queued).
    subroutine enqueue_step(j1, j2, dt, eps)
        integer, intent(in) :: j1, j2
        real(p), intent(in) :: dt, eps
        call enqueue_to_grid(j2)
        call enqueue_from_grid(j1, dt, eps)
    end subroutine

    !> First half: everything of time level j2 to the grid and the grid-space dynamical tendencies.
    subroutine enqueue_to_grid(j2)
        integer, intent(in) :: j2
        type(spdy_spec_seg) :: segs(4)
        integer(c_size_t) :: lev3, lev2

        lev3 = (j2 - 1)*kx*spec_bytes            ! time level j2 of an (mx,nx,kx,2) array
        lev2 = (j2 - 1)*spec_bytes               ! ... of ps
        segs(1) = spdy_spec_seg(int(kx, c_int), at(d_vor, lev3))
        segs(2) = spdy_spec_seg(int(kx, c_int), at(d_div, lev3))
        segs(3) = spdy_spec_seg(int(kx, c_int), at(d_t, lev3))
        segs(4) = spdy_spec_seg(int(kx, c_int), at(d_tr, lev3))
        ! tendencies.f90:89-107, :121-123: u, v (kcos = 2) from vor, div; vorg, divg, tg, trg (kcos = 1); grad(ps) (kcos = 2)
        call spdy_check(spdy_inverse_batch_segs_dev(spectral_plan, int(kx, c_int), at(d_vor, lev3), at(d_div, lev3), d_ug, d_vg, &
            & 2_c_int, 4_c_int, segs, c_null_ptr, 1_c_int, d_plain, 1_c_int, at(d_ps, lev2), d_px, d_py, 2_c_int), &
            & 'inverse_batch_segs_dev')
        call spdy_check(spdy_grid_tendencies_dev(spectral_plan, d_ug, d_vg, at(d_plain, 2*kx*grid_bytes), d_plain, &
            & at(d_plain, kx*grid_bytes), at(d_plain, 3*kx*grid_bytes), d_px, d_py, d_u, d_v, d_pl), 'grid_tendencies_dev')
    end subroutine

    !> Second half: direct transforms, spectral tendencies, implicit correction, diffusion, leapfrog.
    subroutine enqueue_from_grid(j1, dt, eps)
        use dynamical_constants, only: tdrs
        integer, intent(in) :: j1
        real(p), intent(in) :: dt, eps
        real(p) :: sdrag

        sdrag = 1.0/(tdrs*3600.0)                ! time_stepping.f90:77
        call spdy_check(spdy_direct_batch_spectral_step_dev(spectral_plan, d_u, d_v, d_pl, 2_c_int, d_pvor, d_pdiv, d_pspec, &
            & d_vor, d_div, d_t, d_tr, d_ps, d_phis, d_tcorh, d_qcorh, sdrag, int(j1, c_int), dt, eps, wil, d_phi), &
            & 'direct_batch_spectral_step_dev')
    end subroutine

#ifdef SPDY_WITH_PHYSICS
    !> tendencies.f90:203-206 with the model's own physics on the host: phi = get_geopotential(t(:,:,:,1), phis), then
    !  get_physical_tendencies(vor(:,:,:,1), div(:,:,:,1), t(:,:,:,1), tr(:,:,:,1,1), phi, ps(:,:,1), utend, vtend, ttend, trtend)
    !  adds to the dynamical grid tendencies, which sit in the operands of the direct batch: utend = U(1:kx), vtend = V(1:kx),
    !  ttend = PL(kx+1:2kx), trtend = PL(2kx+1:3kx)  (include/spdy.h, spdy_grid_tendencies_dev).
    subroutine host_physics
        use prognostics, only: vor, div, t, ps, tr, phi
        use physics, only: get_physical_tendencies
        real(p), allocatable, save :: utend(:,:,:), vtend(:,:,:), ttend(:,:,:), trtend(:,:,:)

        if (.not. allocated(utend)) allocate(utend(ix,il,kx), vtend(ix,il,kx), ttend(ix,il,kx), trtend(ix,il,kx))
        call spdy_check(spdy_geopotential_dev(spectral_plan, d_t, d_phis, d_phi), 'geopotential_dev')
        ! time level 1 is the leading half of every prognostic array
        call spdy_check(spdy_dev_download(spectral_plan, vor, d_vor, kx*spec_bytes), 'download vor(1)')
        call spdy_check(spdy_dev_download(spectral_plan, div, d_div, kx*spec_bytes), 'download div(1)')
        call spdy_check(spdy_dev_download(spectral_plan, t, d_t, kx*spec_bytes), 'download t(1)')
        call spdy_check(spdy_dev_download(spectral_plan, tr, d_tr, kx*spec_bytes), 'download tr(1)')
        call spdy_check(spdy_dev_download(spectral_plan, ps, d_ps, spec_bytes), 'download ps(1)')
        call spdy_check(spdy_dev_download(spectral_plan, phi, d_phi, kx*spec_bytes), 'download phi')
        call spdy_check(spdy_dev_download(spectral_plan, utend, d_u, kx*grid_bytes), 'download utend')
        call spdy_check(spdy_dev_download(spectral_plan, vtend, d_v, kx*grid_bytes), 'download vtend')
        call spdy_check(spdy_dev_download(spectral_plan, ttend, at(d_pl, kx*grid_bytes), kx*grid_bytes), 'download ttend')
        call spdy_check(spdy_dev_download(spectral_plan, trtend, at(d_pl, 2*kx*grid_bytes), kx*grid_bytes), 'download trtend')
        call get_physical_tendencies(vor(:,:,:,1), div(:,:,:,1), t(:,:,:,1), tr(:,:,:,1,1), phi, ps(:,:,1), utend, vtend, ttend, trtend)
        call spdy_check(spdy_dev_upload(spectral_plan, d_u, utend, kx*grid_bytes), 'upload utend')
        call spdy_check(spdy_dev_upload(spectral_plan, d_v, vtend, kx*grid_bytes), 'upload vtend')
        call spdy_check(spdy_dev_upload(spectral_plan, at(d_pl, kx*grid_bytes), ttend, kx*grid_bytes), 'upload ttend')
        call spdy_check(spdy_dev_upload(spectral_plan, at(d_pl, 2*kx*grid_bytes), trtend, kx*grid_bytes), 'upload trtend')
    end subroutine
#endif

    !> Host arrays of `prognostics` (and phis, tcorh, qcorh) -> HBM; allocates the device state on first use.
    subroutine prognostics_to_device
        use prognostics, only: vor, div, t, ps, tr, phis
        use horizontal_diffusion, only: tcorh, qcorh

        if (ntr /= 1) error stop 'time_stepping: the device step carries one tracer (ntr = 1, params.f90:26)'
        call flush_steps                     ! (the pending steps belong to the state that is about to be replaced)
        call initialize_spectral
        if (.not. c_associated(d_vor)) then
            call alloc(d_vor, 2*kx*spec_bytes); call alloc(d_div, 2*kx*spec_bytes); call alloc(d_t, 2*kx*spec_bytes)
            call alloc(d_tr, 2*kx*spec_bytes);  call alloc(d_ps, 2*spec_bytes);     call alloc(d_phi, kx*spec_bytes)
            call alloc(d_phis, spec_bytes);     call alloc(d_tcorh, spec_bytes);    call alloc(d_qcorh, spec_bytes)
            call alloc(d_ug, kx*grid_bytes);    call alloc(d_vg, kx*grid_bytes);    call alloc(d_plain, 4*kx*grid_bytes)
            call alloc(d_px, grid_bytes);       call alloc(d_py, grid_bytes)
            call alloc(d_u, 3*kx*grid_bytes);   call alloc(d_v, 3*kx*grid_bytes);   call alloc(d_pl, (3*kx + 1)*grid_bytes)
            call alloc(d_pvor, 3*kx*spec_bytes); call alloc(d_pdiv, 3*kx*spec_bytes); call alloc(d_pspec, (3*kx + 1)*spec_bytes)
        end if
        call spdy_check(spdy_dev_upload(spectral_plan, d_vor, vor, 2*kx*spec_bytes), 'upload vor')
        call spdy_check(spdy_dev_upload(spectral_plan, d_div, div, 2*kx*spec_bytes), 'upload div')
        call spdy_check(spdy_dev_upload(spectral_plan, d_t, t, 2*kx*spec_bytes), 'upload t')
        call spdy_check(spdy_dev_upload(spectral_plan, d_tr, tr, 2*kx*spec_bytes), 'upload tr')
        call spdy_check(spdy_dev_upload(spectral_plan, d_ps, ps, 2*spec_bytes), 'upload ps')
        call spdy_check(spdy_dev_upload(spectral_plan, d_phis, phis, spec_bytes), 'upload phis')
        call spdy_check(spdy_dev_upload(spectral_plan, d_tcorh, tcorh, spec_bytes), 'upload tcorh')
        call spdy_check(spdy_dev_upload(spectral_plan, d_qcorh, qcorh, spec_bytes), 'upload qcorh')
        resident = .true.
    end subroutine

    !> HBM -> host arrays of `prognostics` (both time levels) and the geopotential of the last step; waits for the steps
    !  queued so far.
    subroutine prognostics_from_device
        use prognostics, only: vor, div, t, ps, tr, phi

        if (.not. resident) return
        call flush_steps
        call spdy_check(spdy_dev_download(spectral_plan, vor, d_vor, 2*kx*spec_bytes), 'download vor')
        call spdy_check(spdy_dev_download(spectral_plan, div, d_div, 2*kx*spec_bytes), 'download div')
        call spdy_check(spdy_dev_download(spectral_plan, t, d_t, 2*kx*spec_bytes), 'download t')
        call spdy_check(spdy_dev_download(spectral_plan, tr, d_tr, 2*kx*spec_bytes), 'download tr')
        call spdy_check(spdy_dev_download(spectral_plan, ps, d_ps, 2*spec_bytes), 'download ps')
        call spdy_check(spdy_dev_download(spectral_plan, phi, d_phi, kx*spec_bytes), 'download phi')
    end subroutine

    !> The tendencies the last step applied -- after the implicit correction and the diffusion, truncated as step_field_2d
    !  leaves them (time_stepping.f90:155-157): the local arrays of the reference's step().
    subroutine tendencies_from_device(vordt, divdt, tdt, psdt, trdt)
        complex(p), dimension(mx,nx,kx), intent(out) :: vordt, divdt, tdt
        complex(p), intent(out) :: psdt(mx,nx), trdt(mx,nx,kx,ntr)

        if (.not. resident) error stop 'time_stepping%tendencies_from_device before the first step'
        call flush_steps
        ! pvor = vordt | .. ; pdiv = divdt | tdt | trdt ; pspec(3 kx + 1) = psdt  (include/spdy.h, spdy_grid_tendencies_dev)
        call spdy_check(spdy_dev_download(spectral_plan, vordt, d_pvor, kx*spec_bytes), 'download vordt')
        call spdy_check(spdy_dev_download(spectral_plan, divdt, d_pdiv, kx*spec_bytes), 'download divdt')
        call spdy_check(spdy_dev_download(spectral_plan, tdt, at(d_pdiv, kx*spec_bytes), kx*spec_bytes), 'download tdt')
        call spdy_check(spdy_dev_download(spectral_plan, trdt, at(d_pdiv, 2*kx*spec_bytes), kx*spec_bytes), 'download trdt')
        call spdy_check(spdy_dev_download(spectral_plan, psdt, at(d_pspec, 3*kx*spec_bytes), spec_bytes), 'download psdt')
    end subroutine

    !> The gridded snapshot of input_output.f90:183-205 from the DEVICE-resident state: what the reference's subroutine output
    !  computes between its NetCDF calls -- uvspec + five spec_to_grid per level on time level 1 of vor, div, t, tr and on phi,
    !  spec_to_grid of ps, then real(., sp) of u, v, t, q*1.0e-3, phi/grav, p0*exp(ps) -- as ONE transform launch of 5 kx + 1
    !  fields and a float32 epilogue (spdy_output_batch_dev); only the six float32 arrays cross the link.  A host keeps its
    !  NetCDF writer and replaces the two computing blocks of `output` by this call (INTEGRATION.md).
    subroutine output_fields_from_device(u_out, v_out, t_out, q_out, phi_out, ps_out)
        real(sp), dimension(ix,il,kx), intent(out) :: u_out, v_out, t_out, q_out, phi_out
        real(sp), intent(out) :: ps_out(ix,il)
        integer(c_size_t), parameter :: fb = 4_c_size_t*ix*il

        if (.not. resident) error stop 'time_stepping%output_fields_from_device before the first step'
        call flush_steps
        if (.not. c_associated(d_out)) then
            call alloc(d_out, (5*kx + 1)*fb)
            call spdy_check(spdy_output_workspace(spectral_plan), 'output_workspace')
        end if
        ! time level 1 is the leading (mx,nx,kx) slab of every prognostic array
        call spdy_check(spdy_output_batch_dev(spectral_plan, d_vor, d_div, d_t, d_tr, d_phi, d_ps, &
            & d_out, at(d_out, kx*fb), at(d_out, 2*kx*fb), at(d_out, 3*kx*fb), at(d_out, 4*kx*fb), at(d_out, 5*kx*fb)), 'output_batch')
        call spdy_check(spdy_dev_download(spectral_plan, u_out, d_out, kx*fb), 'download u_out')
        call spdy_check(spdy_dev_download(spectral_plan, v_out, at(d_out, kx*fb), kx*fb), 'download v_out')
        call spdy_check(spdy_dev_download(spectral_plan, t_out, at(d_out, 2*kx*fb), kx*fb), 'download t_out')
        call spdy_check(spdy_dev_download(spectral_plan, q_out, at(d_out, 3*kx*fb), kx*fb), 'download q_out')
        call spdy_check(spdy_dev_download(spectral_plan, phi_out, at(d_out, 4*kx*fb), kx*fb), 'download phi_out')
        call spdy_check(spdy_dev_download(spectral_plan, ps_out, at(d_out, 5*kx*fb), fb), 'download ps_out')
    end subroutine

    !> Releases the captured step and the device state (before spectral%finalize_spectral).
    subroutine finalize_time_stepping
        call flush_steps
        if (c_associated(graph)) call spdy_check(spdy_graph_destroy(graph), 'graph_destroy')
        graph = c_null_ptr
        if (c_associated(graph_multi)) call spdy_check(spdy_graph_destroy(graph_multi), 'graph_destroy')
        graph_multi = c_null_ptr
        call release(d_vor); call release(d_div); call release(d_t); call release(d_tr); call release(d_ps); call release(d_phi)
        call release(d_phis); call release(d_tcorh); call release(d_qcorh)
        call release(d_ug); call release(d_vg); call release(d_plain); call release(d_px); call release(d_py)
        call release(d_u); call release(d_v); call release(d_pl); call release(d_pvor); call release(d_pdiv); call release(d_pspec)
        call release(d_out)
        resident = .false.
    end subroutine

    subroutine alloc(ptr, bytes)
        type(c_ptr), intent(out) :: ptr
        integer(c_size_t), intent(in) :: bytes
        call spdy_check(spdy_dev_alloc(spectral_plan, bytes, ptr), 'dev_alloc')
    end subroutine

    subroutine release(ptr)
        type(c_ptr), intent(inout) :: ptr
        if (c_associated(ptr)) call spdy_check(spdy_dev_free(spectral_plan, ptr), 'dev_free')
        ptr = c_null_ptr
    end subroutine

    !> device address `base + bytes`
    pure function at(base, bytes) result(ptr)
        type(c_ptr), intent(in) :: base
        integer(c_size_t), intent(in) :: bytes
        type(c_ptr) :: ptr
        ptr = transfer(transfer(base, 0_c_intptr_t) + int(bytes, c_intptr_t), ptr)
    end function
end module
