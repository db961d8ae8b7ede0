!> Test driver for the level-sharded time step driven from ONE Fortran process: `nranks` OpenMP threads, each with its own
!  plan (on device `rank mod ndev`; ndev = $DROPIN_NDEV, default 1, so that it runs on a 1-GPU box), joined into an in-process
!  rank group (include/spdy.h: spdy_comm_group_create / spdy_comm_create_local).  Every rank holds the full prognostic state,
!  as the reference's prognostics module does (prognostics.f90:13-29); each leapfrog step is ONE call per rank,
!  spdy_sharded_step_dev = time_stepping.f90:35-118 with the transforms of a rank's own levels only and two level exchanges.
!  Reads the seeded state tests/test_fortran_dropin.py writes, runs `nleap` steps of step(2, 2, 2*delt), checks that all
!  ranks end with the same bytes and writes rank 0's state, geopotential and applied tendencies for the test to compare.
program dropin_step_sharded
    use, intrinsic :: iso_c_binding
    use omp_lib
    use types, only: p
    use params
    use dynamical_constants, only: tdrs
    use spdy_c
    implicit none
    integer(c_size_t), parameter :: spec_bytes = 16_c_size_t*mx*nx
    complex(p) :: vor(mx,nx,kx,2), div(mx,nx,kx,2), t(mx,nx,kx,2), tr(mx,nx,kx,2,ntr), ps(mx,nx,2), phis(mx,nx)
    complex(p) :: tcorh(mx,nx), qcorh(mx,nx)
    complex(p), allocatable :: res(:,:)          ! (everything a rank ends with, rank): compared across ranks below
    integer :: nranks, nleap, ndev, nres, r, stat
    type(c_ptr) :: group
    character(len=512) :: fin, fout, arg
    real(p) :: dt, sdrag

    call get_command_argument(1, fin)
    call get_command_argument(2, fout)
    call get_command_argument(3, arg); read (arg, *) nleap
    call get_command_argument(4, arg); read (arg, *) nranks
    ndev = 1
    call get_environment_variable('DROPIN_NDEV', arg, status=stat)
    if (stat == 0) read (arg, *) ndev
    open(10, file=trim(fin), access='stream', form='unformatted', status='old')
    read(10) vor, div, t, tr, ps, phis, tcorh, qcorh
    close(10)
    dt = 2*delt
    sdrag = 1.0/(tdrs*3600.0)                    ! time_stepping.f90:77
    nres = (4*2*kx + 2 + kx + 4*kx + 1)*mx*nx     ! prognostics (two time levels) | phi | vordt divdt tdt trdt psdt
    allocate(res(nres, nranks))
    call spdy_check(spdy_comm_group_create(int(nranks, c_int), group), 'comm_group_create')

    !$omp parallel num_threads(nranks) default(shared)
    call rank_main(omp_get_thread_num())
    !$omp end parallel

    call spdy_check(spdy_comm_group_destroy(group), 'comm_group_destroy')
    do r = 2, nranks
        if (any(transfer(res(:, r), 1_8, 2*nres) /= transfer(res(:, 1), 1_8, 2*nres))) then
            write (*, *) 'rank', r - 1, 'ended with a state that differs from rank 0'
            error stop 1
        end if
    end do
    open(11, file=trim(fout), access='stream', form='unformatted', status='replace')
    write(11) res(:, 1)
    close(11)
contains
    subroutine rank_main(rank)
        integer, intent(in) :: rank
        type(c_ptr) :: plan, comm, d_vor, d_div, d_t, d_tr, d_ps, d_phis, d_tcorh, d_qcorh, d_phi, d_tend
        integer :: i, o
        integer(c_int) :: lo, hi

        call spdy_check(spdy_plan_create(int(trunc, c_int), int(ix, c_int), int(iy, c_int), int(kx, c_int), int(4*kx + 4, c_int), &
            & int(mod(rank, ndev), c_int), plan), 'plan_create')
        call spdy_check(spdy_implicit_init(plan, real(dt, c_double)), 'implicit_init')
        call spdy_check(spdy_comm_create_local(plan, group, int(rank, c_int), comm), 'comm_create_local')
        call spdy_check(spdy_sharded_step_workspace(comm), 'sharded_step_workspace')
        call spdy_check(spdy_comm_level_range(comm, int(kx, c_int), lo, hi), 'comm_level_range')
        !$omp critical
        write (*, '(A,I3,A,I3,A,I3,A,I3)') 'rank', rank, ' of', nranks, ': transforms levels', lo + 1, ' ..', hi
        !$omp end critical
        call spdy_check(spdy_dev_alloc(plan, 2*kx*spec_bytes, d_vor), 'alloc'); call spdy_check(spdy_dev_alloc(plan, 2*kx*spec_bytes, d_div), 'alloc')
        call spdy_check(spdy_dev_alloc(plan, 2*kx*spec_bytes, d_t), 'alloc');   call spdy_check(spdy_dev_alloc(plan, 2*kx*spec_bytes, d_tr), 'alloc')
        call spdy_check(spdy_dev_alloc(plan, 2*spec_bytes, d_ps), 'alloc');     call spdy_check(spdy_dev_alloc(plan, spec_bytes, d_phis), 'alloc')
        call spdy_check(spdy_dev_alloc(plan, spec_bytes, d_tcorh), 'alloc');    call spdy_check(spdy_dev_alloc(plan, spec_bytes, d_qcorh), 'alloc')
        call spdy_check(spdy_dev_alloc(plan, kx*spec_bytes, d_phi), 'alloc');   call spdy_check(spdy_dev_alloc(plan, (4*kx + 1)*spec_bytes, d_tend), 'alloc')
        call spdy_check(spdy_dev_upload(plan, d_vor, vor, 2*kx*spec_bytes), 'upload vor')
        call spdy_check(spdy_dev_upload(plan, d_div, div, 2*kx*spec_bytes), 'upload div')
        call spdy_check(spdy_dev_upload(plan, d_t, t, 2*kx*spec_bytes), 'upload t')
        call spdy_check(spdy_dev_upload(plan, d_tr, tr, 2*kx*spec_bytes), 'upload tr')
        call spdy_check(spdy_dev_upload(plan, d_ps, ps, 2*spec_bytes), 'upload ps')
        call spdy_check(spdy_dev_upload(plan, d_phis, phis, spec_bytes), 'upload phis')
        call spdy_check(spdy_dev_upload(plan, d_tcorh, tcorh, spec_bytes), 'upload tcorh')
        call spdy_check(spdy_dev_upload(plan, d_qcorh, qcorh, spec_bytes), 'upload qcorh')
        do i = 1, nleap                          ! speedy.f90:41: call step(2, 2, 2*delt)
            call spdy_check(spdy_sharded_step_dev(comm, d_vor, d_div, d_t, d_tr, d_ps, d_phis, d_tcorh, d_qcorh, real(sdrag, c_double), &
                & 2_c_int, 2_c_int, real(dt, c_double), real(rob, c_double), real(wil, c_double), d_phi, d_tend), 'sharded_step_dev')
        end do
        ! the transposed form ($SPDY_SHARD_TRANSPOSE=1 when the communicator is created) leaves every array current on the rank's own
        ! coefficients only: make them whole before they are read back (no-ops in the all-gather form)
        call spdy_check(spdy_sharded_state_gather_dev(comm, d_vor, d_div, d_t, d_tr, d_ps), 'sharded_state_gather_dev')
        block
            type(c_ptr) :: arrs(2)
            integer(c_int) :: rows(2)
            arrs = (/ d_phi, d_tend /)
            rows = (/ int(kx, c_int), int(4*kx + 1, c_int) /)
            call spdy_check(spdy_sharded_gather_ranges_dev(comm, 2_c_int, arrs, rows), 'sharded_gather_ranges_dev')
        end block
        call spdy_check(spdy_plan_synchronize(plan), 'plan_synchronize')
        o = 0
        call fetch(plan, rank, o, d_vor, 2*kx); call fetch(plan, rank, o, d_div, 2*kx); call fetch(plan, rank, o, d_t, 2*kx)
        call fetch(plan, rank, o, d_tr, 2*kx);  call fetch(plan, rank, o, d_ps, 2)
        call fetch(plan, rank, o, d_phi, kx);   call fetch(plan, rank, o, d_tend, 4*kx + 1)
        call spdy_check(spdy_comm_destroy(comm), 'comm_destroy')
        call spdy_check(spdy_plan_destroy(plan), 'plan_destroy')        ! (frees the plan's spdy_dev_alloc buffers too)
    end subroutine

    subroutine fetch(plan, rank, o, d_src, nfields)
        type(c_ptr), intent(in) :: plan, d_src
        integer, intent(in) :: rank, nfields
        integer, intent(inout) :: o
        call spdy_check(spdy_dev_download(plan, res(o + 1:o + nfields*mx*nx, rank + 1), d_src, nfields*spec_bytes), 'download')
        o = o + nfields*mx*nx
    end subroutine
end program
