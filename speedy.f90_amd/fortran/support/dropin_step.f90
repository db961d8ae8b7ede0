!> Test driver for the `time_stepping` drop-in: reads a seeded model state written by tests/test_fortran_dropin.py, runs
!  the model's own start-up sequence (first_step) and `nleap` leapfrog steps exactly as the model's main loop would
!  (speedy.f90: call step(2, 2, 2*delt)), and writes the state after every step for the test to compare with the reference sequence.
program dropin_step
    use types, only: p, sp
    use params
    use spectral, only: initialize_spectral, finalize_spectral
    use horizontal_diffusion, only: initialize_horizontal_diffusion, tcorh, qcorh
    use geopotential, only: initialize_geopotential
    use prognostics
    use time_stepping
    use implicit, only: initialize_implicit
    implicit none
    complex(p) :: vordt(mx,nx,kx), divdt(mx,nx,kx), tdt(mx,nx,kx), psdt(mx,nx), trdt(mx,nx,kx,ntr)
    real(sp), dimension(ix,il,kx) :: u_out, v_out, t_out, q_out, phi_out
    real(sp) :: ps_out(ix,il)
    integer :: nleap, i
    real(p) :: dtt
    integer(8) :: c0, c1, cr
    logical :: return_now = .false.
    character(len=512) :: fin, fout, arg

    call get_command_argument(1, fin)
    if (trim(fin) == 'time') then        ! dropin_step time <nsteps>: leapfrog steps per second of this main loop
        call get_command_argument(2, arg)
        read (arg, *) nleap
        call isothermal_state
        call initialize_spectral
        call initialize_geopotential
        call initialize_horizontal_diffusion
        call first_step
        ! the step length a model of this resolution would run with: the host parameters keep the reference's T30 value
        ! (nsteps = 36: 40 minutes) at every resolution, which at T63 is advectively unstable once anything moves -- the physics
        ! hook's stand-in terms took the T63 loop out of the floating-point range after a few hundred steps (round 5)
        dtt = delt*min(1.0_p, 30.0_p/real(trunc, p))
        if (dtt /= delt) call initialize_implicit(2*dtt)
        do i = 1, 20
            call step(2, 2, 2*dtt)
        end do
        call prognostics_from_device
        call system_clock(c0, cr)
        do i = 1, nleap
            call step(2, 2, 2*dtt)
        end do
        call prognostics_from_device     ! waits for the queued steps (and brings the state back, as an output step would)
        call system_clock(c1)
        if (any(t /= t)) error stop 'dropin_step: the state did not stay finite'
        ! (the last field: a checksum of the final state -- the same for every $SPDY_STEPS_PER_LAUNCH)
        write (*, '(F12.2,I8,3I5,ES26.17)') real(nleap, 8)*real(cr, 8)/real(c1 - c0, 8), nleap, trunc, kx, ntr, &
            & sum(abs(real(t, 8))) + sum(abs(real(vor, 8))) + sum(abs(real(ps, 8))) + sum(abs(tr))
        call finalize_time_stepping
        call finalize_spectral
        return_now = .true.
    end if
    if (return_now) goto 99
    call get_command_argument(2, fout)
    call get_command_argument(3, arg)
    read (arg, *) nleap
    open(10, file=trim(fin), access='stream', form='unformatted', status='old')
    read(10) vor, div, t, tr, ps, phis, tcorh, qcorh
    close(10)

    call initialize_spectral
    call initialize_geopotential
    call initialize_horizontal_diffusion
    open(11, file=trim(fout), access='stream', form='unformatted', status='replace')
    call first_step                      ! forward half step, first leapfrog step (both written below as one record)
    call dump
    ! the gridded snapshot (input_output.f90:183-205) after the start-up sequence, straight from the device-resident prognostics
    ! (the seeded test state is not a balanced one: a few 40-minute leapfrog steps later it no longer fits float32)
    call output_fields_from_device(u_out, v_out, t_out, q_out, phi_out, ps_out)
    open(12, file=trim(fout)//'.snapshot', access='stream', form='unformatted', status='replace')
    write(12) u_out, v_out, t_out, q_out, phi_out, ps_out
    close(12)
    do i = 1, nleap
        call step(2, 2, 2*delt)
        call dump
    end do
    close(11)
    call finalize_time_stepping
    call finalize_spectral
99  continue
contains
    !> an atmosphere at rest over a flat surface: stays finite for any step count and step length (the step's cost does not
    !  depend on the values)
    subroutine isothermal_state
        ! (Round 5: the column is the semi-implicit scheme's own reference profile tref(k) = 288 max(0.2, sigma_k)^(R gamma / g),
        ! implicit.f90:62-67 -- NOT isothermal.  An isothermal 250 K column is warmer than tref aloft and linearly unstable under the
        ! semi-implicit scheme with 40-minute steps; at exact rest nothing seeds the instability, but the physics hook's stand-in
        ! terms did: the T63 loop left the floating-point range after 30-50 steps, the T30 loop after ~200.)
        real(p), parameter :: hsg(9) = [0.0_p, 0.05_p, 0.14_p, 0.26_p, 0.42_p, 0.60_p, 0.77_p, 0.90_p, 1.0_p]
        real(p), parameter :: rgam = (2.0_p/7.0_p)*1004.0_p*6.0_p/(1000.0_p*9.81_p)
        integer :: k
        vor = 0; div = 0; t = 0; tr = 0; ps = 0; phis = 0; tcorh = 0; qcorh = 0
        if (kx /= 8) error stop 'dropin_step: the timing state is written for the 8-level sigma set'
        do k = 1, kx
            t(1,1,k,:) = 288.0_p*max(0.2_p, 0.5_p*(hsg(k) + hsg(k+1)))**rgam*sqrt(2.0_p)
        end do
        ! ... and a tracer pattern on it: a passive field (the winds stay zero), so the run stays finite for any length, but
        ! diffusion changes it every step -- the checksum printed by the timing mode then tells a skipped or repeated step
        ! from a correct run
        tr(1,1,:,:,1) = 1.0e-2_p
        tr(3,4,:,:,1) = (1.0e-3_p, 2.0e-3_p)
        tr(7,12,:,:,1) = (-2.0e-3_p, 0.5e-3_p)
    end subroutine

    subroutine dump
        character(len=8) :: env
        integer :: stat
        ! DROPIN_UNMODIFIED_HOST: read the host arrays as the reference's main loop does, without asking for them
        ! (they are current only if time_stepping%host_refresh_interval / $SPDY_HOST_REFRESH makes step() refresh them)
        call get_environment_variable('DROPIN_UNMODIFIED_HOST', env, status=stat)
        if (stat /= 0) call prognostics_from_device
        call tendencies_from_device(vordt, divdt, tdt, psdt, trdt)
        write(11) vor, div, t, tr, ps, phi, vordt, divdt, tdt, trdt, psdt
    end subroutine
end program
