!> Rate of the HOST-pointer drop-in path as a Fortran host sees it (PCIe copies and stream syncs included): the stock call
!  pattern of the reference -- one field per call, function results (spectral.f90:98-122) -- and the level-stack extension.
!  Prints three numbers: round trips/s per-field, round trips/s with *_levels, fields per level stack.  Used by bench.py
!  (extras.host_pointer_dropin); never the headline value.
program dropin_rate
    use types, only: p
    use params
    use spectral
    implicit none
    real(p) :: g(ix,il,kx), g2(ix,il,kx)
    complex(p) :: s(mx,nx,kx)
    integer :: kc(kx), k, it, nit, i, j
    integer(8) :: c0, c1, rate
    real(8) :: t_field, t_lev
    character(len=32) :: arg

    nit = 50
    if (command_argument_count() >= 1) then
        call get_command_argument(1, arg)
        read(arg, *) nit
    end if
    do k = 1, kx
        do j = 1, il
            do i = 1, ix
                g(i,j,k) = sin(0.37_p*i + 0.11_p*j*k) + 0.01_p*k
            end do
        end do
        kc(k) = 1
    end do
    call initialize_spectral
    ! warm-up (plan creation, first-touch of the staging buffers)
    do k = 1, kx
        s(:,:,k) = grid_to_spec(g(:,:,k))
        g2(:,:,k) = spec_to_grid(s(:,:,k), 1)
    end do
    call grid_to_spec_levels(kx, g, s)
    call spec_to_grid_levels(kx, s, kc, g2)

    call system_clock(c0, rate)
    do it = 1, nit
        do k = 1, kx
            s(:,:,k) = grid_to_spec(g(:,:,k))
            g2(:,:,k) = spec_to_grid(s(:,:,k), 1)
        end do
    end do
    call system_clock(c1)
    t_field = real(c1 - c0, 8) / real(rate, 8)

    call system_clock(c0)
    do it = 1, nit
        call grid_to_spec_levels(kx, g, s)
        call spec_to_grid_levels(kx, s, kc, g2)
    end do
    call system_clock(c1)
    t_lev = real(c1 - c0, 8) / real(rate, 8)

    write(*, '(3(es14.6,1x),i4)') real(nit*kx, 8) / t_field, real(nit*kx, 8) / t_lev, sum(g2(:,:,1)) * 0d0, kx
    call finalize_spectral
end program
