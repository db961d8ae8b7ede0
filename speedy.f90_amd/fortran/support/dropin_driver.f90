!> Test driver for the drop-in modules (spectral, horizontal_diffusion, implicit, geopotential): reads seeded inputs written by
!  tests/test_fortran_dropin.py, calls the reference-named API exactly as the model would, and
!  writes the results back for the test to compare.
program dropin_driver
    use types, only: p
    use params
    use spectral
    use horizontal_diffusion
    use implicit
    use geopotential
    implicit none
    complex(p) :: s(mx,nx,2), so(mx,nx), u(mx,nx), v(mx,nx), vor(mx,nx), div(mx,nx), dx(mx,nx), dy(mx,nx)
    complex(p) :: sk(mx,nx,kx), tk(mx,nx,kx), ps(mx,nx), hk(mx,nx,kx), slev(mx,nx,kx)
    real(p) :: g(ix,il,2), go(ix,il), dmp_in(mx,nx), dmp1_in(mx,nx), glev(ix,il,kx), glev2(ix,il,kx)
    complex(p) :: slev2(mx,nx,kx)
    integer :: kc(kx), k
    character(len=512) :: fin, fout

    call get_command_argument(1, fin)
    call get_command_argument(2, fout)
    open(10, file=trim(fin), access='stream', form='unformatted', status='old')
    read(10) s, g, sk, tk, ps, dmp_in, dmp1_in
    close(10)

    call initialize_spectral
    open(11, file=trim(fout), access='stream', form='unformatted', status='replace')
    write(11) el2
    go = spec_to_grid(s(:,:,1), 1);  write(11) go
    go = spec_to_grid(s(:,:,2), 2);  write(11) go
    so = grid_to_spec(g(:,:,1));     write(11) so
    so = laplacian(s(:,:,1));        write(11) so
    so = inverse_laplacian(s(:,:,1)); write(11) so
    so = s(:,:,1); call trunct(so);  write(11) so
    call grad(s(:,:,1), dx, dy);     write(11) dx, dy
    call uvspec(s(:,:,1), s(:,:,2), u, v); write(11) u, v
    call vds(s(:,:,1), s(:,:,2), vor, div); write(11) vor, div
    call vdspec(g(:,:,1), g(:,:,2), vor, div, 2); write(11) vor, div
    ! level stacks through the batched extension and the tail
    do k = 1, kx
        kc(k) = 1 + mod(k, 2)
    end do
    call spec_to_grid_levels(kx, sk, kc, glev); write(11) glev
    call grid_to_spec_levels(kx, glev, slev);   write(11) slev
    ! the reference's own init order (initialization.f90:19-20, time_stepping.f90:12-24) and calls
    call initialize_geopotential
    call initialize_horizontal_diffusion
    call initialize_implicit(real(4800, p))
    write(11) dmp, dmpd, dmps, dmp1, dmp1d, dmp1s, tcorv, qcorv, tref, tref2, tref3
    hk = do_horizontal_diffusion(tk, sk, dmp_in, dmp1_in); write(11) hk          ! 3-D specific
    so = do_horizontal_diffusion(ps, sk(:,:,1), dmps, dmp1s); write(11) so       ! 2-D specific, module tables
    hk = get_geopotential(tk, ps); write(11) hk
    call implicit_terms(sk, tk, ps); write(11) sk, tk, ps
    ! operator + transform sequences over a level stack (one call each)
    call uvspec_to_grid_levels(kx, sk, tk, glev, glev2); write(11) glev, glev2
    call grad_to_grid_levels(kx, sk, glev, glev2);       write(11) glev, glev2
    call vdspec_levels(kx, glev, glev2, slev, slev2, 2); write(11) slev, slev2
    close(11)
    call finalize_spectral
end program
