!> Minimal stand-ins for the HOST MODEL's `types` and `params` modules, used only to build and
!  test the drop-in inside this repository (tests/test_fortran_dropin.py).  In a real integration
!  these two modules are the model's own (source/types.f90, source/params.f90) and this file is
!  not compiled.  Resolution is chosen with -DSPDY_T63 (default T30), like editing params.f90.
module types
    use iso_fortran_env, only: real32, real64
    implicit none
    integer, parameter :: sp = real32, dp = real64, p = dp      ! types.f90:10-12
end module

module params
    use iso_fortran_env, only: real64
    implicit none
#ifdef SPDY_T63
    integer, parameter :: trunc = 63, ix = 192, iy = 48
#else
    integer, parameter :: trunc = 30, ix = 96, iy = 24
#endif
    integer, parameter :: il = 2*iy, kx = 8, nx = trunc + 2, mx = trunc + 1
#ifdef SPDY_NSTEPS
    integer, parameter :: nsteps = SPDY_NSTEPS  ! (experiments: another step length)
#else
    integer, parameter :: nsteps = 36          ! params.f90:30 (read by initialize_horizontal_diffusion)
#endif
    ! read by time_stepping (params.f90:26, :31-33); default-real literals, as the model writes them
    integer, parameter :: ntr = 1
    real(real64), parameter :: delt = real(86400.0/nsteps, real64), rob = real(0.05, real64), wil = real(0.53, real64)
end module
