// Host-side table generation for the spectral transform plan (product code).
//
// Reproduces, value for value, the tables the reference builds once at start-up:
//   geometry.f90:35-89, fftpack.f90:1-67 (rffti1), legendre.f90:23-71,158-237,
//   spectral.f90:20-82, horizontal_diffusion.f90:36-82, implicit.f90:36-165.
// The reference is FP64 in storage only; unsuffixed literals and float() are float32 first
// (SURVEY.md Appendix A).  Those sub-expressions are evaluated in float here too -- an
// "improved" table (exact pi, true Gaussian nodes, double 1/ix) breaks parity at 1e-8.
#pragma once
#include <string>
#include <vector>

namespace spdy {

struct HostTables {
    int trunc = 0, ix = 0, iy = 0, il = 0, kx = 0, nx = 0, mx = 0;
    // geometry
    std::vector<double> sia_half, coa_half, cosgr, cosgr2, hsg, dhs, fsg, dhsr, fsgr;
    std::vector<double> coriol;    // [il] 2*omega*sia (geometry.f90:89), j = 0 southernmost
    double rgas = 0.0, akap = 0.0, grav = 0.0; // physical_constants.f90:18-24 (float32 literals widened)
    // FFT
    std::vector<double> work;      // ix twiddle slots, FFTPACK layout
    int ifac[15] = {0};
    double fwd_scale = 0.0;        // float32(1/ix) widened  (fourier.f90:72)
    double taui = 0.0, sqrt2 = 0.0, hsqt2 = 0.0;   // float32 radix constants widened
    // Legendre
    std::vector<double> epsi, repsi, wt, poly;   // poly[m + mx*(n + nx*j)]
    std::vector<int> nsh2;
    // spectral operators
    std::vector<double> el2, elm2, el4, trfilt, gradx, gradym, gradyp, uvdx, uvdym, uvdyp, vddym, vddyp;
    // horizontal diffusion + implicit
    std::vector<double> dmp, dmpd, dmps, dmp1, dmp1d, dmp1s;
    std::vector<double> tref, tref1, tref2, tref3, xc, xd, xj, dhsx, elz;
    bool implicit_ready = false;
    double implicit_dt = 0.0;
    bool sigma_ready = false;      // hsg..fsgr hold a sigma-level set (geometry.f90:42-60 or set_sigma)
    // geopotential.f90:22-30 (valid when sigma_ready): xgeop1[kx], xgeop2[kx] (xgeop2[0] unused = 0),
    // and the lapse-rate correction factors corf[kx] of :53 (0 for the top and bottom level)
    std::vector<double> xgeop1, xgeop2, corf;
    // horizontal_diffusion.f90:70-82 (valid when sigma_ready): tcorv[kx], qcorv[kx]
    std::vector<double> tcorv, qcorv;

    // Builds everything except the dt-dependent implicit tables.  Returns "" or an error text.
    std::string build(int trunc, int ix, int iy, int kx);
    // Caller-supplied half levels hsg[kx+1] (e.g. a 16-level set: geometry.f90:42-48 only defines kx = 5, 7, 8);
    // derives dhs, fsg, dhsr, fsgr as geometry.f90:51-60 does and invalidates the implicit tables.
    std::string set_sigma(const double *hsg_in);
    // implicit.f90:36-165 (+ dmp1* of :50-56).  Returns "" or an error text.
    std::string build_implicit(double dt);
    // Named lookup for spdy_get_table; nullptr if unknown. *count receives the length.
    const double *lookup(const std::string &name, int *count, std::vector<double> &scratch) const;
};

}  // namespace spdy
