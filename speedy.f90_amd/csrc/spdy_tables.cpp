// Host-side table generation (see spdy_tables.hpp).  Compiled with -ffp-contract=off so the
// table values are the plain IEEE evaluation the reference's compiler produces.
#include "spdy_tables.hpp"

#include <cmath>
#include <cstdint>
#include <cstring>

namespace spdy {
namespace {

// physical_constants.f90:16-25, dynamical_constants.f90:12-22, params.f90:34 -- default-real
// literals, i.e. float32 values widened to double.
const double kRearth = static_cast<double>(6.371e+6f);
const double kGrav = static_cast<double>(9.81f);
const double kCp = static_cast<double>(1004.0f);
const double kGammaLapse = static_cast<double>(6.0f);
const double kThd = static_cast<double>(2.4f), kThdd = static_cast<double>(2.4f), kThds = static_cast<double>(12.0f);
const double kAlph = static_cast<double>(0.5f);
const double kOmega = static_cast<double>(7.292e-05f);
inline double kap32() { return static_cast<double>(2.0f / 7.0f); }

// sin(latitude) of the "Gaussian" rows is the only libm-cosf-dependent table
// (geometry.f90:68).  The values the reference produces (flang / glibc) are pinned here as
// IEEE binary32 bit patterns; a libm whose cosf differs in the last
// bit is overridden so the plan always carries the reference's nodes.
const uint32_t kSiaT30[24] = {
    0x3F7FB2AE, 0x3F7E5B64, 0x3F7BF2FD, 0x3F787C0F, 0x3F73FA50, 0x3F6E7299, 0x3F67EAD8, 0x3F606A13,
    0x3F57F855, 0x3F4E9EB3, 0x3F446734, 0x3F395CD2, 0x3F2D8B67, 0x3F20FFA4, 0x3F13C702, 0x3F05EFB2,
    0x3EEF1127, 0x3ED14231, 0x3EB2928D, 0x3E93232E, 0x3E662B96, 0x3E2519B2, 0x3DC6AD0D, 0x3D04A2C4};
const uint32_t kSiaT63[48] = {
    0x3F7FEC77, 0x3F7F95AC, 0x3F7EF989, 0x3F7E183A, 0x3F7CF1FB, 0x3F7B871D, 0x3F79D802, 0x3F77E51E,
    0x3F75AEFA, 0x3F73362E, 0x3F707B67, 0x3F6D7F61, 0x3F6A42ED, 0x3F66C6EB, 0x3F630C4C, 0x3F5F1415,
    0x3F5ADF59, 0x3F566F3B, 0x3F51C4F0, 0x3F4CE1BD, 0x3F47C6F4, 0x3F4275F7, 0x3F3CF039, 0x3F373739,
    0x3F314C85, 0x3F2B31B6, 0x3F24E876, 0x3F1E7278, 0x3F17D17D, 0x3F110753, 0x3F0A15D1, 0x3F02FED6,
    0x3EF788A6, 0x3EE8D079, 0x3ED9D922, 0x3ECAA6AD, 0x3EBB3D41, 0x3EABA109, 0x3E9BD641, 0x3E8BE132,
    0x3E778C60, 0x3E57132B, 0x3E365FA5, 0x3E157A9C, 0x3DE8DA2B, 0x3DA67FD2, 0x3D47F0BB, 0x3C855767};

inline float bits_to_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// Everything that is a function of the half levels hsg alone:
//   geometry.f90:51-60 (dhs, fsg, dhsr, fsgr), geopotential.f90:22-30 + :52-53 (xgeop1, xgeop2, corf),
//   horizontal_diffusion.f90:70-82 (tcorv, qcorv)
void make_sigma_derived(HostTables &t)
{
    const int kx = t.kx;
    for (int k = 0; k < kx; ++k) {
        t.dhs[k] = t.hsg[k + 1] - t.hsg[k];
        t.fsg[k] = 0.5 * (t.hsg[k + 1] + t.hsg[k]);
    }
    for (int k = 0; k < kx; ++k) {
        t.dhsr[k] = 0.5 / t.dhs[k];
        t.fsgr[k] = kap32() / (2.0 * t.fsg[k]);
    }
    const double rgas = kap32() * kCp;
    for (int k = 0; k < kx; ++k) {
        t.xgeop1[k] = rgas * std::log(t.hsg[k + 1] / t.fsg[k]);
        if (k != kx - 1) t.xgeop2[k + 1] = rgas * std::log(t.fsg[k + 1] / t.hsg[k + 1]);
    }
    for (int k = 0; k < kx; ++k) t.corf[k] = 0.0;
    for (int k = 1; k + 1 < kx; ++k)      // geopotential.f90:53 (left-to-right: ((xgeop1*0.5)*log)/log)
        t.corf[k] = t.xgeop1[k] * 0.5 * std::log(t.hsg[k + 1] / t.fsg[k]) / std::log(t.fsg[k + 1] / t.fsg[k - 1]);
    // horizontal_diffusion.f90:70-82: rgam = rgas*gamma/(1000.*grav), qexp = hscale/hshum (float32 literals widened)
    const double rgam = rgas * kGammaLapse / (1000.0 * kGrav);
    const double qexp = static_cast<double>(7.5f) / static_cast<double>(2.5f);
    for (int k = 0; k < kx; ++k) t.tcorv[k] = t.qcorv[k] = 0.0;
    for (int k = 1; k < kx; ++k) {
        t.tcorv[k] = std::pow(t.fsg[k], rgam);
        if (k > 1) t.qcorv[k] = std::pow(t.fsg[k], qexp);
    }
    t.sigma_ready = true;
}

// ---- geometry.f90:35-89 -----------------------------------------------------------------
void make_geometry(HostTables &t)
{
    static const float lev8[9] = {0.000f, 0.050f, 0.140f, 0.260f, 0.420f, 0.600f, 0.770f, 0.900f, 1.000f};
    static const float lev7[8] = {0.020f, 0.140f, 0.260f, 0.420f, 0.600f, 0.770f, 0.900f, 1.000f};
    static const float lev5[6] = {0.000f, 0.150f, 0.350f, 0.650f, 0.900f, 1.000f};
    const float *lev = t.kx == 8 ? lev8 : t.kx == 7 ? lev7 : t.kx == 5 ? lev5 : nullptr;
    t.hsg.assign(t.kx + 1, 0.0);
    t.dhs.assign(t.kx, 0.0); t.fsg.assign(t.kx, 0.0); t.dhsr.assign(t.kx, 0.0); t.fsgr.assign(t.kx, 0.0);
    t.xgeop1.assign(t.kx, 0.0); t.xgeop2.assign(t.kx, 0.0); t.corf.assign(t.kx, 0.0);
    t.tcorv.assign(t.kx, 0.0); t.qcorv.assign(t.kx, 0.0);
    t.sigma_ready = false;
    if (lev) {
        for (int k = 0; k <= t.kx; ++k) t.hsg[k] = static_cast<double>(lev[k]);
        make_sigma_derived(t);
    }
    t.sia_half.assign(t.iy, 0.0);
    t.coa_half.assign(t.il, 0.0);
    t.cosgr.assign(t.il, 0.0);
    t.cosgr2.assign(t.il, 0.0);
    const uint32_t *pinned = (t.iy == 24) ? kSiaT30 : (t.iy == 48) ? kSiaT63 : nullptr;
    for (int j = 1; j <= t.iy; ++j) {
        // whole expression is default real in the reference: float pi literal, float cos
        const float ang = 3.141592654f * (static_cast<float>(j) - 0.25f) / (static_cast<float>(t.il) + 0.5f);
        float s32 = cosf(ang);
        if (pinned) s32 = bits_to_float(pinned[j - 1]);
        const double s = static_cast<double>(s32);
        const double c = std::sqrt(1.0 - s * s);
        const int jn = t.il - j;   // 0-based mirror row
        t.sia_half[j - 1] = s;
        t.coa_half[j - 1] = c;     // reference only ever fills 1..iy of its (il) array
        t.cosgr[j - 1] = t.cosgr[jn] = 1.0 / c;
        t.cosgr2[j - 1] = t.cosgr2[jn] = 1.0 / (c * c);
    }
    // coriol = 2.0*omega*sia with sia(j) = -sia_half(j), sia(il+1-j) = +sia_half(j)   (geometry.f90:70-71, 89)
    t.coriol.assign(t.il, 0.0);
    for (int j = 0; j < t.iy; ++j) {
        t.coriol[j] = 2.0 * kOmega * (-t.sia_half[j]);
        t.coriol[t.il - 1 - j] = 2.0 * kOmega * t.sia_half[j];
    }
    t.akap = kap32();
    t.rgas = kap32() * kCp;
    t.grav = kGrav;
}

// ---- fftpack.f90:1-67 rffti1 --------------------------------------------------------------
// Factor order: try 4, 2, 3, 5, 7, ...; a factor 2 found after others is moved to the front.
void make_fft(HostTables &t)
{
    const int n = t.ix;
    int left = n, nf = 0, cand = 0;
    for (int pick = 0; left != 1; ++pick) {
        static const int first[4] = {4, 2, 3, 5};
        cand = pick < 4 ? first[pick] : cand + 2;
        while (left % cand == 0) {
            t.ifac[2 + nf] = cand;
            ++nf;
            left /= cand;
            if (cand == 2 && nf > 1) {
                for (int i = nf - 1; i >= 1; --i) t.ifac[2 + i] = t.ifac[1 + i];
                t.ifac[2] = 2;
            }
        }
    }
    t.ifac[0] = n;
    t.ifac[1] = nf;
    t.work.assign(n, 0.0);
    const double tpi = static_cast<double>(8.0f * atanf(1.0f));      // float32 2*pi (fftpack.f90:39)
    const double argh = tpi / static_cast<double>(n);
    int slot = 0, l1 = 1;
    for (int s = 0; s + 1 < nf; ++s) {
        const int ip = t.ifac[2 + s], ido = n / (l1 * ip);
        for (int j = 1; j < ip; ++j) {
            const double argld = static_cast<double>(j * l1) * argh;
            double fi = 0.0;
            for (int q = 0; q < (ido - 1) / 2; ++q) {      // (cos,sin) pairs in slots (2q, 2q+1)
                fi = fi + 1.0;
                const double arg = fi * argld;
                t.work[slot + 2 * q] = std::cos(arg);
                t.work[slot + 2 * q + 1] = std::sin(arg);
            }
            slot += ido;
        }
        l1 *= ip;
    }
    t.fwd_scale = static_cast<double>(1.0f / static_cast<float>(n));  // fourier.f90:72
    t.taui = static_cast<double>(0.5f * sqrtf(3.0f));                  // fftpack.f90:269,787
    t.sqrt2 = static_cast<double>(sqrtf(2.0f));                        // fftpack.f90:341
    t.hsqt2 = static_cast<double>(0.5f * sqrtf(2.0f));                 // fftpack.f90:857
}

// ---- legendre.f90:158-191 Gaussian weights (proper double Newton iteration) -------------------
void make_weights(HostTables &t)
{
    const int n = 2 * t.iy;
    t.wt.assign(t.iy, 0.0);
    double zprev = 2.0, dpoly = 0.0;   // zprev deliberately carries over between latitudes
    for (int i = 1; i <= t.iy; ++i) {
        double z = std::cos(3.141592654 * (static_cast<double>(i) - 0.25) / (static_cast<double>(n) + 0.5));
        while (std::fabs(z - zprev) > 2.220446049250313e-16) {
            double pa = 1.0, pb = 0.0;
            for (int j = 1; j <= n; ++j) {
                const double pc = pb;
                pb = pa;
                pa = ((2.0 * static_cast<double>(j) - 1.0) * z * pb - (static_cast<double>(j) - 1.0) * pc) /
                     static_cast<double>(j);
            }
            dpoly = static_cast<double>(n) * (z * pa - pb) / (z * z - 1.0);
            zprev = z;
            z = zprev - pa / dpoly;
        }
        t.wt[i - 1] = 2.0 / ((1.0 - z * z) * (dpoly * dpoly));
    }
}

// ---- legendre.f90:23-71 and :194-237 ----------------------------------------------------------
void make_legendre(HostTables &t)
{
    const int mx = t.mx, nx = t.nx, me = mx + 1, ne = nx + 1;
    make_weights(t);
    t.nsh2.assign(nx, 0);
    for (int n = 0; n < nx; ++n)
        for (int m = 0; m < mx; ++m)
            if (m + n <= t.trunc + 1 || t.ix != 4 * t.iy) t.nsh2[n] += 2;
    t.epsi.assign(static_cast<size_t>(me) * ne, 0.0);
    t.repsi.assign(static_cast<size_t>(me) * ne, 0.0);
    for (int m = 0; m < me; ++m)
        for (int n = 0; n < ne; ++n) {
            const double mm = static_cast<double>(static_cast<float>(m) * static_cast<float>(m));
            const double ll = static_cast<double>(static_cast<float>(m + n) * static_cast<float>(m + n));
            double e = 0.0;
            if (n != nx && !(n == 0 && m == 0)) e = std::sqrt((ll - mm) / (4.0 * ll - 1.0));
            t.epsi[m + me * n] = e;
            t.repsi[m + me * n] = e > 0.0 ? 1.0 / e : 0.0;
        }
    t.poly.assign(static_cast<size_t>(mx) * nx * t.iy, 0.0);
    std::vector<double> alp(static_cast<size_t>(me) * nx);
    const double tiny = static_cast<double>(1.e-30f);
    for (int j = 0; j < t.iy; ++j) {
        const double x = t.sia_half[j], y = t.coa_half[j];
        // sectoral (n=0) diagonal with float32 normalisation constants, then two-term recursion in n
        alp[0] = static_cast<double>(sqrtf(0.5f));
        for (int m = 1; m < me; ++m) {
            const float fm = static_cast<float>(m);
            const double cq = static_cast<double>(sqrtf(0.5f * (2.0f * fm + 1.0f) / fm));
            alp[m] = cq * y * alp[m - 1];
        }
        for (int m = 0; m < me; ++m) alp[m + me] = (x * alp[m]) * t.repsi[m + me];
        for (int n = 2; n < nx; ++n)
            for (int m = 0; m < me; ++m)
                alp[m + me * n] = (x * alp[m + me * (n - 1)] - t.epsi[m + me * (n - 1)] * alp[m + me * (n - 2)]) *
                                  t.repsi[m + me * n];
        double *dst = &t.poly[static_cast<size_t>(mx) * nx * j];
        for (int n = 0; n < nx; ++n)
            for (int m = 0; m < mx; ++m) {
                const double v = alp[m + me * n];
                dst[m + mx * n] = std::fabs(v) <= tiny ? 0.0 : v;
            }
    }
}

// ---- spectral.f90:20-82 -------------------------------------------------------------------------
void make_spectral(HostTables &t)
{
    const int mx = t.mx, nx = t.nx, me = mx + 1, sz = mx * nx;
    for (auto *v : {&t.el2, &t.elm2, &t.el4, &t.trfilt, &t.gradym, &t.gradyp, &t.uvdx, &t.uvdym, &t.uvdyp,
                    &t.vddym, &t.vddyp})
        v->assign(sz, 0.0);
    t.gradx.assign(mx, 0.0);
    const double a = kRearth;
    for (int n = 0; n < nx; ++n)
        for (int m = 0; m < mx; ++m) {
            const int l = m + n, i = m + mx * n;
            const double el = static_cast<double>(static_cast<float>(l));
            t.el2[i] = static_cast<double>(static_cast<float>(l * (l + 1))) / (a * a);
            t.el4[i] = t.el2[i] * t.el2[i];
            t.trfilt[i] = l <= t.trunc ? 1.0 : 0.0;
            t.elm2[i] = l == 0 ? 0.0 : 1.0 / t.el2[i];
            const double em = t.epsi[m + me * n], ep = t.epsi[m + me * (n + 1)];
            const double fm = static_cast<double>(static_cast<float>(m));
            if (n == 0) {
                t.gradx[m] = fm / a;
                t.uvdx[i] = -a / static_cast<double>(static_cast<float>(m + 1));
            } else {
                t.uvdx[i] = -a * fm / (el * (el + 1.0));
                t.gradym[i] = (el - 1.0) * em / a;
                t.uvdym[i] = -a * em / el;
                t.vddym[i] = (el + 1.0) * em / a;
            }
            t.gradyp[i] = (el + 2.0) * ep / a;
            t.uvdyp[i] = -a * ep / (el + 1.0);
            t.vddyp[i] = el * ep / a;
        }
}

// ---- horizontal_diffusion.f90:36-82 -------------------------------------------------------------
void make_hdiff(HostTables &t)
{
    const int mx = t.mx, nx = t.nx;
    t.dmp.assign(mx * nx, 0.0); t.dmpd.assign(mx * nx, 0.0); t.dmps.assign(mx * nx, 0.0);
    const double cf = 1.0 / (kThd * 3600.0), cd = 1.0 / (kThdd * 3600.0), cs = 1.0 / (kThds * 3600.0);
    const double rlap = static_cast<double>(1.0f / static_cast<float>(t.trunc * (t.trunc + 1)));   // float32 (:55)
    for (int n = 0; n < nx; ++n)
        for (int m = 0; m < mx; ++m) {
            const double l = static_cast<double>(static_cast<float>(m + n));
            const double e1 = l * (l + 1.0) * rlap;
            const double e2 = e1 * e1;
            const double e4 = e2 * e2;              // elap**4 as an integer power
            t.dmp[m + mx * n] = cf * e4;
            t.dmpd[m + mx * n] = cd * e4;
            t.dmps[m + mx * n] = cs * e1;
        }
}

// ---- matrix_inversion.f90: Crout LU (implicit scaling, partial pivoting) + back substitution ----
bool invert(std::vector<double> a, int n, double *inv_out)
{
    std::vector<double> scale(n);
    std::vector<int> perm(n);
    auto A = [&](int r, int c) -> double & { return a[r + n * c]; };
    for (int r = 0; r < n; ++r) {
        double big = 0.0;
        for (int c = 0; c < n; ++c) big = std::fabs(A(r, c)) > big ? std::fabs(A(r, c)) : big;
        if (big == 0.0) return false;
        scale[r] = 1.0 / big;
    }
    for (int c = 0; c < n; ++c) {
        for (int r = 0; r < c; ++r) {
            double s = A(r, c);
            for (int k = 0; k < r; ++k) s = s - A(r, k) * A(k, c);
            A(r, c) = s;
        }
        double best = 0.0;
        int prow = c;
        for (int r = c; r < n; ++r) {
            double s = A(r, c);
            for (int k = 0; k < c; ++k) s = s - A(r, k) * A(k, c);
            A(r, c) = s;
            const double fig = scale[r] * std::fabs(s);
            if (fig >= best) { best = fig; prow = r; }
        }
        if (prow != c) {
            for (int k = 0; k < n; ++k) { const double tmp = A(prow, k); A(prow, k) = A(c, k); A(c, k) = tmp; }
            scale[prow] = scale[c];
        }
        perm[c] = prow;
        if (A(c, c) == 0.0) A(c, c) = static_cast<double>(1.0e-20f);
        if (c != n - 1) {
            const double rcp = 1.0 / A(c, c);
            for (int r = c + 1; r < n; ++r) A(r, c) = A(r, c) * rcp;
        }
    }
    for (int col = 0; col < n; ++col) {
        double *b = inv_out + static_cast<size_t>(n) * col;
        for (int r = 0; r < n; ++r) b[r] = r == col ? 1.0 : 0.0;
        int first = -1;
        for (int r = 0; r < n; ++r) {
            const int p = perm[r];
            double s = b[p];
            b[p] = b[r];
            if (first >= 0) for (int k = first; k < r; ++k) s = s - A(r, k) * b[k];
            else if (s != 0.0) first = r;
            b[r] = s;
        }
        for (int r = n - 1; r >= 0; --r) {
            double s = b[r];
            for (int k = r + 1; k < n; ++k) s = s - A(r, k) * b[k];
            b[r] = s / A(r, r);
        }
    }
    return true;
}

}  // namespace

std::string HostTables::build(int trunc_, int ix_, int iy_, int kx_)
{
    trunc = trunc_; ix = ix_; iy = iy_; il = 2 * iy_; kx = kx_; nx = trunc_ + 2; mx = trunc_ + 1;
    if (trunc < 1 || iy < 1 || kx < 1) return "non-positive dimension";
    if (2 * mx > ix) return "2*(trunc+1) must not exceed ix";
    make_geometry(*this);
    make_fft(*this);
    for (int s = 0; s < ifac[1]; ++s)
        if (ifac[2 + s] != 2 && ifac[2 + s] != 3 && ifac[2 + s] != 4) return "ix has a prime factor other than 2 or 3";
    make_legendre(*this);
    make_spectral(*this);
    make_hdiff(*this);
    dmp1.assign(mx * nx, 0.0); dmp1d.assign(mx * nx, 0.0); dmp1s.assign(mx * nx, 0.0);
    return "";
}

std::string HostTables::set_sigma(const double *hsg_in)
{
    if (!hsg_in) return "null hsg";
    for (int k = 0; k <= kx; ++k) {
        if (!(hsg_in[k] >= 0.0 && hsg_in[k] <= 1.0)) return "half levels must lie in [0, 1]";
        if (k && !(hsg_in[k] > hsg_in[k - 1])) return "half levels must increase strictly";
    }
    for (int k = 0; k <= kx; ++k) hsg[k] = hsg_in[k];
    make_sigma_derived(*this);
    implicit_ready = false;
    return "";
}

std::string HostTables::build_implicit(double dt)
{
    if (!sigma_ready)
        return "no sigma levels: the reference defines them for kx in {5,7,8} only (geometry.f90:42-48); "
               "supply others with spdy_plan_set_sigma";
    const int sz = mx * nx, nl = mx + nx + 1;
    const double kap = kap32(), rgas = kap * kCp;
    const double rgam = rgas * kGammaLapse / (1000.0 * kGrav);
    for (int i = 0; i < sz; ++i) {
        dmp1[i] = 1.0 / (1.0 + dmp[i] * dt);
        dmp1d[i] = 1.0 / (1.0 + dmpd[i] * dt);
        dmp1s[i] = 1.0 / (1.0 + dmps[i] * dt);
    }
    tref.assign(kx, 0.0); tref1.assign(kx, 0.0); tref2.assign(kx, 0.0); tref3.assign(kx, 0.0);
    for (int k = 0; k < kx; ++k) {
        const double floor02 = static_cast<double>(0.2f);          // mixed-kind max(0.2, fsg)
        tref[k] = 288.0 * std::pow(fsg[k] > floor02 ? fsg[k] : floor02, rgam);
        tref1[k] = rgas * tref[k];
        tref2[k] = kap * tref[k];
        tref3[k] = fsgr[k] * tref[k];
    }
    const double xi = dt * kAlph, xxi = xi / (kRearth * kRearth);
    dhsx.assign(kx, 0.0);
    for (int k = 0; k < kx; ++k) dhsx[k] = xi * dhs[k];
    elz.assign(sz, 0.0);
    for (int n = 0; n < nx; ++n)
        for (int m = 0; m < mx; ++m)
            elz[m + mx * n] = static_cast<double>(static_cast<float>(m + n) * static_cast<float>(m + n + 1)) * xxi;
    auto at = [this](std::vector<double> &v, int r, int c) -> double & { return v[r + kx * c]; };
    std::vector<double> ya(kx * kx, 0.0), xa(kx * kx, 0.0), xb(kx * kx, 0.0), xe(kx * kx, 0.0), xf(kx * kx, 0.0), cum(kx);
    xc.assign(kx * kx, 0.0); xd.assign(kx * kx, 0.0); xj.assign(static_cast<size_t>(kx) * kx * nl, 0.0);
    for (int k = 0; k < kx; ++k)
        for (int c = 0; c < kx; ++c) at(ya, k, c) = -kap * tref[k] * dhs[c];
    for (int k = 1; k < kx; ++k) at(xa, k, k - 1) = 0.5 * (kap * tref[k] / fsg[k] - (tref[k] - tref[k - 1]) / dhs[k]);
    for (int k = 0; k + 1 < kx; ++k) at(xa, k, k) = 0.5 * (kap * tref[k] / fsg[k] - (tref[k + 1] - tref[k]) / dhs[k]);
    cum[0] = dhs[0];
    for (int k = 1; k < kx; ++k) cum[k] = cum[k - 1] + dhs[k];
    for (int k = 0; k + 1 < kx; ++k)
        for (int c = 0; c < kx; ++c) {
            at(xb, k, c) = dhs[c] * cum[k];
            if (c <= k) at(xb, k, c) = at(xb, k, c) - dhs[c];
        }
    for (int k = 0; k < kx; ++k)
        for (int c = 0; c < kx; ++c) {
            double s = at(ya, k, c);
            for (int q = 0; q + 1 < kx; ++q) s = s + at(xa, k, q) * at(xb, q, c);
            at(xc, k, c) = s;
        }
    for (int k = 0; k < kx; ++k) {
        for (int c = k + 1; c < kx; ++c) at(xd, k, c) = rgas * std::log(hsg[c + 1] / hsg[c]);
        at(xd, k, k) = rgas * std::log(hsg[k + 1] / fsg[k]);
    }
    for (int k = 0; k < kx; ++k)
        for (int c = 0; c < kx; ++c) {
            double s = 0.0;
            for (int q = 0; q < kx; ++q) s = s + at(xd, k, q) * at(xc, q, c);
            at(xe, k, c) = s;
        }
    for (int l = 1; l <= nl; ++l) {
        const double lam = static_cast<double>(static_cast<float>(l) * static_cast<float>(l + 1)) / (kRearth * kRearth);
        for (int k = 0; k < kx; ++k)
            for (int c = 0; c < kx; ++c) at(xf, k, c) = xi * xi * lam * (rgas * tref[k] * dhs[c] - at(xe, k, c));
        for (int k = 0; k < kx; ++k) at(xf, k, k) = at(xf, k, k) + 1.0;
        if (!invert(xf, kx, &xj[static_cast<size_t>(kx) * kx * (l - 1)])) return "singular semi-implicit matrix";
    }
    for (auto &v : xc) v = v * xi;
    implicit_ready = true;
    implicit_dt = dt;
    return "";
}

const double *HostTables::lookup(const std::string &name, int *count, std::vector<double> &scratch) const
{
    struct Ent { const char *n; const std::vector<double> *v; };
    const Ent ents[] = {
        {"sia_half", &sia_half}, {"coa_half", &coa_half}, {"cosgr", &cosgr}, {"cosgr2", &cosgr2},
        {"hsg", &hsg}, {"dhs", &dhs}, {"fsg", &fsg}, {"dhsr", &dhsr}, {"fsgr", &fsgr}, {"work", &work},
        {"epsi", &epsi}, {"wt", &wt}, {"poly", &poly}, {"el2", &el2}, {"elm2", &elm2}, {"el4", &el4},
        {"trfilt", &trfilt}, {"gradx", &gradx}, {"gradym", &gradym}, {"gradyp", &gradyp}, {"uvdx", &uvdx},
        {"uvdym", &uvdym}, {"uvdyp", &uvdyp}, {"vddym", &vddym}, {"vddyp", &vddyp}, {"dmp", &dmp},
        {"dmpd", &dmpd}, {"dmps", &dmps}, {"dmp1", &dmp1}, {"dmp1d", &dmp1d}, {"dmp1s", &dmp1s},
        {"tref", &tref}, {"tref1", &tref1}, {"tref2", &tref2}, {"tref3", &tref3}, {"xc", &xc}, {"xd", &xd},
        {"xj", &xj}, {"dhsx", &dhsx}, {"elz", &elz}, {"xgeop1", &xgeop1}, {"xgeop2", &xgeop2}, {"corf", &corf},
        {"tcorv", &tcorv}, {"qcorv", &qcorv}, {"coriol", &coriol}};
    for (const auto &e : ents)
        if (name == e.n) { *count = static_cast<int>(e.v->size()); return e.v->data(); }
    if (name == "ifac") {
        scratch.assign(ifac, ifac + 15);
        *count = 15;
        return scratch.data();
    }
    if (name == "nsh2") {
        scratch.assign(nsh2.begin(), nsh2.end());
        *count = static_cast<int>(nsh2.size());
        return scratch.data();
    }
    return nullptr;
}

}  // namespace spdy
