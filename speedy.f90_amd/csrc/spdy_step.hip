// gfx950 kernels for the spectral-space side of a speedy.f90 time step: everything between the direct transforms
// of step n and the inverse transforms of step n+1, so that prognostic spectra can stay in HBM across steps.
//   implicit_terms            implicit.f90:168-217       (any number of levels)
//   get_spectral_tendencies   tendencies.f90:242-293
//   get_geopotential          geopotential.f90:33-57
//   diffusion block of step   time_stepping.f90:62-96    (7 x do_horizontal_diffusion + ctmp + sdrag)
//   step_field_2d/3d          time_stepping.f90:121-167  (leapfrog + Robert-Asselin-Williams filter + trunct)
// All of it is elementwise per spectral coefficient (m,n) with short sequential loops over the kx levels: HBM/L2-bound
// work on a few hundred KB, one lane per coefficient (or per coefficient x level), reference operation order.
#include "spdy_kernels.hpp"
#include "spdy_cpx.hpp"

#include <algorithm>

#ifdef SPDY_PHASE_TRACE
__device__ long long g_step_trace[2 * 16];   // [wave 0 / wave 1][mark] s_memtime of block 0 (debug build only; tools/step_trace.py)
extern "C" __attribute__((visibility("default"))) int spdy_debug_step_trace(long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_step_trace), sizeof(g_step_trace));
}
#define STEP_MARK(m_) do { if (blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y < 2) g_step_trace[threadIdx.y * 16 + (m_)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STEP_MARK(m_) do { } while (0)
#endif
namespace spdy {

// Workgroup barrier that publishes LDS only: __syncthreads() also drains vmcnt, i.e. it waits for every global load in flight
// -- the kernels below exchange data through LDS alone and keep prefetches in flight across their barriers.
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------------------------------
// implicit_terms (implicit.f90:168-217).  Block = 64 coefficients x KY level rows (one wave per row, so the
// kx x kx matrices xd, xc are wave-uniform scalar loads; xj depends on l = m'+n per lane).  The three
// mat-vecs exchange their level vectors through LDS; every sum runs in the reference's order.
// ------------------------------------------------------------------------------------------
__global__ void implicit_kernel(DevPlan p, double *__restrict__ divdt, double *__restrict__ tdt, double *__restrict__ psdt)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];           // [3][kx][64] complex
    const int kx = p.kx, sz = p.mx * p.nx, tx = threadIdx.x, ty = threadIdx.y, ky = blockDim.y;
    const int e = blockIdx.x * 64 + tx;
    const bool valid = e < sz;
    const int ec = valid ? e : sz - 1;
    const int m = ec % p.mx, n = ec / p.mx, l = m + n;
    double *tl = sm, *yl = sm + (size_t)kx * 128, *dl = sm + (size_t)kx * 256;
    auto at = [&](double *b, int k) { return b + ((size_t)k * 64 + tx) * 2; };
    for (int k = ty; k < kx; k += ky) {
        const cpx v = ld(tdt, (long)k * sz + ec);
        at(tl, k)[0] = v.re; at(tl, k)[1] = v.im;
    }
    __syncthreads();
    const cpx ps0 = ld(psdt, ec);
    const double ez = p.elz[ec];
    for (int k = ty; k < kx; k += ky) {                                   // ye = xd*tdt + tref1*psdt ; yf = divdt + elz*ye
        cpx ye = {0.0, 0.0};
        for (int k1 = 0; k1 < kx; ++k1) ye = ye + p.xd[k + kx * k1] * cpx{at(tl, k1)[0], at(tl, k1)[1]};
        ye = ye + p.tref1[k] * ps0;
        const cpx yf = ld(divdt, (long)k * sz + ec) + ez * ye;
        at(yl, k)[0] = yf.re; at(yl, k)[1] = yf.im;
    }
    __syncthreads();
    for (int k = ty; k < kx; k += ky) {                                   // divdt = xj(:,:,l)*yf   (l = 0: stays 0)
        cpx d = {0.0, 0.0};
        if (l != 0) {
            const double *xj = p.xj + (long)kx * kx * (l - 1);
            for (int k1 = 0; k1 < kx; ++k1) d = d + xj[k + kx * k1] * cpx{at(yl, k1)[0], at(yl, k1)[1]};
        }
        at(dl, k)[0] = d.re; at(dl, k)[1] = d.im;
    }
    __syncthreads();
    if (ty == 0 && valid) {                                               // psdt = psdt - sum_k dhsx(k)*divdt(k)
        cpx ps = ps0;
        for (int k = 0; k < kx; ++k) ps = ps - p.dhsx[k] * cpx{at(dl, k)[0], at(dl, k)[1]};
        st(psdt, e, ps);
    }
    for (int k = ty; k < kx; k += ky) {                                   // tdt = tdt + xc*divdt
        cpx t = {at(tl, k)[0], at(tl, k)[1]};
        for (int k1 = 0; k1 < kx; ++k1) t = t + p.xc[k + kx * k1] * cpx{at(dl, k1)[0], at(dl, k1)[1]};
        if (valid) {
            st(tdt, (long)k * sz + e, t);
            st(divdt, (long)k * sz + e, cpx{at(dl, k)[0], at(dl, k)[1]});
        }
    }
}

static size_t implicit_lds(int kx) { return (size_t)3 * kx * 64 * 16; }

constexpr size_t spectral_step_lds(int kx, int bx = 16)   // 5 planes + (kx+1) sigma rows + 2 rows + row-major xd, xc + 5 level tables + 4 planes and a row (time level 2)
{
    return ((size_t)(6 * kx + 3) * 2 * bx + 2 * kx * ((kx + 1) & ~1) + ((5 * kx + 1) & ~1) + (size_t)(4 * kx + 1) * 2 * bx) * sizeof(double);
}
constexpr size_t spectral_step_lds_max() { return spectral_step_lds(16); }
constexpr int GT_BX = 16;            // grid points per block of the grid-tendencies kernel
constexpr size_t grid_tendencies_lds(int kx, int bx = GT_BX) { return ((size_t)(6 * kx + 2 * (kx + 1) + 3) * bx + kx) * sizeof(double); }
constexpr int STEP_BX = 16;          // coefficients per block of the one-launch spectral step
hipError_t prepare_device_step_kernels(int kx)
{
    static_assert(grid_tendencies_lds(16) <= 64 * 1024, "the grid-tendencies kernel fits the default dynamic LDS limit");
    if (implicit_lds(kx) > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(implicit_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)implicit_lds(kx));
        if (e != hipSuccess) return e;
    }
    static_assert(spectral_step_lds_max() <= 64 * 1024, "the one-launch spectral step fits the default dynamic LDS limit");
    return hipSuccess;
}

hipError_t launch_implicit(const DevPlan &p, double *divdt, double *tdt, double *psdt, hipStream_t s)
{
    const int sz = p.mx * p.nx, ky = std::min(p.kx, 16);
    if (implicit_lds(p.kx) > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(implicit_kernel, dim3((sz + 63) / 64), dim3(64, ky), implicit_lds(p.kx), s, p, divdt, tdt, psdt);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// step_field_2d / step_field_3d (time_stepping.f90:121-167), several prognostic arrays per launch (blockIdx.y).
//   field: (mx,nx,nlev,2) -- both time levels, level 1 first; fdt: (mx,nx,nlev), truncated in place like the
//   reference's intent(inout) dummy.  The second filter statement reads the already updated level-1 value (and
//   the updated output(:,:,j1) when j1 == 1), exactly as the Fortran array statements do.
// ------------------------------------------------------------------------------------------
__global__ void step_fields_kernel(DevPlan p, StepOps ops, int j1, double dt, double eps, double wil, int do_trunct)
{
    const int op = blockIdx.y, sz = p.mx * p.nx;
    const long total = (long)ops.nlev[op] * sz, i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    double *f = ops.field[op], *fdt = ops.fdt[op];
    cpx fd = ld(fdt, i);
    if (do_trunct) {                                                      // trunct (spectral.f90:229-233)
        fd = p.trfilt[(int)(i % sz)] * fd;
        st(fdt, i, fd);
    }
    const cpx o1 = ld(f, i), o2 = ld(f, total + i);
    const cpx fnew = o1 + dt * fd;
    const cpx oj = j1 == 1 ? o1 : o2;
    const cpx n1 = oj + (wil * eps) * ((o1 - 2.0 * oj) + fnew);
    const cpx oj2 = j1 == 1 ? n1 : o2;
    const cpx n2 = fnew - ((1.0 - wil) * eps) * ((n1 - 2.0 * oj2) + fnew);
    st(f, i, n1);
    st(f, total + i, n2);
}

hipError_t launch_step_fields(const DevPlan &p, const StepOps &ops, int j1, double dt, double eps, double wil, int do_trunct,
                              hipStream_t s)
{
    int maxlev = 0;
    for (int i = 0; i < ops.nops; ++i) maxlev = std::max(maxlev, ops.nlev[i]);
    const long total = (long)maxlev * p.mx * p.nx;
    if (ops.nops <= 0 || total <= 0) return hipSuccess;
    hipLaunchKernelGGL(step_fields_kernel, dim3((unsigned)((total + 255) / 256), ops.nops), dim3(256), 0, s, p, ops, j1, dt, eps,
                       wil, do_trunct);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// The diffusion block of `step` (time_stepping.f90:62-96): one lane per (level, coefficient), four prognostics.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ cpx hd(cpx field, cpx fdt, double dmp, double dmp1) { return dmp1 * (fdt - dmp * field); }

__global__ void hdiff_step_kernel(DevPlan p, HdiffStep h)
{
    const int sz = p.mx * p.nx;
    const long total = (long)p.kx * sz, i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i % sz), k = (int)(i / sz), m = e % p.mx;
    const double dmp = h.dmp[e], dmpd = h.dmpd[e], dmps = h.dmps[e], dmp1 = h.dmp1[e], dmp1d = h.dmp1d[e], dmp1s = h.dmp1s[e];
    const cpx vo = ld(h.vor, i), dv = ld(h.div, i);
    cpx vdt = hd(vo, ld(h.vordt, i), dmp, dmp1);                           // :63-64
    cpx ddt = hd(dv, ld(h.divdt, i), dmpd, dmp1d);
    const cpx ctmp = ld(h.t, i) + p.tcorv[k] * ld(h.tcorh, e);            // :66-72
    cpx tdt = hd(ctmp, ld(h.tdt, i), dmp, dmp1);                          // :74
    if (m == 0 && k == 0) {                                               // :77-81 zonal-mean wind drag, top level
        vdt = vdt - h.sdrag * vo;
        ddt = ddt - h.sdrag * dv;
    }
    vdt = hd(vo, vdt, dmps, dmp1s);                                       // :83-85
    ddt = hd(dv, ddt, dmps, dmp1s);
    tdt = hd(ctmp, tdt, dmps, dmp1s);
    st(h.vordt, i, vdt);
    st(h.divdt, i, ddt);
    st(h.tdt, i, tdt);
    if (h.tr) {                                                           // :88-96 (ntr = 1)
        const cpx cq = ld(h.tr, i) + p.qcorv[k] * ld(h.qcorh, e);
        st(h.trdt, i, hd(cq, ld(h.trdt, i), dmpd, dmp1d));
    }
}

hipError_t launch_hdiff_step(const DevPlan &p, const HdiffStep &h, hipStream_t s)
{
    const long total = (long)p.kx * p.mx * p.nx;
    hipLaunchKernelGGL(hdiff_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, h);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// get_geopotential (geopotential.f90:33-57): hydrostatic integration from the bottom level up, then the
// lapse-rate correction of the zonal (m' = 0) coefficients.  The recursion runs on the uncorrected values,
// as the reference's three separate loops do.  One lane per coefficient.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void geopotential_column(const DevPlan &p, int e, int sz, bool zonal, const double *t, cpx phis,
                                                    double *phi)
{
    const int kx = p.kx;
    cpx tk1 = ld(t, (long)(kx - 1) * sz + e);
    cpx ph = phis + p.xgeop1[kx - 1] * tk1;
    st(phi, (long)(kx - 1) * sz + e, ph);
    for (int k = kx - 2; k >= 0; --k) {
        const cpx tk = ld(t, (long)k * sz + e);
        ph = (ph + p.xgeop2[k + 1] * tk1) + p.xgeop1[k] * tk;
        cpx out = ph;
        if (zonal && k >= 1) out = ph + p.corf[k] * (tk1 - ld(t, (long)(k - 1) * sz + e));
        st(phi, (long)k * sz + e, out);
        tk1 = tk;
    }
}

__global__ void geopotential_kernel(DevPlan p, const double *__restrict__ t, const double *__restrict__ phis, double *__restrict__ phi)
{
    const int sz = p.mx * p.nx, e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= sz) return;
    geopotential_column(p, e, sz, e % p.mx == 0, t, ld(phis, e), phi);
}

hipError_t launch_geopotential(const DevPlan &p, const double *t, const double *phis, double *phi, hipStream_t s)
{
    const int sz = p.mx * p.nx;
    hipLaunchKernelGGL(geopotential_kernel, dim3((sz + 63) / 64), dim3(64), 0, s, p, t, phis, phi);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// get_spectral_tendencies (tendencies.f90:242-293).  One lane per coefficient; the level recurrences stream
// through registers (two passes over div, one over t), no per-level arrays.
// ------------------------------------------------------------------------------------------
__global__ void spectral_tendencies_kernel(DevPlan p, const double *__restrict__ div, const double *__restrict__ t,
                                           const double *__restrict__ ps, const double *__restrict__ phis,
                                           double *__restrict__ divdt, double *__restrict__ tdt, double *__restrict__ psdt,
                                           double *__restrict__ phi)
{
    const int sz = p.mx * p.nx, kx = p.kx, e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= sz) return;
    // vertical mean divergence and pressure tendency (:256-263)
    cpx dmean = {0.0, 0.0};
    for (int k = 0; k < kx; ++k) dmean = dmean + p.dhs[k] * ld(div, (long)k * sz + e);
    cpx pst = ld(psdt, e) - dmean;
    if (e == 0) pst = {0.0, 0.0};
    st(psdt, e, pst);
    // sigma-dot and temperature tendency (:265-285)
    cpx sig = {0.0, 0.0}, dumk = {0.0, 0.0};
    for (int k = 0; k < kx; ++k) {
        cpx sig1 = {0.0, 0.0}, dumk1 = {0.0, 0.0};
        if (k < kx - 1) {
            sig1 = sig - p.dhs[k] * (ld(div, (long)k * sz + e) - dmean);
            dumk1 = (p.tref[k + 1] - p.tref[k]) * sig1;
        }
        const cpx td = ((ld(tdt, (long)k * sz + e) - p.dhsr[k] * (dumk1 + dumk)) + p.tref3[k] * (sig1 + sig)) - p.tref2[k] * dmean;
        st(tdt, (long)k * sz + e, td);
        sig = sig1;
        dumk = dumk1;
    }
    // geopotential and divergence tendency (:287-292): divdt = divdt - laplacian(phi + rgas*tref*ps)
    geopotential_column(p, e, sz, e % p.mx == 0, t, ld(phis, e), phi);
    const cpx psv = ld(ps, e);
    const double l2 = p.el2[e];
    for (int k = 0; k < kx; ++k) {
        const cpx x = ld(phi, (long)k * sz + e) + p.rgtref[k] * psv;
        st(divdt, (long)k * sz + e, ld(divdt, (long)k * sz + e) - l2 * (-x));
    }
}

hipError_t launch_spectral_tendencies(const DevPlan &p, const double *div, const double *t, const double *ps, const double *phis,
                                      double *divdt, double *tdt, double *psdt, double *phi, hipStream_t s)
{
    const int sz = p.mx * p.nx;
    hipLaunchKernelGGL(spectral_tendencies_kernel, dim3((sz + 63) / 64), dim3(64), 0, s, p, div, t, ps, phis, divdt, tdt, psdt, phi);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Grid-space dynamical tendencies (tendencies.f90:105-197): one lane per grid point, the level recurrences
// (vertical means, sigma-dot, the half-level "temp" fluxes) stream through registers with a one-level look-ahead.
// Inputs are the results of the step's inverse transforms; outputs are laid out as the operands of ONE
// spdy_direct_batch_dev launch (tendencies.f90:212-234):
//   U, V [3 kx]      : (utend, vtend) | (-ug*tgg, -vg*tgg) | (-ug*trg, -vg*trg)      -> three kx-stacks of vdspec pairs
//   plain [3 kx + 1] : 0.5*(ug^2 + vg^2) | ttend | trtend | -umean*px - vmean*py
// (the host adds its physical tendencies to utend, vtend, ttend, trtend in between, tendencies.f90:203-206).
// ------------------------------------------------------------------------------------------
// kx > 16 (more level rows than a block holds): one lane per grid point, the level recurrences stream through registers
__global__ void grid_tendencies_serial_kernel(DevPlan p, GridTend g)
{
    const int gsz = p.ix * p.il, kx = p.kx, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= gsz) return;
    const int j = i / p.ix;
    const double cor = p.coriol[j], px = g.px[i], py = g.py[i], rgas = p.rgas, akap = p.akap;
#define LV(a, k) (a)[(long)(k) * gsz + i]
    // vertical means (:109-117)
    double umean = 0.0, vmean = 0.0, dmean = 0.0;
    for (int k = 0; k < kx; ++k) {
        const double dh = p.dhs[k];
        umean = umean + LV(g.ug, k) * dh;
        vmean = vmean + LV(g.vg, k) * dh;
        dmean = dmean + LV(g.divg, k) * dh;
    }
    g.plain[(long)(3 * kx) * gsz + i] = (-umean) * px - vmean * py;                       // (:125)
    // level sweep: at level k the fluxes temp(k), temp(k+1) of the three advected quantities are needed, i.e. sigdt(k),
    // sigdt(k+1) and the fields of levels k-1, k, k+1
    double sig = 0.0, sigm = 0.0;                                                          // sigdt(k), sigm(k): level 1 = 0
    double ug_m = 0.0, vg_m = 0.0, tgg_m = 0.0, tr_m = 0.0;                                // fields of level k-1
    double ug_c = LV(g.ug, 0), vg_c = LV(g.vg, 0), tg_c = LV(g.tg, 0), tr_c = LV(g.trg, 0);
    double tu = 0.0, tv = 0.0, tt = 0.0, tq = 0.0;                                         // temp(k) of u, v, t, tracer (temp(1) = 0)
    for (int k = 0; k < kx; ++k) {
        const double dh = p.dhs[k], dhr = p.dhsr[k];
        const double vor = LV(g.vorg, k) + cor, dv = LV(g.divg, k);                        // (:103-107 coriolis)
        const double tgg = tg_c - p.tref[k];                                               // (:149)
        const double puv = (ug_c - umean) * px + (vg_c - vmean) * py;                      // (:136)
        const double sig1 = sig - dh * (puv + dv - dmean);                                 // sigdt(k+1) (:139-142; the loop also sets level kx+1)
        const double sigm1 = sigm - dh * puv;
        // next level's fields and fluxes temp(k+1) (zero at kx+1: :152-153)
        double ug_n = 0.0, vg_n = 0.0, tg_n = 0.0, tr_n = 0.0, tu1 = 0.0, tv1 = 0.0, tt1 = 0.0, tq1 = 0.0;
        if (k + 1 < kx) {
            ug_n = LV(g.ug, k + 1); vg_n = LV(g.vg, k + 1); tg_n = LV(g.tg, k + 1); tr_n = LV(g.trg, k + 1);
            const double tgg_n = tg_n - p.tref[k + 1];
            tu1 = sig1 * (ug_n - ug_c);                                                    // (:156)
            tv1 = sig1 * (vg_n - vg_c);                                                    // (:166)
            tt1 = sig1 * (tgg_n - tgg) + sigm1 * (p.tref[k + 1] - p.tref[k]);              // (:176-177)
            tq1 = sig1 * (tr_n - tr_c);                                                    // (:188)
            if (k + 1 == 1 || k + 1 == 2) tq1 = 0.0;                                       // temp(:,:,2:3) = 0 (:191)
        }
        LV(g.u, k) = vg_c * vor - tgg * rgas * px - (tu1 + tu) * dhr;                      // utend (:160-161)
        LV(g.v, k) = -ug_c * vor - tgg * rgas * py - (tv1 + tv) * dhr;                     // vtend (:170-171)
        LV(g.plain, kx + k) = tgg * dv - (tt1 + tt) * dhr + p.fsgr[k] * tgg * (sig1 + sig) + p.tref3[k] * (sigm1 + sigm)
                              + akap * (tg_c * puv - tgg * dmean);                         // ttend (:181-184)
        LV(g.plain, 2 * kx + k) = tr_c * dv - (tq1 + tq) * dhr;                            // trtend (:194)
        LV(g.plain, k) = 0.5 * (ug_c * ug_c + vg_c * vg_c);                                // kinetic energy (:220)
        LV(g.u, kx + k) = -ug_c * tgg;  LV(g.v, kx + k) = -vg_c * tgg;                     // (:224)
        LV(g.u, 2 * kx + k) = -ug_c * tr_c;  LV(g.v, 2 * kx + k) = -vg_c * tr_c;           // (:229)
        sig = sig1; sigm = sigm1;
        ug_m = ug_c; vg_m = vg_c; tgg_m = tgg; tr_m = tr_c;
        ug_c = ug_n; vg_c = vg_n; tg_c = tg_n; tr_c = tr_n;
        tu = tu1; tv = tv1; tt = tt1; tq = tq1;
    }
    (void)ug_m; (void)vg_m; (void)tgg_m; (void)tr_m;
#undef LV
}

// Block = BX (16) grid points x kx level rows: every thread fetches its own level of the six fields at once
// (one coalesced memory round trip for the block instead of one per level of a serial per-point loop: 24 -> see DESIGN at
// T63 L16); the vertical means and the sigma-dot prefix sums are short loops over LDS by wave 0, everything else is per
// (point, level) with the neighbouring levels read from LDS.  Expressions as in the reference's loops.
// KM: level count bound of the instantiation (8 or 16); FULL: kx == KM, the level loops' guards fold (they are ONE wave's
// instruction stream while the block waits, like the recurrences of the spectral step).
// SH: level-sharded form (LevelShard, spdy_kernels.hpp): the six inputs come from ONE level-block stack that holds all levels,
// the outputs are this rank's own levels only, laid out as ITS direct-batch operands.  Same expressions, same order.
// Owner of level k among R ranks with blocks [kx r / R, kx (r + 1) / R): r = (R (k + 1) - 1) / kx.
struct LevelBlock { int r, lo, nl; };
__device__ __forceinline__ LevelBlock level_block(int k, int kx, int R)
{
    const int r = (R * (k + 1) - 1) / kx, lo = (kx * r) / R;
    return LevelBlock{r, lo, (kx * (r + 1)) / R - lo};
}

template <int KM, bool FULL, bool SH, bool WT = false>
__global__ __launch_bounds__(GT_BX * KM) void grid_tendencies_kernel(DevPlan p, GridTend g)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    // (TR, the transposed form of the sharded step: the slabs are this rank's npts points, global point = pt0 + i)
    const bool TR = SH && g.tr_out != nullptr;
    const int gsz = TR ? g.npts : p.ix * p.il, kx = FULL ? KM : p.kx, tx = threadIdx.x, k = threadIdx.y;
    constexpr int BX = GT_BX;                                              // grid points per block
    const int i0 = blockIdx.x * BX + tx;
    const bool valid = i0 < gsz;
    const int i = valid ? i0 : gsz - 1, ig = TR ? g.pt0 + i : i, j = ig / p.ix;
    double *su = sm, *sv = sm + kx * BX, *st = sm + 2 * kx * BX, *sq = sm + 3 * kx * BX, *sp = sm + 4 * kx * BX, *sd = sm + 5 * kx * BX;
    double *ssig = sm + 6 * kx * BX, *ssigm = ssig + (kx + 1) * BX, *smean = ssigm + (kx + 1) * BX;   // smean: umean, vmean, dmean rows
    double *stab = smean + 3 * BX;                                                                    // dhs[kx]
#define LV(a_, k_) (a_)[(long)(k_) * gsz + i]
#define S(b_, k_) (b_)[(k_) * BX + tx]
    // where this thread's level of input field f lives, and where (whether) its outputs go
    int og = kx, ok = k, o_ps = 3 * kx;                                    // output group stride, level slot, slot of the level-free field
    bool own = true;
    double ug_c, vg_c, tg_c, tr_c, dv, vor_in;
    double *out_u = g.u, *out_v = g.v, *out_pl = g.plain;
    if (SH) {
        const LevelBlock b = level_block(k, kx, g.sh.nranks);
        const long s0 = (long)6 * b.lo + (k - b.lo);                       // slab of (field 0, level k) in the F = 6 block stack
        ug_c = LV(g.ug, s0); vg_c = LV(g.ug, s0 + b.nl); vor_in = LV(g.ug, s0 + 2 * b.nl); dv = LV(g.ug, s0 + 3 * b.nl);
        tg_c = LV(g.ug, s0 + 4 * b.nl); tr_c = LV(g.ug, s0 + 5 * b.nl);
        const int lo = (kx * g.sh.rank) / g.sh.nranks, hi = (kx * (g.sh.rank + 1)) / g.sh.nranks;
        og = hi - lo; ok = k - lo; o_ps = 3 * og; own = k >= lo && k < hi;
        if (TR) {       // every level's results, into the block of the level's OWNER: u | v | plain (3 nl each) | the level-free field
            out_u = g.tr_out + ((long)9 * b.lo + b.r) * gsz; out_v = out_u + (long)3 * b.nl * gsz; out_pl = out_v + (long)3 * b.nl * gsz;
            og = b.nl; ok = k - b.lo; own = true;
        }
    } else {
        ug_c = LV(g.ug, k); vg_c = LV(g.vg, k); tg_c = LV(g.tg, k); tr_c = LV(g.trg, k); dv = LV(g.divg, k); vor_in = LV(g.vorg, k);
    }
    const double vor = vor_in + p.coriol[j];                                               // (:103-107 coriolis)
    const double px = g.px[ig], py = g.py[ig], rgas = p.rgas, akap = p.akap;
    const double dhr = p.dhsr[k], trefk = p.tref[k];
    const double tgg = tg_c - trefk;                                                       // (:149)
    if (tx == 0) stab[k] = p.dhs[k];
    S(su, k) = ug_c; S(sv, k) = vg_c; S(st, k) = tgg; S(sq, k) = tr_c; S(sd, k) = dv;
    lds_sync();
    if (k == 0) {                                                                          // vertical means (:109-117)
        double umean = 0.0, vmean = 0.0, dmean = 0.0;
        double cu[KM], cv[KM], cd[KM], dh[KM];                                             // operands first, then the chains
        UNROLL for (int kk = 0; kk < KM; ++kk) {
            const int kc = min(kk, kx - 1);
            cu[kk] = S(su, kc); cv[kk] = S(sv, kc); cd[kk] = S(sd, kc); dh[kk] = stab[kc];
        }
        UNROLL for (int kk = 0; kk < KM; ++kk) if (kk < kx) {
            umean = umean + cu[kk] * dh[kk];
            vmean = vmean + cv[kk] * dh[kk];
            dmean = dmean + cd[kk] * dh[kk];
        }
        smean[tx] = umean; smean[BX + tx] = vmean; smean[2 * BX + tx] = dmean;
    }
    lds_sync();
    const double umean = smean[tx], vmean = smean[BX + tx], dmean = smean[2 * BX + tx];
    const double puv = (ug_c - umean) * px + (vg_c - vmean) * py;                          // (:136)
    S(sp, k) = puv;
    lds_sync();
    if (k == 0) {                                                                          // sigdt, sigm at the half levels (:139-142)
        double sig = 0.0, sigm = 0.0;
        S(ssig, 0) = 0.0; S(ssigm, 0) = 0.0;
        double cp[KM], cd[KM], dh[KM];
        UNROLL for (int kk = 0; kk < KM; ++kk) { const int kc = min(kk, kx - 1); cp[kk] = S(sp, kc); cd[kk] = S(sd, kc); dh[kk] = stab[kc]; }
        UNROLL for (int kk = 0; kk < KM; ++kk) if (kk < kx) {
            sig = sig - dh[kk] * (cp[kk] + cd[kk] - dmean);
            sigm = sigm - dh[kk] * cp[kk];
            S(ssig, kk + 1) = sig; S(ssigm, kk + 1) = sigm;
        }
    }
    lds_sync();
    if (!valid) return;
    if (TR) {           // the level-free field goes into EVERY rank's block (level row q writes rank q's copy)
        if (k < g.sh.nranks) {
            const int lo_q = (kx * k) / g.sh.nranks, nl_q = (kx * (k + 1)) / g.sh.nranks - lo_q;
            g.tr_out[((long)9 * lo_q + k + 9 * nl_q) * gsz + i] = (-umean) * px - vmean * py;
        }
    } else if (k == 0) g.plain[(long)o_ps * gsz + i] = (-umean) * px - vmean * py;         // (:125)
    if (SH && !own) return;
    const double sig = S(ssig, k), sig1 = S(ssig, k + 1), sigm = S(ssigm, k), sigm1 = S(ssigm, k + 1);
    // fluxes temp(k) and temp(k+1) of the advected quantities (zero at the top level 1 and at kx+1: :152-153)
    double tu = 0.0, tv = 0.0, tt = 0.0, tq = 0.0, tu1 = 0.0, tv1 = 0.0, tt1 = 0.0, tq1 = 0.0;
    if (k > 0) {
        tu = sig * (ug_c - S(su, k - 1));                                                  // (:156)
        tv = sig * (vg_c - S(sv, k - 1));                                                  // (:166)
        tt = sig * (tgg - S(st, k - 1)) + sigm * (trefk - p.tref[k - 1]);                  // (:176-177)
        tq = sig * (tr_c - S(sq, k - 1));                                                  // (:188)
        if (k == 1 || k == 2) tq = 0.0;                                                    // temp(:,:,2:3) = 0 (:191)
    }
    if (k + 1 < kx) {
        tu1 = sig1 * (S(su, k + 1) - ug_c);
        tv1 = sig1 * (S(sv, k + 1) - vg_c);
        tt1 = sig1 * (S(st, k + 1) - tgg) + sigm1 * (p.tref[k + 1] - trefk);
        tq1 = sig1 * (S(sq, k + 1) - tr_c);
        if (k + 1 == 1 || k + 1 == 2) tq1 = 0.0;
    }
    // (WT: a launch with several MB of output writes it through, launch_grid_tendencies)
#define GT_ST(a_, k_, v_) do { if (WT) st1_wt(&LV(a_, k_), (v_)); else LV(a_, k_) = (v_); } while (0)
    GT_ST(out_u, ok, vg_c * vor - tgg * rgas * px - (tu1 + tu) * dhr);                       // utend (:160-161)
    GT_ST(out_v, ok, -ug_c * vor - tgg * rgas * py - (tv1 + tv) * dhr);                      // vtend (:170-171)
    GT_ST(out_pl, og + ok, tgg * dv - (tt1 + tt) * dhr + p.fsgr[k] * tgg * (sig1 + sig) + p.tref3[k] * (sigm1 + sigm)
                            + akap * (tg_c * puv - tgg * dmean));                          // ttend (:181-184)
    GT_ST(out_pl, 2 * og + ok, tr_c * dv - (tq1 + tq) * dhr);                             // trtend (:194)
    GT_ST(out_pl, ok, 0.5 * (ug_c * ug_c + vg_c * vg_c));                                 // kinetic energy (:220)
    GT_ST(out_u, og + ok, -ug_c * tgg);  GT_ST(out_v, og + ok, -vg_c * tgg);                   // (:224)
    GT_ST(out_u, 2 * og + ok, -ug_c * tr_c);  GT_ST(out_v, 2 * og + ok, -vg_c * tr_c);         // (:229)
#undef GT_ST
#undef LV
#undef S
}


hipError_t launch_grid_tendencies(const DevPlan &p, const GridTend &g, hipStream_t s)
{
    const int gsz = p.ix * p.il;
    if (g.sh.nranks >= 1 && (p.kx > 16 || g.sh.nranks > p.kx || g.sh.rank < 0 || g.sh.rank >= g.sh.nranks)) return hipErrorInvalidValue;
    if (p.kx > 16) hipLaunchKernelGGL(grid_tendencies_serial_kernel, dim3((gsz + 63) / 64), dim3(64), 0, s, p, g);
    else if (g.sh.nranks >= 1) {
        if (g.tr_out && (g.npts <= 0 || g.pt0 < 0 || g.pt0 + g.npts > gsz)) return hipErrorInvalidValue;
        const int npts = g.tr_out ? g.npts : gsz;
        const dim3 grd((npts + GT_BX - 1) / GT_BX), blk(GT_BX, p.kx);
        const size_t lds = grid_tendencies_lds(p.kx, GT_BX);
        if (p.kx == 8) hipLaunchKernelGGL((grid_tendencies_kernel<8, true, true>), grd, blk, lds, s, p, g);
        else if (p.kx < 8) hipLaunchKernelGGL((grid_tendencies_kernel<8, false, true>), grd, blk, lds, s, p, g);
        else if (p.kx == 16) hipLaunchKernelGGL((grid_tendencies_kernel<16, true, true>), grd, blk, lds, s, p, g);
        else hipLaunchKernelGGL((grid_tendencies_kernel<16, false, true>), grd, blk, lds, s, p, g);
    } else {
        // 16 points x kx levels per block: at model sizes the kernel is a latency chain per block, and 4x as many (smaller)
        // blocks spread its LDS traffic and loads over 4x as many CUs (T30: 72 -> 288 blocks; T63 L16 step 92.3 -> 89.2 us)
        constexpr int bx = GT_BX;
        const dim3 grd((gsz + bx - 1) / bx), blk(bx, p.kx);
        const size_t lds = grid_tendencies_lds(p.kx, bx);
        const bool wt = write_through_policy(p, (long)(9 * p.kx + 1) * gsz * 8);     // the launch's output: [3kx] + [3kx] + [3kx+1] grids
#define GT_LAUNCH(KM_, FULL_)                                                                                     \
    do {                                                                                                          \
        if (wt) hipLaunchKernelGGL((grid_tendencies_kernel<KM_, FULL_, false, true>), grd, blk, lds, s, p, g);    \
        else hipLaunchKernelGGL((grid_tendencies_kernel<KM_, FULL_, false, false>), grd, blk, lds, s, p, g);      \
    } while (0)
        if (p.kx == 8) GT_LAUNCH(8, true);
        else if (p.kx < 8) GT_LAUNCH(8, false);
        else if (p.kx == 16) GT_LAUNCH(16, true);
        else GT_LAUNCH(16, false);
#undef GT_LAUNCH
    }
    return hipGetLastError();
}

// The spectral-space end of get_grid_point_tendencies (tendencies.f90:125-126, 218-233), in place on the outputs of the
// direct batch: divdt -= laplacian(KE) ; tdt += ttend ; trdt += trtend ; psdt(1,1) = 0.
//   pdiv [3 kx] = (divdt | tdt part | trdt part) from the vdspec pairs, pspec [3 kx + 1] = (KE | ttend | trtend | psdt)
__global__ void tendency_combine_kernel(DevPlan p, double *__restrict__ pdiv, double *__restrict__ pspec)
{
    const int sz = p.mx * p.nx, kx = p.kx;
    const long total = (long)3 * kx * sz, i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) st(pspec, (long)3 * kx * sz, cpx{0.0, 0.0});
    if (i >= total) return;
    const int e = (int)(i % sz), blk = (int)(i / ((long)kx * sz));
    const cpx a = ld(pdiv, i), b = ld(pspec, i);
    st(pdiv, i, blk == 0 ? a - p.el2[e] * (-b) : a + b);
}

hipError_t launch_tendency_combine(const DevPlan &p, double *pdiv, double *pspec, hipStream_t s)
{
    const long total = (long)3 * p.kx * p.mx * p.nx;
    hipLaunchKernelGGL(tendency_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, pdiv, pspec);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// The whole spectral-space tail of a time step in ONE launch: tendency combination (tendencies.f90:125-126, 218-233),
// get_spectral_tendencies (:242-293), implicit_terms (implicit.f90:168-217), the diffusion block and step_field_*
// (time_stepping.f90:62-167).  At model sizes each of the five kernels above is a ~6 us launch around a few hundred KB;
// fused, the tendencies never leave the CU between them.  Block = BX (16) coefficients x kx level rows (kx <= 16): thread
// (e, k) owns the tendencies of one coefficient at one level in registers; the level-coupled parts (vertical sums,
// geopotential recursion, the kx x kx mat-vecs) go through LDS.  Every expression is the one of the separate kernels in the
// same order; the results agree with the unfused sequence to rounding (the compiler may contract a*b + c differently in the
// two translation contexts: include/spdy.h states "to rounding", the tests hold both forms to 1e-12 against the CPU checker).
// ------------------------------------------------------------------------------------------
// NJ = 16-byte pieces of a mat-vec row held in registers: 4 for up to 8 levels, 8 for up to 16; the block is STEP_BX (16)
// coefficients x kx level rows = 128 / 256 threads (launch bounds: two blocks per CU at NJ = 8)
// FULL: the level count IS the bound (8 or 16, the reference's and config 5's): every `kk < kx` guard and index clamp of the
// unrolled level loops folds away.  The level recurrences are executed by ONE wave while the block waits, so their
// instruction count (not a latency) is what the block pays: 1300 instructions at 4 cycles each were 2.7 us at kx = 16.
// SH: level-sharded form -- the direct batches' outputs of all ranks arrive as ONE level-block stack (LevelShard, F = 9, X = 1;
// SpecStep::sh) and the final tendencies leave through tend_out in the plain layout; everything else (the prognostics, the
// solve, the leapfrog) is on the full columns as ever, redundantly on every rank.  Same expressions, same order.
template <int NJ, bool FULL, bool SH>
__device__ __forceinline__ void spectral_step_body(const DevPlan &p, const SpecStep &a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];           // complex planes [kx][64]: divdt, tdt, phi / yf / d, div, t; + rows
    const int kx = FULL ? 2 * NJ : p.kx, sz = p.mx * p.nx, tx = threadIdx.x, k = threadIdx.y;
    constexpr int BX = STEP_BX;                                            // coefficients per block: LDS offsets are immediates
    // (TR, the transposed form of the sharded step: the blocks of this launch are the coefficients [e0, e0 + ne); the level-block
    // stack has slabs of ne values and is indexed with es = e - e0, everything else with the global e as ever)
    const bool TR = SH && a.ne > 0;
    const int e = (TR ? a.e0 : 0) + blockIdx.x * BX + tx, e_end = TR ? a.e0 + a.ne : sz;
    const bool valid = e < e_end;
    const int ec = valid ? e : e_end - 1, m = ec % p.mx, n = ec / p.mx, l = m + n;
    const int es = TR ? ec - a.e0 : ec;                                    // index into the direct batches' outputs
    const long ssz = TR ? a.ne : sz;                                       // ... and their slab size
    const long i = (long)k * sz + ec;                                     // this thread's (level, coefficient)
    const size_t PL = (size_t)kx * 2 * BX;                                 // doubles per complex plane [kx][BX]
    double *sdiv = sm, *stdt = sm + PL, *sy = sm + 2 * PL;
    auto at = [&](double *b, int kk) { return b + ((size_t)kk * BX + tx) * 2; };
    auto get = [&](double *b, int kk) { return cpx{at(b, kk)[0], at(b, kk)[1]}; };
    auto put = [&](double *b, int kk, cpx z) { at(b, kk)[0] = z.re; at(b, kk)[1] = z.im; };
    STEP_MARK(0);
    // Everything the kernel reads from global memory at a fixed place is requested HERE, in one batch with the tendencies:
    // time level 2 of the prognostics (consumed by the leapfrog at the very end, parked in LDS meanwhile), the surface
    // geopotential and the surface-pressure tendency (consumed by one level row each, but a load under a lane-dependent
    // condition is a branch and a full wait).  Requested where they are used, each was one more trip to memory on the block's
    // critical path.
    const long lvl2 = (long)kx * sz;
    const cpx vor2 = ld(a.vor, lvl2 + i), div2 = ld(a.div, lvl2 + i), t2 = ld(a.t, lvl2 + i), tr2 = ld(a.tr, lvl2 + i);
    // the direct batch's outputs: group f of stack X (A = pvor / raw_u, B = pdiv / raw_v, C = pspec) at this thread's level is
    // element  (lv + f * L) + off_X  of pointer p_X.  Plain layout: three stacks [3 kx] with their own pointers.  Level-block
    // layout: one pointer, block of this level's owner = [A | B | C] (3 nl each) + its copy of the level-free psdt.
    long lv = (long)k * sz, L = (long)kx * sz, offB = 0, offC = 0, ipsdt = (long)3 * kx * sz + es;
    const double *pA = a.raw_u ? a.raw_u : a.pvor, *pB = a.raw_u ? a.raw_v : a.pdiv, *pC = a.pspec;
    if (SH) {
        const LevelBlock b = level_block(k, kx, a.sh.nranks);
        lv = ((long)9 * b.lo + b.r + (k - b.lo)) * ssz; L = (long)b.nl * ssz; offB = 3 * L; offC = 6 * L;
        ipsdt = (long)9 * (kx / a.sh.nranks) * ssz + es;                  // block 0 = [9 nl_0] + psdt, nl_0 = floor(kx / R)
        pA = pB = pC = a.pvor;
    }
    const cpx ps2 = ld(a.ps, sz + ec), phs = ld(a.phis, ec), psdt_in = ld(pC, ipsdt);
    // ---- tendency combination on the direct batch's outputs
    cpx vordt, pd0, pd1, pd2;
    if (a.raw_u) {
        // vds of the three (u, v) pairs of this level where they are read (one launch less per step at T63, where the
        // transform kernel does not apply it): the expressions of vds_kernel (spdy_kernels.hip), row by row.  Every load is
        // unconditional at a clamped row; the rows the reference special-cases only choose among loaded values.
        const int nm = max(n - 1, 0), np = min(n + 1, p.nx - 1);
        const long i0 = lv + es, rm = lv + nm * p.mx + m, rp = lv + np * p.mx + m;      // (never TR: launch_spectral_step)
        const double gx = p.gradx[m], dm = p.vddym[ec], dp = p.vddyp[ec];
        const cpx u0 = ld(pA, i0), u0m = ld(pA, rm), u0p = ld(pA, rp);
        const cpx v0 = ld(pB, offB + i0), v0m = ld(pB, offB + rm), v0p = ld(pB, offB + rp);
        const cpx u1 = ld(pA, L + i0), v1m = ld(pB, offB + L + rm), v1p = ld(pB, offB + L + rp);
        const cpx u2 = ld(pA, 2 * L + i0), v2m = ld(pB, offB + 2 * L + rm), v2p = ld(pB, offB + 2 * L + rp);
        auto vds_vor = [&](cpx um, cpx up, cpx v) {
            if (n == 0) return times_i(gx * v) - dp * up;
            if (n == p.nx - 1) return dm * um;
            return (dm * um - dp * up) + times_i(gx * v);
        };
        auto vds_div = [&](cpx vm, cpx vp, cpx u) {
            if (n == 0) return times_i(gx * u) + dp * vp;
            if (n == p.nx - 1) return (-dm) * vm;
            return ((-dm) * vm + dp * vp) + times_i(gx * u);
        };
        vordt = vds_vor(u0m, u0p, v0);
        pd0 = vds_div(v0m, v0p, u0);
        pd1 = vds_div(v1m, v1p, u1);
        pd2 = vds_div(v2m, v2p, u2);
    } else {
        vordt = ld(pA, lv + es);
        pd0 = ld(pB, offB + lv + es);
        pd1 = ld(pB, offB + L + lv + es);
        pd2 = ld(pB, offB + 2 * L + lv + es);
    }
    cpx divdt = pd0 - p.el2[ec] * (-ld(pC, offC + lv + es));
    cpx tdt = pd1 + ld(pC, offC + L + lv + es);
    cpx trdt = pd2 + ld(pC, offC + 2 * L + lv + es);
    // ---- get_spectral_tendencies (time level 1 of div, t, ps).  Every thread brings its own level of div and t into LDS (one
    // coalesced global round trip for the block); the three level recurrences -- vertical mean, sigma-dot prefix sum, the
    // hydrostatic integration -- are then short loops over LDS by ONE wave each (k = 0 and k = 1 run them side by side),
    // and everything else (tdt, divdt updates, phi write-out) is per (coefficient, level) again.  As one thread per
    // coefficient reading global memory level by level this phase was 40 us at T63 L16.
    const cpx ps1 = ld(a.ps, ec);
    double *sdv = sm + 3 * PL, *st1 = sm + 4 * PL, *ssig = sm + 5 * PL;                                  // ssig: kx + 1 rows
    double *smisc = ssig + (size_t)(kx + 1) * 2 * BX;                                                     // rows: dmean, psdt
    put(sdv, k, ld(a.div, i));
    put(st1, k, ld(a.t, i));
    // implicit-solve operands, fetched now and consumed four barriers later (lds_sync leaves them in flight): this thread's row
    // of xj(:,:,l) into registers, its share of the xd / xc matrices (row-major copies, the same for every lane of a wave)
    // on the way to LDS.  Fetched where they are used, the three mat-vecs were half of this kernel's time.
    const int kxp = FULL ? 2 * NJ : p.kxp;
    double *sxd = smisc + 4 * BX, *sxc = sxd + kx * kxp;
    double2 xjr[4];                                             // (levels 8..15 of the row: fetched at the start of the solve)
    {
        const double2 *r2 = reinterpret_cast<const double2 *>(p.xjt + ((size_t)max(l - 1, 0) * kx + k) * kxp);
        UNROLL for (int j = 0; j < 4; ++j) xjr[j] = r2[min(j, kxp / 2 - 1)];
    }
    double cpd[2] = {0.0, 0.0}, cpc[2] = {0.0, 0.0};
    UNROLL for (int j = 0; j < 2; ++j) {
        const int q = k * BX + tx + j * BX * kx;
        if (q < kx * kxp) { cpd[j] = p.xdt[q]; cpc[j] = p.xct[q]; }
    }
    // per-level tables of the level recurrences below, into LDS: rows dhs, xgeop1, xgeop2, corf, dhsx.  (Read from global
    // memory inside the recurrences, every iteration waited for its own scalar loads: 7.7 us of this kernel at T63 L16.)
    constexpr int KM = 2 * NJ;                                  // level count bound of this instantiation (8 or 16)
    double *stb = sxc + kx * kxp;
    {
        const int q = k * BX + tx;
        if (q < kx) {
            stb[q] = p.dhs[q]; stb[kx + q] = p.xgeop1[q]; stb[2 * kx + q] = p.xgeop2[q]; stb[3 * kx + q] = p.corf[q];
            stb[4 * kx + q] = p.dhsx[q];
        }
    }
    double *sl2 = stb + ((5 * kx + 1) & ~1);                    // planes vor2, div2, t2, tr2 + a row for ps2
    STEP_MARK(1);
    lds_sync();
    STEP_MARK(2);
    put(sl2, k, vor2); put(sl2 + PL, k, div2); put(sl2 + 2 * PL, k, t2); put(sl2 + 3 * PL, k, tr2);
    if (k == 0) put(sl2 + 4 * PL, 0, ps2);
    cpx psdt = {0.0, 0.0};
    // The recurrences run over registers: every level's operands are fetched from LDS first (one batch of reads at fixed
    // offsets, indices clamped to kx - 1), the dependent chain is then arithmetic only.
    if (k == 0) {                                               // vertical mean, surface-pressure tendency, sigma-dot (:256-275)
        psdt = psdt_in;
        cpx dv[KM];
        double dh[KM];
        UNROLL for (int kk = 0; kk < KM; ++kk) { const int kc = min(kk, kx - 1); dv[kk] = get(sdv, kc); dh[kk] = stb[kc]; }
        if (ec == 0) psdt = {0.0, 0.0};                                    // tendencies.f90:126
        cpx dmean = {0.0, 0.0};
        UNROLL for (int kk = 0; kk < KM; ++kk) if (kk < kx) dmean = dmean + dh[kk] * dv[kk];
        psdt = psdt - dmean;
        if (ec == 0) psdt = {0.0, 0.0};
        put(smisc, 0, dmean);
        cpx sig = {0.0, 0.0};
        put(ssig, 0, sig);
        UNROLL for (int kk = 0; kk < KM; ++kk) if (kk < kx) {
            cpx sig1 = {0.0, 0.0};
            if (kk < kx - 1) sig1 = sig - dh[kk] * (dv[kk] - dmean);
            put(ssig, kk + 1, sig1);
            sig = sig1;
        }
    }
    // (on another wave than the loops above where the block has one: BX = 16 puts four level rows into a wave)
    if (k == min(kx - 1, max(1, 64 / BX))) {                     // get_geopotential (geopotential.f90:33-57) into the sy plane
        const bool zonal = m == 0;
        cpx tk[KM];
        UNROLL for (int kk = 0; kk < KM; ++kk) tk[kk] = get(st1, min(kk, kx - 1));
        cpx tk1 = get(st1, kx - 1);
        cpx ph = phs + stb[kx + kx - 1] * tk1;
        put(sy, kx - 1, ph);
        UNROLL for (int hb = KM - 8; hb >= 0; hb -= 8) {         // the table rows eight levels at a time (registers)
            double g1[8], g2n[8], cf[8];                         // xgeop1[kk], xgeop2[kk + 1], corf[kk] of kk = hb .. hb + 7
            UNROLL for (int j = 0; j < 8; ++j) {
                const int kc = min(hb + j, kx - 1);
                g1[j] = stb[kx + kc]; g2n[j] = stb[2 * kx + min(hb + j + 1, kx - 1)]; cf[j] = stb[3 * kx + kc];
            }
            UNROLL for (int kk = hb + 7; kk >= hb; --kk) if (kk <= KM - 2 && kk <= kx - 2) {
                ph = (ph + g2n[kk - hb] * tk1) + g1[kk - hb] * tk[kk];
                cpx out = ph;
                if (kk >= 1) {
                    const cpx oz = ph + cf[kk - hb] * (tk1 - tk[kk - 1]);
                    if (zonal) out = oz;
                }
                put(sy, kk, out);
                tk1 = tk[kk];
            }
        }
    }
    lds_sync();
    STEP_MARK(3);
    {   // this thread's level: temperature and divergence tendencies (:277-292), phi out
        const cpx dmean = get(smisc, 0), sig = get(ssig, k), sig1 = get(ssig, k + 1);
        const cpx dumk = k > 0 ? (p.tref[k] - p.tref[k - 1]) * sig : cpx{0.0, 0.0};
        const cpx dumk1 = k < kx - 1 ? (p.tref[k + 1] - p.tref[k]) * sig1 : cpx{0.0, 0.0};
        tdt = ((tdt - p.dhsr[k] * (dumk1 + dumk)) + p.tref3[k] * (sig1 + sig)) - p.tref2[k] * dmean;
        const cpx ph = get(sy, k);
        if (valid) st(a.phi, i, ph);
        const cpx x = (valid ? ph : cpx{0.0, 0.0}) + p.rgtref[k] * ps1;
        divdt = divdt - p.el2[ec] * (-x);
    }
    UNROLL for (int j = 0; j < 2; ++j) {
        const int q = k * BX + tx + j * BX * kx;
        if (q < kx * kxp) { sxd[q] = cpd[j]; sxc[q] = cpc[j]; }
    }
    lds_sync();                                            // (sy is reused by the implicit solve below)
    put(sdiv, k, divdt);
    put(stdt, k, tdt);
    lds_sync();
    STEP_MARK(4);
    // ---- implicit_terms
    const double ez = p.elz[ec];
    {   // psdt lives with thread k == 0: broadcast through LDS scratch slot 0 of sy's fourth plane
        double *sps = smisc + 2 * BX;
        if (k == 0) { sps[2 * tx] = psdt.re; sps[2 * tx + 1] = psdt.im; }
        lds_sync();
        const cpx ps0 = {sps[2 * tx], sps[2 * tx + 1]};
        // mat-vec rows from the row-major copies; the sums keep the reference's order k1 = 0..kx-1
        auto matvec = [&](const double *row, double *src, cpx acc) {                 // row in LDS; operands first, then the chain
            cpx v[KM];
            double r[KM];
            UNROLL for (int k1 = 0; k1 < KM; ++k1) { const int kc = min(k1, kx - 1); v[k1] = get(src, kc); r[k1] = row[kc]; }
            UNROLL for (int k1 = 0; k1 < KM; ++k1) if (k1 < kx) acc = acc + r[k1] * v[k1];
            return acc;
        };
        double2 xjl[4];
        if (NJ > 4) {
            const double2 *r2 = reinterpret_cast<const double2 *>(p.xjt + ((size_t)max(l - 1, 0) * kx + k) * kxp);
            UNROLL for (int j = 0; j < 4; ++j) xjl[j] = r2[min(4 + j, kxp / 2 - 1)];
        }
        cpx ye = matvec(sxd + k * kxp, stdt, cpx{0.0, 0.0});
        ye = ye + p.tref1[k] * ps0;
        put(sy, k, get(sdiv, k) + ez * ye);                                // yf
        lds_sync();
        cpx d = {0.0, 0.0};
        {
            cpx yv[KM];
            UNROLL for (int k1 = 0; k1 < KM; ++k1) yv[k1] = get(sy, min(k1, kx - 1));
            UNROLL for (int j = 0; j < 4; ++j) {
                if (2 * j < kx) d = d + xjr[j].x * yv[2 * j];
                if (2 * j + 1 < kx) d = d + xjr[j].y * yv[2 * j + 1];
            }
            if (NJ > 4) {
                UNROLL for (int j = 4; j < 8; ++j) {
                    if (2 * j < kx) d = d + xjl[j - 4].x * yv[2 * j];
                    if (2 * j + 1 < kx) d = d + xjl[j - 4].y * yv[2 * j + 1];
                }
            }
            if (l == 0) d = {0.0, 0.0};
        }
        lds_sync();
        put(sy, k, d);                                                     // divdt after the solve
        lds_sync();
        if (k == 0) {
            cpx ps = ps0, dk[KM];
            double hx[KM];
            UNROLL for (int kk = 0; kk < KM; ++kk) { const int kc = min(kk, kx - 1); dk[kk] = get(sy, kc); hx[kk] = stb[4 * kx + kc]; }
            UNROLL for (int kk = 0; kk < KM; ++kk) if (kk < kx) ps = ps - hx[kk] * dk[kk];
            psdt = ps;
        }
        tdt = matvec(sxc + k * kxp, sy, get(stdt, k));
        divdt = d;
    }
    STEP_MARK(5);
    if (!valid) return;
    // ---- diffusion block of step() (time level 1 of the prognostics)
    cpx vo, dv, t1v, tr1;
    {
        const double dmp = p.dmp_t[0][ec], dmpd = p.dmp_t[1][ec], dmps = p.dmp_t[2][ec];
        const double dmp1 = p.dmp_t[3][ec], dmp1d = p.dmp_t[4][ec], dmp1s = p.dmp_t[5][ec];
        vo = ld(a.vor, i); dv = ld(a.div, i); t1v = ld(a.t, i); tr1 = ld(a.tr, i);
        vordt = hd(vo, vordt, dmp, dmp1);
        divdt = hd(dv, divdt, dmpd, dmp1d);
        const cpx ctmp = t1v + p.tcorv[k] * ld(a.tcorh, ec);
        tdt = hd(ctmp, tdt, dmp, dmp1);
        if (m == 0 && k == 0) {
            vordt = vordt - a.sdrag * vo;
            divdt = divdt - a.sdrag * dv;
        }
        vordt = hd(vo, vordt, dmps, dmp1s);
        divdt = hd(dv, divdt, dmps, dmp1s);
        tdt = hd(ctmp, tdt, dmps, dmp1s);
        const cpx cq = tr1 + p.qcorv[k] * ld(a.qcorh, ec);
        trdt = hd(cq, trdt, dmpd, dmp1d);
    }
    STEP_MARK(6);
    // ---- step_field_3d for vor, div, t, tr and step_field_2d for ps; tendencies written back truncated, as the
    // separate kernels leave them
    const double trf = p.trfilt[ec];
    auto stepf = [&](double *f, long idx, long off2, cpx o1, cpx o2, cpx fd, double *fdt_out, long fdt_idx) {
        if (a.do_trunct) fd = trf * fd;
        st(fdt_out, fdt_idx, fd);
        const cpx fnew = o1 + a.dt * fd;
        const cpx oj = a.j1 == 1 ? o1 : o2;
        const cpx n1 = oj + (a.wil * a.eps) * ((o1 - 2.0 * oj) + fnew);
        const cpx oj2 = a.j1 == 1 ? n1 : o2;
        const cpx n2 = fnew - ((1.0 - a.wil) * a.eps) * ((n1 - 2.0 * oj2) + fnew);
        st(f, idx, n1);
        st(f, off2 + idx, n2);
    };
    // (each thread reads back what it parked itself: no barrier needed)
    if (SH) {           // the operands are other ranks' blocks too (and, raw, neighbouring rows of other threads): tendencies leave apart
        stepf(a.vor, i, lvl2, vo, get(sl2, k), vordt, a.tend_out, i);
        stepf(a.div, i, lvl2, dv, get(sl2 + PL, k), divdt, a.tend_out, lvl2 + i);
        stepf(a.t, i, lvl2, t1v, get(sl2 + 2 * PL, k), tdt, a.tend_out, 2 * lvl2 + i);
        stepf(a.tr, i, lvl2, tr1, get(sl2 + 3 * PL, k), trdt, a.tend_out, 3 * lvl2 + i);
        if (k == 0) stepf(a.ps, ec, sz, ps1, get(sl2 + 4 * PL, 0), psdt, a.tend_out, 4 * lvl2 + ec);
    } else {
        stepf(a.vor, i, lvl2, vo, get(sl2, k), vordt, a.pvor, i);
        stepf(a.div, i, lvl2, dv, get(sl2 + PL, k), divdt, a.pdiv, i);
        stepf(a.t, i, lvl2, t1v, get(sl2 + 2 * PL, k), tdt, a.pdiv, (long)kx * sz + i);
        stepf(a.tr, i, lvl2, tr1, get(sl2 + 3 * PL, k), trdt, a.pdiv, (long)2 * kx * sz + i);
        if (k == 0) stepf(a.ps, ec, sz, ps1, get(sl2 + 4 * PL, 0), psdt, a.pspec, (long)3 * kx * sz + ec);
    }
    STEP_MARK(7);
}

// blocks of 16 x kx threads: NJ = 4 up to 8 levels, NJ = 8 up to 16 (two waves per SIMD: 260 blocks at T63 must not need two rounds)
template <int NJ, bool FULL, bool SH>
__global__ __launch_bounds__(STEP_BX * 2 * NJ, NJ > 4 ? 2 : 1) void spectral_step_kernel(DevPlan p, SpecStep a) { spectral_step_body<NJ, FULL, SH>(p, a); }

// Coefficients per block.  16 (x kx level rows): with 64 the T63 launch was 65 blocks of up to 1024 threads whose three
// kx-term mat-vecs went through ONE CU's LDS each; 260 blocks of a quarter the size use the whole chip (captured step
// T63 L16 96.9 -> 92.2 us, T30 L16 57.1 -> 51.6 us; kx = 8: 0.5-1 us).  Needs 2 * bx >= kxp for the xd / xc staging.
int spectral_step_bx(const DevPlan &) { return STEP_BX; }

hipError_t launch_spectral_step(const DevPlan &p, const SpecStep &a, hipStream_t s)
{
    const int sz = p.mx * p.nx;
    if (p.kx > 16) return hipErrorInvalidValue;
    const int bx = spectral_step_bx(p);
    const size_t lds = spectral_step_lds(p.kx, bx);
    dim3 grd((sz + bx - 1) / bx), blk(bx, p.kx);
    if (a.sh.nranks >= 1) {
        if (a.sh.nranks > p.kx || !a.tend_out) return hipErrorInvalidValue;
        if (a.ne > 0) {                                                    // transposed form: this rank's coefficient range only
            if (a.raw_u || a.e0 < 0 || a.e0 % bx || a.e0 + a.ne > sz) return hipErrorInvalidValue;
            grd = dim3((a.ne + bx - 1) / bx);
        }
        if (p.kx == 8) hipLaunchKernelGGL((spectral_step_kernel<4, true, true>), grd, blk, lds, s, p, a);
        else if (p.kx < 8) hipLaunchKernelGGL((spectral_step_kernel<4, false, true>), grd, blk, lds, s, p, a);
        else if (p.kx == 16) hipLaunchKernelGGL((spectral_step_kernel<8, true, true>), grd, blk, lds, s, p, a);
        else hipLaunchKernelGGL((spectral_step_kernel<8, false, true>), grd, blk, lds, s, p, a);
        return hipGetLastError();
    }
    if (p.kx == 8) hipLaunchKernelGGL((spectral_step_kernel<4, true, false>), grd, blk, lds, s, p, a);
    else if (p.kx < 8) hipLaunchKernelGGL((spectral_step_kernel<4, false, false>), grd, blk, lds, s, p, a);
    else if (p.kx == 16) hipLaunchKernelGGL((spectral_step_kernel<8, true, false>), grd, blk, lds, s, p, a);
    else hipLaunchKernelGGL((spectral_step_kernel<8, false, false>), grd, blk, lds, s, p, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Output path (input_output.f90:184-206): after the 5 kx + 1 inverse transforms, the gridded fields are scaled and
// rounded to float32.  gather_spectra packs the separately stored plain spectra (t, q, phi levels and ps) into one
// stack so that the whole snapshot is ONE transform launch; output_cast does the float32 epilogue.
// ------------------------------------------------------------------------------------------
__global__ void gather_spectra_kernel(GatherOps g, int sz2)
{
    const int op = blockIdx.y;
    const long total = (long)g.nfld[op] * sz2, i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // in double2 units
    if (i >= total) return;
    reinterpret_cast<double2 *>(g.dst[op])[i] = reinterpret_cast<const double2 *>(g.src[op])[i];
}

hipError_t launch_gather_spectra(const DevPlan &p, const GatherOps &g, hipStream_t s)
{
    int maxf = 0;
    for (int i = 0; i < g.nops; ++i) maxf = std::max(maxf, g.nfld[i]);
    const int sz2 = p.mx * p.nx;                                    // complex = one double2
    const long total = (long)maxf * sz2;
    if (g.nops <= 0 || total <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_spectra_kernel, dim3((unsigned)((total + 255) / 256), g.nops), dim3(256), 0, s, g, sz2);
    return hipGetLastError();
}

// kind per segment: 0 plain, 1 times `factor`, 2 divided by `factor`, 3 factor*exp(x)
__global__ void output_cast_kernel(OutputCast c, int gsz)
{
    const int op = blockIdx.y;
    const long total = (long)c.nfld[op] * gsz, i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= total) return;
    const double2 v = *reinterpret_cast<const double2 *>(c.src[op] + i);
    const int kind = c.kind[op];
    const double f = c.factor[op];
    double a = v.x, b = v.y;
    if (kind == 1) { a = a * f; b = b * f; }
    else if (kind == 2) { a = a / f; b = b / f; }
    else if (kind == 3) { a = f * exp(a); b = f * exp(b); }
    *reinterpret_cast<float2 *>(c.dst[op] + i) = make_float2((float)a, (float)b);
}

hipError_t launch_output_cast(const DevPlan &p, const OutputCast &c, hipStream_t s)
{
    int maxf = 0;
    for (int i = 0; i < c.nops; ++i) maxf = std::max(maxf, c.nfld[i]);
    const int gsz = p.ix * p.il;                                    // even
    const long pairs = (long)maxf * gsz / 2;
    if (c.nops <= 0 || pairs <= 0) return hipSuccess;
    hipLaunchKernelGGL(output_cast_kernel, dim3((unsigned)((pairs + 255) / 256), c.nops), dim3(256), 0, s, c, gsz);
    return hipGetLastError();
}

}  // namespace spdy
