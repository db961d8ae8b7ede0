// C ABI (include/spdy.h) over the gfx950 kernels: plan management, table upload, host-pointer
// drop-ins and device-pointer batched entry points.  No CPU compute path exists here: a plan
// without a device can only report its tables.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "spdy_plan.hpp"
#include "spdy_t63_sched.hpp"
#include <chrono>

using spdy::DevPlan;
using spdy::HostTables;
using namespace spdy_detail;

namespace {
thread_local std::string g_err;
}

namespace spdy_detail {

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

int dev_alloc(spdy_plan *p, size_t bytes, void **out)
{
    void *ptr = nullptr;
    HIP_TRY(hipMalloc(&ptr, bytes ? bytes : 8));
    p->allocs.push_back(ptr);
    *out = ptr;
    return SPDY_OK;
}

}  // namespace spdy_detail

namespace {

int dev_upload(spdy_plan *p, const std::vector<double> &v, const double **out)
{
    void *ptr = nullptr;
    int rc = dev_alloc(p, v.size() * sizeof(double), &ptr);
    if (rc) return rc;
    if (!v.empty()) HIP_TRY(hipMemcpy(ptr, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
    *out = static_cast<const double *>(ptr);
    return SPDY_OK;
}

// MFMA A-operand images of the Legendre table (v_mfma_f64_16x16x4_f64: lane l supplies
// A[row = l & 15][k = l >> 4]); zero padded so ragged edges need no masking in the kernel.
void build_mfma_tables(const HostTables &t, int ks_inv, int jt, int nt_dir, int js_dir,
                       std::vector<double> &inv, std::vector<double> &dir)
{
    const int mx = t.mx, nx = t.nx, iy = t.iy;
    auto P = [&](int m, int n, int j) { return t.poly[m + mx * (n + (size_t)nx * j)]; };
    inv.assign((size_t)mx * 2 * ks_inv * jt * 64, 0.0);
    dir.assign((size_t)mx * 2 * nt_dir * js_dir * 64, 0.0);
    for (int m = 0; m < mx; ++m)
        for (int par = 0; par < 2; ++par) {
            // inverse (legendre.f90:92-103): rows = latitudes, k = n of this parity with m+n <= trunc+1
            for (int ks = 0; ks < ks_inv; ++ks)
                for (int tt = 0; tt < jt; ++tt)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int j = 16 * tt + (lane & 15), n = 2 * (4 * ks + (lane >> 4)) + par;
                        double v = 0.0;
                        if (j < iy && n < nx && m + n <= t.trunc + 1) v = P(m, n, j);
                        inv[((((size_t)m * 2 + par) * ks_inv + ks) * jt + tt) * 64 + lane] = v;
                    }
            // direct (legendre.f90:142-154): rows = n of this parity with n <= trunc and m+n <= trunc+1,
            // k = latitude; Gaussian weight folded in (legendre.f90:131-132)
            for (int tt = 0; tt < nt_dir; ++tt)
                for (int ks = 0; ks < js_dir; ++ks)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int n = 2 * (16 * tt + (lane & 15)) + par, j = 4 * ks + (lane >> 4);
                        double v = 0.0;
                        if (j < iy && n <= t.trunc && m + n <= t.trunc + 1) v = P(m, n, j) * t.wt[j];
                        dir[((((size_t)m * 2 + par) * nt_dir + tt) * js_dir + ks) * 64 + lane] = v;
                    }
        }
}

// A fragments for v_mfma_f64_4x4x4_4b_f64 with both parities packed into one instruction
// (block = (lane>>2)&3, A row i = lane&3, k = lane>>4; blocks 0,1 -> even-n sums, 2,3 -> odd-n sums).
void build_packed_tables(const HostTables &t, int ks_inv, int js_dir, std::vector<double> &inv2, std::vector<double> &dir2)
{
    const int mx = t.mx, nx = t.nx, iy = t.iy;
    auto P = [&](int m, int n, int j) { return t.poly[m + mx * (n + (size_t)nx * j)]; };
    inv2.assign((size_t)mx * ks_inv * 64, 0.0);
    dir2.assign((size_t)mx * js_dir * 64, 0.0);
    for (int m = 0; m < mx; ++m) {
        for (int ks = 0; ks < ks_inv; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int blk = (lane >> 2) & 3, i = lane & 3, k = lane >> 4, par = blk >> 1;
                const int lat = 16 + 4 * (blk & 1) + i, n = 2 * (4 * ks + k) + par;
                if (lat < iy && n < nx && m + n <= t.trunc + 1) inv2[((size_t)m * ks_inv + ks) * 64 + lane] = P(m, n, lat);
            }
        for (int ks = 0; ks < js_dir; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int blk = (lane >> 2) & 3, i = lane & 3, k = lane >> 4, par = blk >> 1;
                const int n = 2 * (4 * (blk & 1) + i) + par, lat = 4 * ks + k;
                if (lat < iy && n <= t.trunc && m + n <= t.trunc + 1)
                    dir2[((size_t)m * js_dir + ks) * 64 + lane] = P(m, n, lat) * t.wt[lat];
            }
    }
}

// A-operand images of the fused T63 kernels (spdy_fused_t63.inc): per Legendre wave and slot (quad q, parity, n-group g)
// one double2 per lane and latitude chunk; 4x4x4 block = zonal wavenumber 4q+blk, A row = lane&3, k = lane>>4.
//   direct : A[i = n row][k = lat]  = P(m,n,lat)*wt(lat),  .x/.y = the chunk's two k-steps of 4 latitudes
//   inverse: A[i = lat][k = n]      = P(m,n,lat),          .x/.y = the chunk's two groups of 4 latitudes
void build_t63_images(const HostTables &t, std::vector<double> &dir, std::vector<double> &inv)
{
    using namespace spdy::t63;
    const int mx = t.mx, nx = t.nx;
    auto P = [&](int m, int n, int j) { return t.poly[m + mx * (n + (size_t)nx * j)]; };
    dir.assign((size_t)NLW * MAXS * NCH * 64 * 2, 0.0);
    inv.assign((size_t)NLW * MAXS * NCH * 64 * 2, 0.0);
    for (int w = 0; w < NLW; ++w)
        for (int i = 0; i < 4; ++i) {
            const int q = wave_quad(w, i);
            for (int par = 0; par < 2; ++par)
                for (int dirflag = 0; dirflag < 2; ++dirflag)
                    for (int g = 0; g < ngrp(dirflag, q, par); ++g) {
                        const int s = slot_of(dirflag, w, i, par, g);
                        for (int c = 0; c < NCH; ++c)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int h = 0; h < 2; ++h) {
                                    const int blk = (lane >> 2) & 3, row = lane & 3, k = lane >> 4, m = 4 * q + blk;
                                    double v = 0.0;
                                    if (dirflag) {
                                        const int n = 2 * (4 * g + row) + par, lat = CP * c + 4 * h + k;
                                        if (n <= t.trunc && m + n <= t.trunc + 1) v = P(m, n, lat) * t.wt[lat];
                                    } else {
                                        const int n = 2 * (4 * g + k) + par, lat = CP * c + 4 * h + row;
                                        if (n < nx && m + n <= t.trunc + 1) v = P(m, n, lat);
                                    }
                                    (dirflag ? dir : inv)[((size_t)afrag(w, s, c) * 64 + lane) * 2 + h] = v;
                                }
                    }
        }
}

int upload_all(spdy_plan *p)
{
    HostTables &t = p->tab;
    DevPlan &d = p->dev;
    d.trunc = t.trunc; d.ix = t.ix; d.iy = t.iy; d.il = t.il; d.kx = t.kx; d.nx = t.nx; d.mx = t.mx;
    d.num_cu = p->num_cu;
    d.fs = (2 * t.mx + 15) / 16 * 16;
    d.ks_inv = ((t.nx + 1) / 2 + 3) / 4;
    d.jt = (t.iy + 15) / 16;
    d.nt_dir = ((t.nx + 1) / 2 + 15) / 16;
    d.js_dir = t.iy / 4;
    d.rgas = t.rgas;
    d.akap = t.akap;
    HIP_TRY(spdy::prepare_device_kernels());
    HIP_TRY(spdy::prepare_device_step_kernels(t.kx));
    std::vector<double> inv, dir;
    build_mfma_tables(t, d.ks_inv, d.jt, d.nt_dir, d.js_dir, inv, dir);
    int rc;
#define UP(vec, field) if ((rc = dev_upload(p, vec, &d.field))) return rc
    UP(inv, pa_inv); UP(dir, pa_dir); UP(t.cosgr, cosgr); UP(t.cosgr2, cosgr2); UP(t.coriol, coriol);
    d.pa_inv2 = d.pa_dir2 = d.img_s2g = d.img_g2s = d.img_s2g3 = nullptr;
    if (t.trunc == 30) {
        std::vector<double> inv2, dir2;
        build_packed_tables(t, d.ks_inv, d.js_dir, inv2, dir2);
        UP(inv2, pa_inv2); UP(dir2, pa_dir2);
        // register images of the fused kernels: wave w, slot s holds zonal wavenumber mslot(s, w) (spdy_fused_t30.inc)
        auto mslot = [](int sl, int w) { return (sl & 1) ? 4 * sl + 3 - w : 4 * sl + w; };
        std::vector<double> is2g((size_t)4 * 60 * 64), ig2s((size_t)4 * 72 * 64);
        for (int w = 0; w < 4; ++w)
            for (int lane = 0; lane < 64; ++lane) {
                int ai = 0;
                auto put = [&](std::vector<double> &img, int per_wave, double v) {
                    img[((size_t)(w * (per_wave / 2) + ai / 2) * 64 + lane) * 2 + (ai & 1)] = v;
                    ++ai;
                };
                for (int sl = 0; sl < 8; ++sl) {                                   // inverse: 3 fragments per k-step
                    const int mc = std::min(mslot(sl, w), t.mx - 1);
                    for (int ks = 0; ks < 4 - (sl >> 1); ++ks) {
                        put(is2g, 60, inv[(((size_t)mc * 2 + 0) * 4 + ks) * 2 * 64 + lane]);
                        put(is2g, 60, inv[(((size_t)mc * 2 + 1) * 4 + ks) * 2 * 64 + lane]);
                        put(is2g, 60, inv2[((size_t)mc * 4 + ks) * 64 + lane]);
                    }
                }
                ai = 0;
                for (int sl = 0; sl < 8; ++sl) {                                   // direct: 12 / 6 fragments per slot
                    const int mc = std::min(mslot(sl, w), t.mx - 1);
                    if (sl < 4) for (int q = 0; q < 12; ++q) put(ig2s, 72, dir[((size_t)mc * 2 * 6 + q) * 64 + lane]);
                    else for (int q = 0; q < 6; ++q) put(ig2s, 72, dir2[((size_t)mc * 6 + q) * 64 + lane]);
                }
            }
        UP(is2g, img_s2g); UP(ig2s, img_g2s);
        // small-batch form of the inverse kernel: per third of the latitudes one packed fragment per k-step (the construction of
        // inv2 -- which is the third 16..23 -- at latitude offsets 0, 8, 16)
        std::vector<double> is2g3((size_t)3 * 4 * 20 * 64, 0.0);
        auto P = [&](int m, int n, int j) { return t.poly[m + t.mx * (n + (size_t)t.nx * j)]; };
        for (int part = 0; part < 3; ++part)
            for (int w = 0; w < 4; ++w)
                for (int lane = 0; lane < 64; ++lane) {
                    int ai = 0;
                    for (int sl = 0; sl < 8; ++sl) {
                        const int mc = std::min(mslot(sl, w), t.mx - 1);
                        for (int ks = 0; ks < 4 - (sl >> 1); ++ks, ++ai) {
                            const int blk = (lane >> 2) & 3, i = lane & 3, k = lane >> 4, par = blk >> 1;
                            const int lat = 8 * part + 4 * (blk & 1) + i, n = 2 * (4 * ks + k) + par;
                            double v = 0.0;
                            if (lat < t.iy && n < t.nx && mc + n <= t.trunc + 1) v = P(mc, n, lat);
                            is2g3[((size_t)((part * 4 + w) * 10 + ai / 2) * 64 + lane) * 2 + (ai & 1)] = v;
                        }
                    }
                }
        UP(is2g3, img_s2g3);
    }
    d.img_g2s63 = d.img_s2g63 = nullptr;
    d.rows_ws = nullptr; d.rows_ws_fields = 0;
    if (t.trunc == 63) {
        std::vector<double> i63d, i63i;
        build_t63_images(t, i63d, i63i);
        UP(i63d, img_g2s63); UP(i63i, img_s2g63);
        // row workspace of the staged small-batch direct transform (spdy_fused_t63.inc): a model step's direct batch is up to
        // three segments of at most max_batch fields, and the form only runs below one pair per two CUs -- model-shaped plans
        // get room for all of it (98 KB per field), throughput-sized plans for the 256 fields such a launch can have
        const size_t fields = std::min<size_t>((size_t)3 * p->max_batch + 2, 258);
        void *ws = nullptr;
        if ((rc = dev_alloc(p, fields * t.il * 2 * t.mx * sizeof(double), &ws))) return rc;
        d.rows_ws = static_cast<double *>(ws);
        d.rows_ws_fields = (int)fields;
    }
    UP(t.el2, el2); UP(t.elm2, elm2); UP(t.trfilt, trfilt); UP(t.gradx, gradx); UP(t.gradym, gradym);
    UP(t.gradyp, gradyp); UP(t.uvdx, uvdx); UP(t.uvdym, uvdym); UP(t.uvdyp, uvdyp); UP(t.vddym, vddym);
    UP(t.vddyp, vddyp);
    {   // gradient tiles of the mixed inverse kernel: gradx indexed like the [nx][mx] tables, and one all-zero spectrum
        std::vector<double> gxe((size_t)t.mx * t.nx), zero((size_t)2 * t.mx * t.nx, 0.0);
        for (int n = 0; n < t.nx; ++n)
            for (int m = 0; m < t.mx; ++m) gxe[(size_t)n * t.mx + m] = t.gradx[m];
        UP(gxe, gradx_e);
        const double *z = nullptr;
        if ((rc = dev_upload(p, zero, &z))) return rc;
        p->d_zero_spec = z;
    }
#undef UP
    // FFT constants: FFTPACK `work` split by stage (fftpack.f90:45-66 layout)
    spdy::FftConstants fc;
    std::memset(&fc, 0, sizeof(fc));
    const int nfirst = (t.ifac[2] - 1) * 48;              // ido=48 stage: (ip-1) blocks of 48
    for (int i = 0; i < nfirst; ++i) fc.first[i] = t.work[i];
    for (int i = 0; i < 36; ++i) fc.a[i] = t.work[nfirst + i];
    for (int i = 0; i < 9; ++i) fc.b[i] = t.work[nfirst + 36 + i];
    fc.taui = t.taui; fc.sqrt2 = t.sqrt2; fc.hsqt2 = t.hsqt2; fc.scale = t.fwd_scale;
    HIP_TRY(spdy::upload_fft_constants(t.ix, fc));

    void *ptr;
    // the Fourier workspace and the host-API staging buffers are allocated on demand (ensure_four / ensure_staging):
    // a device-resident T30 host needs neither (at the bench configuration they would be 1 GB)
    // At T63 small plans (model-shaped batches: workspace <= 64 MB) get it right away, so that a graph capture needs no
    // warm-up call; a throughput-sized plan (1536 fields: 355 MB) allocates it at the first call that takes that path.
    {
        const size_t ws = (size_t)p->max_batch * (t.il * d.fs + 2 * spec_elems(p)) * sizeof(double);
        if (p->fused_mode == 0 || (t.trunc != 30 && ws <= ((size_t)64 << 20))) RC(ensure_four(p));
    }
    if ((rc = dev_alloc(p, sizeof(int) * (size_t)p->max_batch, &ptr))) return rc;
    p->d_kcos = static_cast<int *>(ptr);
    for (int i = 0; i < 6; ++i) {
        if ((rc = dev_alloc(p, sizeof(double) * t.mx * t.nx, &ptr))) return rc;
        p->d_dmp[i] = static_cast<double *>(ptr);
    }
    const std::vector<double> *src[3] = {&t.dmp, &t.dmpd, &t.dmps};
    for (int i = 0; i < 3; ++i)
        HIP_TRY(hipMemcpy(p->d_dmp[i], src[i]->data(), sizeof(double) * t.mx * t.nx, hipMemcpyHostToDevice));
    const int kx = t.kx;
    struct { double **dst; size_t n; } imp[7] = {{&p->d_xd, (size_t)kx * kx}, {&p->d_xc, (size_t)kx * kx},
                                                 {&p->d_xj, (size_t)kx * kx * (t.mx + t.nx + 1)},
                                                 {&p->d_tref1, (size_t)kx}, {&p->d_dhsx, (size_t)kx},
                                                 {&p->d_elz, (size_t)t.mx * t.nx},
                                                 {&p->d_levtab, (size_t)spdy::LEVTAB_COUNT * kx}};
    for (auto &e : imp) {
        if ((rc = dev_alloc(p, e.n * sizeof(double), &ptr))) return rc;
        *e.dst = static_cast<double *>(ptr);
    }
    {
        const int kxp = (kx + 1) & ~1;
        const size_t nl = (size_t)(t.mx + t.nx + 1);
        if ((rc = dev_alloc(p, (size_t)kx * kxp * (2 + nl) * sizeof(double), &ptr))) return rc;
        p->d_xt = static_cast<double *>(ptr);
        d.kxp = kxp;
        d.xdt = p->d_xt; d.xct = p->d_xt + (size_t)kx * kxp; d.xjt = p->d_xt + (size_t)2 * kx * kxp;
    }
    for (int i = 0; i < 6; ++i) d.dmp_t[i] = p->d_dmp[i];
    d.xd = p->d_xd; d.xc = p->d_xc; d.xj = p->d_xj; d.tref1 = p->d_tref1; d.dhsx = p->d_dhsx; d.elz = p->d_elz;
    const double *lt = p->d_levtab;
    d.dhs = lt; d.dhsr = lt + kx; d.fsgr = lt + 2 * kx; d.tref = lt + 3 * kx; d.tref2 = lt + 4 * kx; d.tref3 = lt + 5 * kx;
    d.rgtref = lt + 6 * kx; d.xgeop1 = lt + 7 * kx; d.xgeop2 = lt + 8 * kx; d.corf = lt + 9 * kx; d.tcorv = lt + 10 * kx;
    d.qcorv = lt + 11 * kx;
    return upload_level_tables(p);
}

}  // namespace

namespace spdy_detail {

// Per-level tables in DevPlan order.  Sigma-level functions are valid once sigma levels exist, tref* once
// spdy_implicit_init ran (zeros before).
int upload_level_tables(spdy_plan *p)
{
    if (p->device < 0) return SPDY_OK;
    const HostTables &t = p->tab;
    const int kx = t.kx;
    std::vector<double> h((size_t)spdy::LEVTAB_COUNT * kx, 0.0);
    auto put = [&](int slot, const std::vector<double> &v) {
        for (int k = 0; k < kx && k < (int)v.size(); ++k) h[(size_t)slot * kx + k] = v[k];
    };
    put(0, t.dhs); put(1, t.dhsr); put(2, t.fsgr); put(3, t.tref); put(4, t.tref2); put(5, t.tref3);
    for (int k = 0; k < kx && k < (int)t.tref.size(); ++k) h[(size_t)6 * kx + k] = t.rgas * t.tref[k];
    put(7, t.xgeop1); put(8, t.xgeop2); put(9, t.corf); put(10, t.tcorv); put(11, t.qcorv);
    HIP_TRY(hipMemcpy(p->d_levtab, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    return SPDY_OK;
}

int ensure_four(spdy_plan *p)
{
    if (p->four) return SPDY_OK;
    NOT_CAPTURING(p, "allocating the Fourier workspace (call spdy_plan_set_fused / one transform before the capture)");
    const size_t n = (size_t)p->max_batch * p->tab.il * p->dev.fs;
    void *ptr;
    RC(dev_alloc(p, n * sizeof(double), &ptr));
    p->four = static_cast<double *>(ptr);
    HIP_TRY(hipMemset(p->four, 0, n * sizeof(double)));
    // two spectral temporaries for the operator + transform sequences (uvspec/grad -> grid, grid -> vds)
    for (double **t : {&p->tmp_c, &p->tmp_d}) {
        RC(dev_alloc(p, (size_t)p->max_batch * spec_elems(p) * sizeof(double), &ptr));
        *t = static_cast<double *>(ptr);
    }
    return SPDY_OK;
}

int ensure_staging(spdy_plan *p, size_t elems)
{
    NOT_CAPTURING(p, "a host-pointer entry point");
    p->pending.clear();               // (a call that failed between its d2h and its sync must not leave copies behind)
    if (!p->dstage[0]) {
        // four buffers big enough for max_batch grids (the largest array kind)
        p->stage_elems = (size_t)p->max_batch * p->tab.il * p->tab.ix;
        void *ptr;
        for (int i = 0; i < 4; ++i) {
            RC(dev_alloc(p, p->stage_elems * sizeof(double), &ptr));
            p->dstage[i] = static_cast<double *>(ptr);
        }
        // the host-mapped twin set for small calls (spdy_plan::hstage); any failure here just leaves the route off
        size_t kb = 512;
        if (const char *env = getenv("SPDY_HOST_STAGE_KB")) kb = (size_t)(atol(env) > 0 ? atol(env) : 0);
        const size_t want = std::min(p->stage_elems, kb * 1024 / sizeof(double));
        if (want >= grid_elems(p)) {
            bool ok = true;
            for (int i = 0; i < 4 && ok; ++i) {
                ok = hipHostMalloc(&ptr, want * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
                if (ok) p->hstage[i] = static_cast<double *>(ptr);
            }
            if (ok) ok = hipHostMalloc(&ptr, sizeof(int) * (size_t)p->max_batch, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
            if (ok) {
                p->h_kcos = static_cast<int *>(ptr);
                p->hstage_elems = want;
                // the completion stamp is optional: without it sync() ends host-staged calls in hipStreamSynchronize
                if (hipHostMalloc(&ptr, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
                    p->h_stamp = static_cast<unsigned *>(ptr);
                    *p->h_stamp = 0;
                } else
                    (void)hipGetLastError();
            } else {
                // the route stays off as a whole: no buffers, no size, no kcos twin
                (void)hipGetLastError();
                for (int i = 0; i < 4; ++i) { if (p->hstage[i]) (void)hipHostFree(p->hstage[i]); p->hstage[i] = nullptr; }
                p->hstage_elems = 0;
            }
        }
    }
    double *const *set = (p->hstage_elems && elems <= p->hstage_elems) ? p->hstage : p->dstage;
    p->stage_a = set[0]; p->stage_b = set[1]; p->stage_c = set[2]; p->stage_d = set[3];
    return SPDY_OK;
}

int check_batch(const spdy_plan *p, int nb)
{
    if (nb < 0 || nb > p->max_batch) return fail(SPDY_ERR_ARG, "nb=%d outside [0, max_batch=%d]", nb, p->max_batch);
    return SPDY_OK;
}

static bool in_host_stage(const spdy_plan *p, const double *ptr)
{
    for (int i = 0; i < 4; ++i)
        if (p->hstage[i] && ptr >= p->hstage[i] && ptr < p->hstage[i] + p->hstage_elems) return true;
    return false;
}

int h2d(spdy_plan *p, double *dst, const double *src, size_t n)
{
    NOT_CAPTURING(p, "a host-pointer entry point");
    if (!n) return SPDY_OK;
    // host-mapped staging: nothing is in flight on it (every host-pointer call ends with sync), the kernels read it in place
    if (in_host_stage(p, dst)) std::memcpy(dst, src, n * sizeof(double));
    else HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, p->stream));
    return SPDY_OK;
}
int d2h(spdy_plan *p, double *dst, const double *src, size_t n)
{
    NOT_CAPTURING(p, "a host-pointer entry point");
    if (!n) return SPDY_OK;
    if (in_host_stage(p, src)) p->pending.push_back({dst, src, n * sizeof(double)});   // copied out by sync()
    else HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    return SPDY_OK;
}
// A one-thread kernel behind a host-staged call's kernels writes the call's sequence number into host-mapped memory; the calling
// thread spins on it instead of sleeping in hipStreamSynchronize (whose wake-up is the larger part of a 44 us one-field call).
__global__ void stamp_kernel(unsigned *stamp, unsigned seq)
{
    __threadfence_system();
    __hip_atomic_store(stamp, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int sync(spdy_plan *p)
{
    NOT_CAPTURING(p, "synchronising");
    // ($SPDY_HOST_SPIN=0: always hipStreamSynchronize.  Measured round 5, flang-built hosts: one-field round trips 22.0 -> 28.8 k/s
    // at T30, 15.9 -> 17.8 k/s at T63; stacks of eight 112 -> 125 k/s at T30.)
    static const bool spin = !(getenv("SPDY_HOST_SPIN") && atoi(getenv("SPDY_HOST_SPIN")) == 0);
    if (spin && p->h_stamp && !p->pending.empty()) {
        const unsigned seq = ++p->stamp_seq;
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, p->stream, p->h_stamp, seq);
        if (hipPeekAtLastError() == hipSuccess) {          // (peek: an earlier sticky error stays for hipStreamSynchronize below to report)
            // A bounded spin: a host-staged call's kernels take 10-60 us, so the stamp is normally there within a few hundred
            // microseconds; past 2 ms the thread stops burning its core (hosts with several ranks or OpenMP threads per core)
            // and sleeps in the runtime, which also reports an asynchronous kernel fault at once instead of after a timeout.
            const auto t0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (__atomic_load_n(p->h_stamp, __ATOMIC_ACQUIRE) != seq) {
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#elif defined(__aarch64__)
                asm volatile("yield" ::: "memory");
#else
                asm volatile("" ::: "memory");
#endif
                if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(2000)) break;   // fall back to the runtime
            }
            if (__atomic_load_n(p->h_stamp, __ATOMIC_ACQUIRE) == seq) {
                for (const auto &c : p->pending) std::memcpy(c.dst, c.src, c.bytes);
                p->pending.clear();
                return SPDY_OK;
            }
        }
    }
    const hipError_t e = hipStreamSynchronize(p->stream);
    if (e != hipSuccess) p->pending.clear();
    HIP_TRY(e);
    for (const auto &c : p->pending) std::memcpy(c.dst, c.src, c.bytes);
    p->pending.clear();
    return SPDY_OK;
}

}  // namespace spdy_detail

namespace {

// Fourier workspace rows are fs doubles apart; the reference layout packs 2*mx per row.
int four_to_host(spdy_plan *p, double *dst, int nb)
{
    NOT_CAPTURING(p, "a host-pointer entry point");
    const size_t w = 2 * (size_t)p->tab.mx * sizeof(double);
    if (nb) HIP_TRY(hipMemcpy2DAsync(dst, w, p->four, p->dev.fs * sizeof(double), w, (size_t)nb * p->tab.il,
                                     hipMemcpyDeviceToHost, p->stream));
    return SPDY_OK;
}
int four_from_host(spdy_plan *p, const double *src, int nb)
{
    NOT_CAPTURING(p, "a host-pointer entry point");
    const size_t w = 2 * (size_t)p->tab.mx * sizeof(double);
    if (nb) HIP_TRY(hipMemcpy2DAsync(p->four, p->dev.fs * sizeof(double), src, w, w, (size_t)nb * p->tab.il,
                                     hipMemcpyHostToDevice, p->stream));
    return SPDY_OK;
}

// Launch one transform kernel; when profiling is on, bracket it with HIP events recorded on
// the very stream it runs on (kind: SPDY_K_*).
// Fused single-pass kernels exist for T30.  Measured on MI355X (tools/small_batch.sh) they beat the
// four-kernel path at every batch size (one launch and no HBM intermediate: 2x even at 8..91 fields), so
// "auto" means fused whenever the resolution has them.
bool use_fused(const spdy_plan *p, int nb)
{
    (void)nb;
    return p->tab.trunc == 30 && p->fused_mode != 0;
}
// T63: fused field-pair kernels for the plain transforms (spdy_fused_t63.inc); the operator-fused modes use the
// multi-kernel sequences
// T63, direct transform: below ~80 fields a launch is one field pair per workgroup on a fraction of the CUs and costs the
// pair's pipeline latency (26 us, tools/t63_small_batch.py); the four-kernel pipeline spreads such a batch over more
// workgroups (21-24 us).  Until round 3 "auto" switched to it there -- and with it a field's BITS depended on whether it
// travelled in a batch of 79 or 80 (the two paths agree to rounding, not bitwise).  A drop-in must not do that to its host:
// auto now means the fused kernels at every batch size (the whole-pair walk accumulates every field the same way whatever
// the batch; the inverse transform's by-chunk walk for small batches is bit-identical to its whole-pair walk), at the price
// of ~4 us on a lone small direct transform.  spdy_plan_set_fused(0) still selects the four-kernel path.
bool use_fused63(const spdy_plan *p, int nb) { (void)nb; return p->tab.trunc == 63 && p->fused_mode != 0; }
// The composite entry points (uvspec/grad -> grid, vdspec, the mixed batches) are one fused launch against two to four
// four-kernel sequences: the fused kernels win there at any size.
bool use_fused63_composite(const spdy_plan *p) { return p->tab.trunc == 63 && p->fused_mode != 0; }
// T30: a step's whole direct batch is ONE mixed-tile launch (MODE 3) -- what a latency-bound, model-sized batch wants.  At
// throughput sizes (the launch streams: >= 16 MB of grids) the launch count does not matter and the mixed-mode direct kernel --
// out of registers, it reads its vds factors where it uses them -- runs at 0.48 of the HBM roofline where the pair kernel
// (0.58) and the plain kernel (0.69) on their own average 0.54 (round 6, same box: 0.485 -> 0.541): such batches go out as the
// two launches.  The INVERSE mixed launch keeps its one launch at every size (0.58 against 0.56 split, measured the same day).
// A field's bits do not depend on the launch it travels in (tests/test_gpu_determinism.py, test_gpu_fused_ops.py).
bool one_mixed_launch(const spdy_plan *p, long fields)
{
    return !spdy::streams(fields * (long)grid_elems(p) * 8);
}

template <class F> int timed(spdy_plan *p, int kind, F &&launch)
{
    if (!p->profiling || p->capturing) { HIP_TRY(launch()); return SPDY_OK; }
    spdy_plan::Span sp{kind, nullptr, nullptr};
    HIP_TRY(hipEventCreate(&sp.t0));
    HIP_TRY(hipEventCreate(&sp.t1));
    HIP_TRY(hipEventRecord(sp.t0, p->stream));
    HIP_TRY(launch());
    HIP_TRY(hipEventRecord(sp.t1, p->stream));
    p->spans.push_back(sp);
    return SPDY_OK;
}

}  // namespace

namespace spdy_detail {
bool use_raw63(const spdy_plan *p, int npairs)
{
    return p->tab.trunc == 63 && p->fused_mode != 0 && p->tab.kx <= 16 && npairs <= p->max_batch && p->tab.implicit_ready && p->tab.sigma_ready;
}

int direct_batch_raw63(spdy_plan *p, int npairs, const double *ug, const double *vg, int kcos, int nplain, const double *grid, double *spec,
                       double *raw_u, double *raw_v)
{
    RC(check_batch(p, npairs));
    RC(check_batch(p, nplain));
    if (!raw_u || !raw_v) {
        RC(ensure_four(p));
        raw_u = p->tmp_c; raw_v = p->tmp_d;
    }
    const double *sc = kcos == 2 ? p->dev.cosgr : p->dev.cosgr2;
    spdy::T63Batch b{};
    b.nseg = 3;
    b.seg[0] = spdy::T63Seg{ug, raw_u, sc, nullptr, npairs, 1, 0, 0};
    b.seg[1] = spdy::T63Seg{vg, raw_v, sc, nullptr, npairs, 1, 0, 0};
    b.seg[2] = spdy::T63Seg{grid, spec, nullptr, nullptr, nplain, 1, 0, 0};
    return timed(p, SPDY_K_G2S_FUSED, [&] { return spdy::launch_g2s_fused_t63_batch(p->dev, b, p->num_cu, p->stream); });
}
}  // namespace spdy_detail

extern "C" {

const char *spdy_last_error(void) { return g_err.c_str(); }

int spdy_plan_create(int trunc, int ix, int iy, int kx, int max_batch, int device, spdy_plan **plan)
{
    if (!plan) return fail(SPDY_ERR_ARG, "null plan pointer");
    *plan = nullptr;
    if (max_batch < 1) return fail(SPDY_ERR_ARG, "max_batch must be >= 1");
    if (!((trunc == 30 && ix == 96 && iy == 24) || (trunc == 63 && ix == 192 && iy == 48)))
        return fail(SPDY_ERR_UNSUPPORTED, "kernels are built for T30 (96x48) and T63 (192x96); got trunc=%d ix=%d iy=%d",
                    trunc, ix, iy);
    if (kx < 1 || kx > SPDY_MAX_KX) return fail(SPDY_ERR_UNSUPPORTED, "kx=%d outside 1..%d", kx, (int)SPDY_MAX_KX);
    spdy_plan *p = new spdy_plan;
    const std::string err = p->tab.build(trunc, ix, iy, kx);
    if (!err.empty()) {
        delete p;
        return fail(SPDY_ERR_TABLE, "table generation: %s", err.c_str());
    }
    p->max_batch = max_batch;
    if (device == SPDY_DEVICE_AUTO) {
        // one process per GPU: $SPDY_DEVICE wins, else the launcher's local rank modulo the visible devices
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) ndev = 1;
        device = 0;
        if (const char *env = getenv("SPDY_DEVICE")) device = atoi(env);
        else
            for (const char *name : {"LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID", "MPI_LOCALRANKID"})
                if (const char *v = getenv(name)) { device = atoi(v) % ndev; break; }
        if (device < 0) device = 0;
    } else if (device < SPDY_DEVICE_AUTO) {
        delete p;
        return fail(SPDY_ERR_ARG, "device=%d: use a device index, SPDY_DEVICE_AUTO or SPDY_DEVICE_NONE", device);
    }
    p->device = device;
    if (device >= 0) {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || device >= ndev) {
            delete p;
            return fail(SPDY_ERR_NO_DEVICE, "HIP device %d not available (%s, %d visible); there is no CPU fallback",
                        device, hipGetErrorString(e), ndev);
        }
        int rc = SPDY_OK;
        if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreate(&p->own_stream)) != hipSuccess)
            rc = fail(SPDY_ERR_HIP, "device init: %s", hipGetErrorString(e));
        p->stream = p->own_stream;
        if (!rc) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
                p->num_cu = prop.multiProcessorCount;
            if (const char *env = getenv("SPDY_FUSED")) p->fused_mode = atoi(env);
            if (const char *env = getenv("SPDY_WG_PER_CU")) p->wg_per_cu = atoi(env) >= 1 ? atoi(env) : 1;
            // launch-policy switches: the environment is read HERE, once per plan; afterwards spdy_plan_set_option
            spdy::LaunchOpts &lo = p->dev.lo;
            lo.t30_nopart = getenv("SPDY_T30_NOPART") != nullptr;
            lo.t30_nosplit = getenv("SPDY_T30_NOSPLIT") != nullptr;
            lo.t63_nosplit = getenv("SPDY_T63_NOSPLIT") != nullptr;
            lo.t63_nostage = getenv("SPDY_T63_NOSTAGE") != nullptr;
            if (const char *env = getenv("SPDY_T63_NP2_FROM")) lo.t63_np2_from = atoi(env);
            if (const char *env = getenv("SPDY_WT_MIN_MB")) lo.wt_min_mb = atoi(env) > 0 ? atoi(env) : 0;
            p->t63_derive = getenv("SPDY_T63_NODERIVE") == nullptr;
        }
        if (!rc) rc = upload_all(p);
        if (rc) {
            const std::string keep = g_err;
            spdy_plan_destroy(p);
            g_err = keep;
            return rc;
        }
    }
    *plan = p;
    return SPDY_OK;
}

int spdy_plan_destroy(spdy_plan *p)
{
    if (!p) return SPDY_OK;
    // graphs captured from this plan point at its workspaces and tables: they die with it (spdy_graph_launch then
    // returns SPDY_ERR_STATE; spdy_graph_destroy still frees the handle)
    for (spdy_graph *g : p->graphs) {
        if (g->exec) (void)hipGraphExecDestroy(g->exec);
        if (g->graph) (void)hipGraphDestroy(g->graph);
        g->exec = nullptr; g->graph = nullptr; g->plan = nullptr;
    }
    p->graphs.clear();
    if (p->device >= 0) {
        (void)hipSetDevice(p->device);
        if (p->capturing) { hipGraph_t dead = nullptr; (void)hipStreamEndCapture(p->stream, &dead); if (dead) (void)hipGraphDestroy(dead); }
        if (p->stream) (void)hipStreamSynchronize(p->stream);
        // communicators enqueue on this plan's stream and read its dimensions: they are shut down with it (the handles stay
        // valid for spdy_comm_destroy; every other call on them returns SPDY_ERR_STATE)
        release_comms(p);
        for (void *a : p->allocs) (void)hipFree(a);
        for (double *h : p->hstage) if (h) (void)hipHostFree(h);
        if (p->h_stamp) (void)hipHostFree(p->h_stamp);
        if (p->h_kcos) (void)hipHostFree(p->h_kcos);
        if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
    }
    delete p;
    return SPDY_OK;
}

int spdy_plan_set_stream(spdy_plan *p, void *hip_stream)
{
    NEED_DEVICE(p);
    if (p->capturing) return fail(SPDY_ERR_STATE, "cannot change the stream while a graph capture is open");
    p->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : p->own_stream;
    return SPDY_OK;
}

int spdy_plan_synchronize(spdy_plan *p)
{
    NEED_DEVICE(p);
    return sync(p);
}

/* ---- device memory for hosts without a HIP binding of their own (a Fortran model: fortran/time_stepping.f90) ---- */
int spdy_dev_alloc(spdy_plan *p, size_t bytes, void **d_ptr)
{
    NEED_DEVICE(p);
    NOT_CAPTURING(p, "spdy_dev_alloc");
    if (!d_ptr) return fail(SPDY_ERR_ARG, "null pointer");
    *d_ptr = nullptr;
    if (!bytes) return SPDY_OK;
    // recorded with the plan's own allocations: released by spdy_dev_free, or with the plan (a host that finalises its
    // modules in the "wrong" order -- the plan before the state that lives in these buffers -- leaks nothing)
    void *ptr = nullptr;
    RC(dev_alloc(p, bytes, &ptr));
    int rc = SPDY_OK;
    if (hipMemsetAsync(ptr, 0, bytes, p->stream) != hipSuccess) rc = fail(SPDY_ERR_HIP, "hipMemsetAsync failed");
    if (!rc) rc = sync(p);
    if (rc) {
        p->allocs.pop_back();
        (void)hipFree(ptr);
        return rc;
    }
    *d_ptr = ptr;
    return SPDY_OK;
}

int spdy_dev_free(spdy_plan *p, void *d_ptr)
{
    if (!d_ptr) return SPDY_OK;
    NEED_DEVICE(p);
    NOT_CAPTURING(p, "spdy_dev_free");
    for (auto it = p->allocs.begin(); it != p->allocs.end(); ++it)
        if (*it == d_ptr) {
            RC(sync(p));               // work queued on the plan's stream may still use it
            p->allocs.erase(it);
            HIP_TRY(hipFree(d_ptr));
            return SPDY_OK;
        }
    return fail(SPDY_ERR_ARG, "spdy_dev_free: not a live spdy_dev_alloc buffer of this plan");
}

int spdy_dev_upload(spdy_plan *p, void *d_dst, const void *src, size_t bytes)
{
    NEED_DEVICE(p);
    NOT_CAPTURING(p, "spdy_dev_upload");
    if (bytes && (!d_dst || !src)) return fail(SPDY_ERR_ARG, "null pointer");
    if (bytes) HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, p->stream));
    return sync(p);
}

int spdy_dev_download(spdy_plan *p, void *dst, const void *d_src, size_t bytes)
{
    NEED_DEVICE(p);
    NOT_CAPTURING(p, "spdy_dev_download");
    if (bytes && (!dst || !d_src)) return fail(SPDY_ERR_ARG, "null pointer");
    if (bytes) HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, p->stream));
    return sync(p);
}

int spdy_plan_set_profiling(spdy_plan *p, int on)
{
    NEED_DEVICE(p);
    p->profiling = on != 0;
    return SPDY_OK;
}

int spdy_plan_set_fused(spdy_plan *p, int mode)
{
    NEED_PLAN(p);
    if (mode < -1 || mode > 1) return fail(SPDY_ERR_ARG, "fused mode must be -1, 0 or 1");
    if (mode == 0 && p->device >= 0) {
        HIP_TRY(hipSetDevice(p->device));
        RC(ensure_four(p));
    }
    p->fused_mode = mode;
    return SPDY_OK;
}

int spdy_plan_set_option(spdy_plan *p, const char *name, int value)
{
    NEED_PLAN(p);
    if (!name) return fail(SPDY_ERR_ARG, "null option name");
    if (p->capturing) return fail(SPDY_ERR_STATE, "launch options cannot change while a graph capture is open");
    spdy::LaunchOpts &lo = p->dev.lo;
    const std::string n(name);
    if (n == "t30_part") lo.t30_nopart = !value;
    else if (n == "t30_split") lo.t30_nosplit = !value;
    else if (n == "t63_split") lo.t63_nosplit = !value;
    else if (n == "t63_stage") lo.t63_nostage = !value;
    else if (n == "t63_derive") p->t63_derive = value != 0;
    else if (n == "t63_np2_from") { if (value < 1) return fail(SPDY_ERR_ARG, "t63_np2_from must be >= 1"); lo.t63_np2_from = value; }
    else if (n == "wt_min_mb") { if (value < 0) return fail(SPDY_ERR_ARG, "wt_min_mb must be >= 0"); lo.wt_min_mb = value; }
    else return fail(SPDY_ERR_ARG, "unknown launch option '%s'", name);
    return SPDY_OK;
}

int spdy_plan_get_profile(spdy_plan *p, double *ms, int *launches)
{
    NEED_DEVICE(p);
    if (!ms || !launches) return fail(SPDY_ERR_ARG, "null output");
    if (p->capturing) return fail(SPDY_ERR_STATE, "cannot read the profile while a graph capture is open");
    for (int k = 0; k < SPDY_K_COUNT; ++k) { ms[k] = 0.0; launches[k] = 0; }
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (auto &sp : p->spans) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, sp.t0, sp.t1));
        ms[sp.kind] += t;
        launches[sp.kind] += 1;
        (void)hipEventDestroy(sp.t0);
        (void)hipEventDestroy(sp.t1);
    }
    p->spans.clear();
    return SPDY_OK;
}

int spdy_wave_placement(spdy_plan *p, int *simd_of_wave, int *violations)
{
    NEED_DEVICE(p);
    NOT_CAPTURING(p, "spdy_wave_placement");
    if (!simd_of_wave || !violations) return fail(SPDY_ERR_ARG, "null output");
    const int nwg = p->num_cu;
    void *ptr = nullptr;
    HIP_TRY(hipMalloc(&ptr, sizeof(int) * 8 * (size_t)nwg));
    std::vector<int> h(8 * (size_t)nwg, -1);
    hipError_t e = spdy::launch_wave_placement(static_cast<int *>(ptr), nwg, p->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), ptr, sizeof(int) * h.size(), hipMemcpyDeviceToHost, p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    (void)hipFree(ptr);
    HIP_TRY(e);
    int bad = 0;
    for (int g = 0; g < nwg; ++g) {
        const int *w = &h[8 * (size_t)g];
        bool ok = true;
        for (int i = 0; i < 4; ++i) ok = ok && w[i] == w[i + 4] && w[i] >= 0 && w[i] < 4;
        for (int i = 0; i < 4 && ok; ++i)
            for (int j = i + 1; j < 4; ++j) ok = ok && w[i] != w[j];
        bad += ok ? 0 : 1;
    }
    std::memcpy(simd_of_wave, h.data(), sizeof(int) * 8);
    *violations = bad;
    return SPDY_OK;
}

int spdy_plan_dims(const spdy_plan *p, int *dims)
{
    NEED_PLAN(p);
    if (!dims) return fail(SPDY_ERR_ARG, "null dims");
    const HostTables &t = p->tab;
    const int v[8] = {t.trunc, t.ix, t.iy, t.il, t.kx, t.nx, t.mx, p->max_batch};
    std::memcpy(dims, v, sizeof(v));
    return SPDY_OK;
}

int spdy_get_table(const spdy_plan *p, const char *name, double *buf, int cap)
{
    NEED_PLAN(p);
    if (!name) return fail(SPDY_ERR_ARG, "null table name");
    std::vector<double> scratch;
    int n = 0;
    const double *src = p->tab.lookup(name, &n, scratch);
    if (!src) return fail(SPDY_ERR_ARG, "unknown table '%s'", name);
    if (buf && cap > 0) std::memcpy(buf, src, sizeof(double) * (size_t)(n < cap ? n : cap));
    return n;
}

/* ---------------------------------------------------------------- transforms, device pointers */
int spdy_spec_to_grid_dev(spdy_plan *p, int nb, const double *d_spec, const int *d_kcos, int kcos_all, double *d_grid)
{
    NEED_DEVICE(p);
    RC(check_batch(p, nb));
    if (nb && (!d_spec || !d_grid)) return fail(SPDY_ERR_ARG, "null device pointer");
    if (use_fused(p, nb))
        return timed(p, SPDY_K_S2G_FUSED, [&] {
            return spdy::launch_s2g_fused(p->dev, nb, d_spec, d_kcos, kcos_all, d_grid, p->num_cu * p->wg_per_cu, p->stream);
        });
    if (use_fused63_composite(p))               // (the inverse kernel serves every batch size: by-chunk items for small ones)
        return timed(p, SPDY_K_S2G_FUSED, [&] {
            return spdy::launch_s2g_fused_t63(p->dev, nb, d_spec, d_kcos, kcos_all, d_grid, p->num_cu, p->stream);
        });
    RC(ensure_four(p));
    RC(timed(p, SPDY_K_LEGENDRE_INV, [&] { return spdy::launch_legendre_inv(p->dev, nb, d_spec, p->four, p->stream); }));
    RC(timed(p, SPDY_K_FOURIER_INV, [&] { return spdy::launch_fourier_inv(p->dev, nb, p->four, d_kcos, kcos_all, d_grid, p->stream); }));
    return SPDY_OK;
}

int spdy_grid_to_spec_dev(spdy_plan *p, int nb, const double *d_grid, double *d_spec)
{
    NEED_DEVICE(p);
    RC(check_batch(p, nb));
    if (nb && (!d_spec || !d_grid)) return fail(SPDY_ERR_ARG, "null device pointer");
    if (use_fused(p, nb))
        return timed(p, SPDY_K_G2S_FUSED, [&] {
            return spdy::launch_g2s_fused(p->dev, nb, d_grid, nullptr, d_spec, p->num_cu * p->wg_per_cu, p->stream, nullptr, nullptr, 0, nullptr,
                                          nullptr, !in_host_stage(p, d_grid));
        });
    if (use_fused63(p, nb))
        return timed(p, SPDY_K_G2S_FUSED, [&] { return spdy::launch_g2s_fused_t63(p->dev, nb, d_grid, nullptr, d_spec, p->num_cu, p->stream); });
    RC(ensure_four(p));
    RC(timed(p, SPDY_K_FOURIER_DIR, [&] { return spdy::launch_fourier_dir(p->dev, nb, d_grid, nullptr, p->four, p->stream); }));
    RC(timed(p, SPDY_K_LEGENDRE_DIR, [&] { return spdy::launch_legendre_dir(p->dev, nb, p->four, d_spec, p->stream); }));
    return SPDY_OK;
}

/* ---------------------------------------------------------------- transforms, host pointers */
int spdy_spec_to_grid_batch(spdy_plan *p, int nb, const double *spec, const int *kcos, double *grid)
{
    NEED_DEVICE(p);
    RC(check_batch(p, nb));
    RC(ensure_staging(p, nb * grid_elems(p)));
    if (nb && (!spec || !grid)) return fail(SPDY_ERR_ARG, "null pointer");
    RC(h2d(p, p->stage_a, spec, nb * spec_elems(p)));
    // one flag for the whole batch (always so for the reference's one-field calls): passed by value, no array to move
    bool same = true;
    for (int i = 1; kcos && i < nb; ++i) same = same && kcos[i] == kcos[0];
    const int *dk = nullptr;
    if (kcos && nb && !same) {
        if (on_host_stage(p)) { std::memcpy(p->h_kcos, kcos, sizeof(int) * nb); dk = p->h_kcos; }
        else { HIP_TRY(hipMemcpyAsync(p->d_kcos, kcos, sizeof(int) * nb, hipMemcpyHostToDevice, p->stream)); dk = p->d_kcos; }
    }
    RC(spdy_spec_to_grid_dev(p, nb, p->stage_a, dk, (kcos && nb) ? kcos[0] : 1, p->stage_b));
    RC(d2h(p, grid, p->stage_b, nb * grid_elems(p)));
    return sync(p);
}

int spdy_grid_to_spec_batch(spdy_plan *p, int nb, const double *grid, double *spec)
{
    NEED_DEVICE(p);
    RC(check_batch(p, nb));
    RC(ensure_staging(p, nb * grid_elems(p)));
    if (nb && (!spec || !grid)) return fail(SPDY_ERR_ARG, "null pointer");
    RC(h2d(p, p->stage_a, grid, nb * grid_elems(p)));
    RC(spdy_grid_to_spec_dev(p, nb, p->stage_a, p->stage_b));
    RC(d2h(p, spec, p->stage_b, nb * spec_elems(p)));
    return sync(p);
}

int spdy_spec_to_grid(spdy_plan *p, const double *spec, int kcos, double *grid)
{
    return spdy_spec_to_grid_batch(p, 1, spec, &kcos, grid);
}

int spdy_grid_to_spec(spdy_plan *p, const double *grid, double *spec)
{
    return spdy_grid_to_spec_batch(p, 1, grid, spec);
}

/* ---------------------------------------------------------------- stages, host pointers */
int spdy_legendre_inv(spdy_plan *p, int nb, const double *spec, double *four)
{
    NEED_DEVICE(p);
    RC(check_batch(p, nb));
    RC(ensure_staging(p));
    RC(ensure_four(p));
    if (nb && (!spec || !four)) return fail(SPDY_ERR_ARG, "null pointer");
    RC(h2d(p, p->stage_a, spec, nb * spec_elems(p)));
    KERNEL(spdy::launch_legendre_inv(p->dev, nb, p->stage_a, p->four, p->stream));
    RC(four_to_host(p, four, nb));
    RC(sync(p));
    // Im(m'=0) is produced as an exact 0 +/- 0 by the contraction; nothing to fix up.
    return SPDY_OK;
}

int spdy_legendre_dir(spdy_plan *p, int nb, const double *four, double *spec)
{
    NEED_DEVICE(p);
    RC(check_batch(p, nb));
    RC(ensure_staging(p));
    RC(ensure_four(p));
    if (nb && (!spec || !four)) return fail(SPDY_ERR_ARG, "null pointer");
    RC(four_from_host(p, four, nb));
    KERNEL(spdy::launch_legendre_dir(p->dev, nb, p->four, p->stage_a, p->stream));
    RC(d2h(p, spec, p->stage_a, nb * spec_elems(p)));
    return sync(p);
}

int spdy_fourier_inv(spdy_plan *p, int nb, const double *four, int kcos, double *grid)
{
    NEED_DEVICE(p);
    RC(check_batch(p, nb));
    RC(ensure_staging(p));
    RC(ensure_four(p));
    if (nb && (!grid || !four)) return fail(SPDY_ERR_ARG, "null pointer");
    RC(four_from_host(p, four, nb));
    KERNEL(spdy::launch_fourier_inv(p->dev, nb, p->four, nullptr, kcos, p->stage_a, p->stream));
    RC(d2h(p, grid, p->stage_a, nb * grid_elems(p)));
    return sync(p);
}

int spdy_fourier_dir(spdy_plan *p, int nb, const double *grid, double *four)
{
    NEED_DEVICE(p);
    RC(check_batch(p, nb));
    RC(ensure_staging(p));
    RC(ensure_four(p));
    if (nb && (!grid || !four)) return fail(SPDY_ERR_ARG, "null pointer");
    RC(h2d(p, p->stage_a, grid, nb * grid_elems(p)));
    KERNEL(spdy::launch_fourier_dir(p->dev, nb, p->stage_a, nullptr, p->four, p->stream));
    RC(four_to_host(p, four, nb));
    return sync(p);
}

/* ---------------------------------------------------------------- spectral operators */
int spdy_laplacian_dev(spdy_plan *p, int nb, const double *in, double *out)
{
    NEED_DEVICE(p);
    KERNEL(spdy::launch_scale_op(p->dev, spdy::OP_LAPLACIAN, nb, in, out, p->stream));
    return SPDY_OK;
}
int spdy_inverse_laplacian_dev(spdy_plan *p, int nb, const double *in, double *out)
{
    NEED_DEVICE(p);
    KERNEL(spdy::launch_scale_op(p->dev, spdy::OP_INV_LAPLACIAN, nb, in, out, p->stream));
    return SPDY_OK;
}
int spdy_trunct_dev(spdy_plan *p, int nb, double *inout)
{
    NEED_DEVICE(p);
    KERNEL(spdy::launch_scale_op(p->dev, spdy::OP_TRUNCT, nb, inout, inout, p->stream));
    return SPDY_OK;
}
int spdy_grad_dev(spdy_plan *p, int nb, const double *psi, double *psdx, double *psdy)
{
    NEED_DEVICE(p);
    KERNEL(spdy::launch_grad(p->dev, nb, psi, psdx, psdy, p->stream));
    return SPDY_OK;
}
int spdy_vds_dev(spdy_plan *p, int nb, const double *u, const double *v, double *vor, double *dv)
{
    NEED_DEVICE(p);
    KERNEL(spdy::launch_vds(p->dev, nb, u, v, vor, dv, p->stream));
    return SPDY_OK;
}
int spdy_uvspec_dev(spdy_plan *p, int nb, const double *vor, const double *dv, double *u, double *v)
{
    NEED_DEVICE(p);
    KERNEL(spdy::launch_uvspec(p->dev, nb, vor, dv, u, v, p->stream));
    return SPDY_OK;
}
/* T63, row f1: uvspec / grad evaluated where the fused inverse kernel loads its operands (csrc/spdy_fused_t63.inc,
 * t63_inv_load_b_op) -- segments U, V of (vor, div) and d/dlambda, d/dmu of psi.  Model-sized launches only; larger ones keep
 * the operator kernel in front (it runs at 4-6 TB/s there and the transform launch is not latency-bound).  $SPDY_T63_NODERIVE /
 * spdy_plan_set_option("t63_derive", 0): always the operator kernel (A/B runs, the determinism test).                        */
static int append_uv_segs(spdy::T63Batch &b, int k, int npairs, const double *vor, const double *dv, double *ug, double *vg, int kcos)
{
    b.seg[k++] = spdy::T63Seg{vor, ug, dv, nullptr, npairs, kcos, 0, spdy::T63_OP_U};
    b.seg[k++] = spdy::T63Seg{dv, vg, vor, nullptr, npairs, kcos, 0, spdy::T63_OP_V};
    return k;
}
static int append_grad_segs(spdy::T63Batch &b, int k, int ngrad, const double *psi, double *gx, double *gy, int kcos)
{
    b.seg[k++] = spdy::T63Seg{psi, gx, psi, nullptr, ngrad, kcos, 0, spdy::T63_OP_GX};
    b.seg[k++] = spdy::T63Seg{psi, gy, psi, nullptr, ngrad, kcos, 0, spdy::T63_OP_GY};
    return k;
}
static bool derive63(const spdy_plan *p, int op_fields_each, int nsets, int plain_fields_pairs)
{
    // op_fields_each fields in each of nsets derived segments (pairs are formed inside a segment)
    const int op_pairs = nsets * ((op_fields_each + 1) / 2);
    return p->t63_derive && spdy::s2g_t63_derives(p->num_cu, op_pairs + plain_fields_pairs, op_pairs);
}

/* uvspec / grad followed by the two inverse transforms their callers always do, in one pass where the fused
 * kernels exist; otherwise the operator kernel into plan-owned temporaries and two ordinary transforms.      */
static int derived_to_grid(spdy_plan *p, int nb, int mode, const double *in0, const double *in1, double *g0, double *g1, int kcos)
{
    NEED_DEVICE(p);
    RC(check_batch(p, nb));
    if (nb && (!in0 || (mode == 1 && !in1) || !g0 || !g1)) return fail(SPDY_ERR_ARG, "null device pointer");
    kcos = kcos == 1 ? 1 : 2;   // fourier.f90:47-51: anything but 1 means "times cosgr"
    if (use_fused(p, nb))
        return timed(p, SPDY_K_S2G_FUSED, [&] {
            return spdy::launch_s2g_fused(p->dev, nb, in0, nullptr, kcos, g0, p->num_cu * p->wg_per_cu, p->stream, mode, in1, g1);
        });
    if (use_fused63_composite(p) && derive63(p, nb, 2, 0)) {   // T63, model-sized: the operator rides in the transform launch
        spdy::T63Batch b{};
        b.nseg = mode == 1 ? append_uv_segs(b, 0, nb, in0, in1, g0, g1, kcos) : append_grad_segs(b, 0, nb, in0, g0, g1, kcos);
        return timed(p, SPDY_K_S2G_FUSED, [&] { return spdy::launch_s2g_fused_t63_batch(p->dev, b, p->num_cu, p->stream); });
    }
    RC(ensure_four(p));
    if (mode == 1) KERNEL(spdy::launch_uvspec(p->dev, nb, in0, in1, p->tmp_c, p->tmp_d, p->stream));
    else {
        // rows of psdy that grad leaves untouched (l > trunc+1 inside row nx) are never read by the transform
        KERNEL(spdy::launch_grad(p->dev, nb, in0, p->tmp_c, p->tmp_d, p->stream));
    }
    if (use_fused63_composite(p)) {             // both derived spectra in ONE fused launch (two segments)
        spdy::T63Batch b{};
        b.nseg = 2;
        b.seg[0] = spdy::T63Seg{p->tmp_c, g0, nullptr, nullptr, nb, kcos, 0, 0};
        b.seg[1] = spdy::T63Seg{p->tmp_d, g1, nullptr, nullptr, nb, kcos, 0, 0};
        return timed(p, SPDY_K_S2G_FUSED, [&] { return spdy::launch_s2g_fused_t63_batch(p->dev, b, p->num_cu, p->stream); });
    }
    RC(spdy_spec_to_grid_dev(p, nb, p->tmp_c, nullptr, kcos, g0));
    RC(spdy_spec_to_grid_dev(p, nb, p->tmp_d, nullptr, kcos, g1));
    return SPDY_OK;
}
int spdy_uvspec_to_grid_dev(spdy_plan *p, int nb, const double *vor, const double *dv, double *ug, double *vg, int kcos)
{
    return derived_to_grid(p, nb, 1, vor, dv, ug, vg, kcos);
}
int spdy_grad_to_grid_dev(spdy_plan *p, int nb, const double *psi, double *gx, double *gy, int kcos)
{
    return derived_to_grid(p, nb, 2, psi, nullptr, gx, gy, kcos);
}

/* host-pointer forms (level stacks of the Fortran host): inputs to stage_a/b, grids come back from stage_c/d */
static int derived_to_grid_host(spdy_plan *p, int nb, int mode, const double *in0, const double *in1, double *g0, double *g1, int kcos)
{
    NEED_DEVICE(p);
    RC(check_batch(p, nb));
    if (nb && (!in0 || (mode == 1 && !in1) || !g0 || !g1)) return fail(SPDY_ERR_ARG, "null pointer");
    RC(ensure_staging(p, nb * grid_elems(p)));
    RC(h2d(p, p->stage_a, in0, nb * spec_elems(p)));
    if (mode == 1) RC(h2d(p, p->stage_b, in1, nb * spec_elems(p)));
    RC(derived_to_grid(p, nb, mode, p->stage_a, p->stage_b, p->stage_c, p->stage_d, kcos));
    RC(d2h(p, g0, p->stage_c, nb * grid_elems(p)));
    RC(d2h(p, g1, p->stage_d, nb * grid_elems(p)));
    return sync(p);
}
int spdy_uvspec_to_grid(spdy_plan *p, int nb, const double *vor, const double *dv, double *ug, double *vg, int kcos)
{
    return derived_to_grid_host(p, nb, 1, vor, dv, ug, vg, kcos);
}
int spdy_grad_to_grid(spdy_plan *p, int nb, const double *psi, double *gx, double *gy, int kcos)
{
    return derived_to_grid_host(p, nb, 2, psi, nullptr, gx, gy, kcos);
}

/* vdspec: scale on load, two direct transforms, then vds.  The multi-kernel path keeps the two intermediate
 * spectra in plan-owned temporaries, so ug/vg/vorm/divm may be the caller's own device buffers.       */
int spdy_vdspec_dev(spdy_plan *p, int nb, const double *ug, const double *vg, double *vorm, double *divm, int kcos)
{
    NEED_DEVICE(p);
    RC(check_batch(p, nb));
    if (nb && (!ug || !vg || !vorm || !divm)) return fail(SPDY_ERR_ARG, "null device pointer");
    const double *sc = kcos == 2 ? p->dev.cosgr : p->dev.cosgr2;
    if (use_fused(p, nb)) {
        // one pass: the pair (ug[i], vg[i]) is one tile, vds is applied to the two spectra while they are in LDS
        KERNEL(spdy::launch_g2s_fused(p->dev, nb, ug, sc, vorm, p->num_cu * p->wg_per_cu, p->stream, vg, divm, 0, nullptr, nullptr,
                                      !in_host_stage(p, ug)));
        return SPDY_OK;
    }
    if (use_fused63_composite(p) && spdy::g2s_t63_staged(p->dev, p->num_cu, nb)) {
        // T63, model-sized (row f1, round 6): the pairs (ug[i], vg[i]) as ONE segment of the staged form -- rows launch, then the
        // contraction applies vds to the pair's spectra in registers: no vds launch, no raw spectra in HBM
        spdy::T63Batch b{};
        b.nseg = 1;
        b.seg[0] = spdy::T63Seg{ug, vorm, sc, nullptr, nb, 1, 0, spdy::T63_OP_VDS};
        b.vds_src2 = vg; b.vds_dst2 = divm;
        return timed(p, SPDY_K_G2S_FUSED, [&] { return spdy::launch_g2s_fused_t63_batch(p->dev, b, p->num_cu, p->stream); });
    }
    if (use_fused63_composite(p)) {             // both scaled transforms in ONE fused launch (two segments) + vds
        RC(ensure_four(p));
        spdy::T63Batch b{};
        b.nseg = 2;
        b.seg[0] = spdy::T63Seg{ug, p->tmp_c, sc, nullptr, nb, 1, 0, 0};
        b.seg[1] = spdy::T63Seg{vg, p->tmp_d, sc, nullptr, nb, 1, 0, 0};
        RC(timed(p, SPDY_K_G2S_FUSED, [&] { return spdy::launch_g2s_fused_t63_batch(p->dev, b, p->num_cu, p->stream); }));
        KERNEL(spdy::launch_vds(p->dev, nb, p->tmp_c, p->tmp_d, vorm, divm, p->stream));
        return SPDY_OK;
    }
    RC(ensure_four(p));
    KERNEL(spdy::launch_fourier_dir(p->dev, nb, ug, sc, p->four, p->stream));
    KERNEL(spdy::launch_legendre_dir(p->dev, nb, p->four, p->tmp_c, p->stream));
    KERNEL(spdy::launch_fourier_dir(p->dev, nb, vg, sc, p->four, p->stream));
    KERNEL(spdy::launch_legendre_dir(p->dev, nb, p->four, p->tmp_d, p->stream));
    KERNEL(spdy::launch_vds(p->dev, nb, p->tmp_c, p->tmp_d, vorm, divm, p->stream));
    return SPDY_OK;
}

int spdy_inverse_batch_dev(spdy_plan *p, int npairs, const double *vor, const double *dv, double *ug, double *vg, int kcos_pairs,
                           int nplain, const double *spec, const int *d_kcos, int kcos_all, double *grid)
{
    NEED_DEVICE(p);
    RC(check_batch(p, npairs));
    RC(check_batch(p, nplain));
    if ((npairs && (!vor || !dv || !ug || !vg)) || (nplain && (!spec || !grid))) return fail(SPDY_ERR_ARG, "null device pointer");
    kcos_pairs = kcos_pairs == 1 ? 1 : 2;
    if (use_fused(p, npairs) && npairs > 0 && nplain > 0)
        return timed(p, SPDY_K_S2G_FUSED, [&] {
            return spdy::launch_s2g_fused(p->dev, npairs, vor, nullptr, kcos_pairs, ug, p->num_cu * p->wg_per_cu, p->stream, 3, dv, vg,
                                          nplain, spec, d_kcos, kcos_all, grid);
        });
    if (npairs > 0 && nplain > 0 && use_fused63_composite(p)) {
        // T63: U, V and the plain spectra as three segments of ONE fused launch; uvspec on load (model sizes) or as a kernel in front
        spdy::T63Batch b{};
        b.nseg = 3;
        if (derive63(p, npairs, 2, (nplain + 1) / 2)) append_uv_segs(b, 0, npairs, vor, dv, ug, vg, kcos_pairs);
        else {
            RC(ensure_four(p));
            KERNEL(spdy::launch_uvspec(p->dev, npairs, vor, dv, p->tmp_c, p->tmp_d, p->stream));
            b.seg[0] = spdy::T63Seg{p->tmp_c, ug, nullptr, nullptr, npairs, kcos_pairs, 0, 0};
            b.seg[1] = spdy::T63Seg{p->tmp_d, vg, nullptr, nullptr, npairs, kcos_pairs, 0, 0};
        }
        b.seg[2] = spdy::T63Seg{spec, grid, nullptr, d_kcos, nplain, kcos_all, 0, 0};
        return timed(p, SPDY_K_S2G_FUSED, [&] { return spdy::launch_s2g_fused_t63_batch(p->dev, b, p->num_cu, p->stream); });
    }
    if (npairs) RC(spdy_uvspec_to_grid_dev(p, npairs, vor, dv, ug, vg, kcos_pairs));
    if (nplain) RC(spdy_spec_to_grid_dev(p, nplain, spec, d_kcos, kcos_all, grid));
    return SPDY_OK;
}

int spdy_inverse_batch_grad_dev(spdy_plan *p, int npairs, const double *vor, const double *dv, double *ug, double *vg, int kcos_pairs,
                                int nplain, const double *spec, const int *d_kcos, int kcos_all, double *grid,
                                int ngrad, const double *psi, double *gx, double *gy, int kcos_grad)
{
    NEED_DEVICE(p);
    RC(check_batch(p, npairs));
    RC(check_batch(p, nplain));
    RC(check_batch(p, ngrad));
    if ((npairs && (!vor || !dv || !ug || !vg)) || (nplain && (!spec || !grid)) || (ngrad && (!psi || !gx || !gy)))
        return fail(SPDY_ERR_ARG, "null device pointer");
    kcos_pairs = kcos_pairs == 1 ? 1 : 2;
    kcos_grad = kcos_grad == 1 ? 1 : 2;
    if (use_fused(p, npairs) && npairs > 0 && nplain > 0 && ngrad > 0)   // T30: uvspec pairs, gradients and plain fields in ONE launch
        return timed(p, SPDY_K_S2G_FUSED, [&] {
            return spdy::launch_s2g_fused(p->dev, npairs, vor, nullptr, kcos_pairs, ug, p->num_cu * p->wg_per_cu, p->stream, 3, dv, vg,
                                          nplain, spec, d_kcos, kcos_all, grid, ngrad, psi, gx, gy, kcos_grad, p->d_zero_spec);
        });
    if (use_fused63_composite(p) && npairs > 0 && nplain > 0 && ngrad > 0 && npairs + ngrad <= p->max_batch) {
        // T63: U, V, d/dlambda, d/dmu and the plain spectra are five segments of ONE fused launch (the gradient alone would be a
        // one-workgroup launch of a full pipeline latency).  Model sizes: uvspec and grad are evaluated on load (row f1, round 6 --
        // no operator launch); larger batches: ONE operator launch (uvspec | grad) into the plan's temporaries in front.
        spdy::T63Batch b{};
        b.nseg = 5;
        if (derive63(p, npairs, 2, (nplain + 1) / 2 + 2 * ((ngrad + 1) / 2))) {
            append_uv_segs(b, 0, npairs, vor, dv, ug, vg, kcos_pairs);
            append_grad_segs(b, 3, ngrad, psi, gx, gy, kcos_grad);
        } else {
            RC(ensure_four(p));
            const size_t off = (size_t)npairs * spec_elems(p);
            KERNEL(spdy::launch_uvspec_grad(p->dev, npairs, vor, dv, p->tmp_c, p->tmp_d, ngrad, psi, p->tmp_c + off, p->tmp_d + off, p->stream));
            b.seg[0] = spdy::T63Seg{p->tmp_c, ug, nullptr, nullptr, npairs, kcos_pairs, 0, 0};
            b.seg[1] = spdy::T63Seg{p->tmp_d, vg, nullptr, nullptr, npairs, kcos_pairs, 0, 0};
            b.seg[3] = spdy::T63Seg{p->tmp_c + off, gx, nullptr, nullptr, ngrad, kcos_grad, 0, 0};
            b.seg[4] = spdy::T63Seg{p->tmp_d + off, gy, nullptr, nullptr, ngrad, kcos_grad, 0, 0};
        }
        b.seg[2] = spdy::T63Seg{spec, grid, nullptr, d_kcos, nplain, kcos_all, 0, 0};
        return timed(p, SPDY_K_S2G_FUSED, [&] { return spdy::launch_s2g_fused_t63_batch(p->dev, b, p->num_cu, p->stream); });
    }
    RC(spdy_inverse_batch_dev(p, npairs, vor, dv, ug, vg, kcos_pairs, nplain, spec, d_kcos, kcos_all, grid));
    if (ngrad) RC(spdy_grad_to_grid_dev(p, ngrad, psi, gx, gy, kcos_grad));
    return SPDY_OK;
}

int spdy_inverse_batch_segs_dev(spdy_plan *p, int npairs, const double *vor, const double *dv, double *ug, double *vg, int kcos_pairs,
                                int nseg, const spdy_spec_seg *segs, const int *d_kcos, int kcos_all, double *grid,
                                int ngrad, const double *psi, double *gx, double *gy, int kcos_grad)
{
    NEED_DEVICE(p);
    if (nseg < 0 || nseg > SPDY_MAX_SPEC_SEGS || (nseg && !segs)) return fail(SPDY_ERR_ARG, "0..SPDY_MAX_SPEC_SEGS segments");
    // drop empty segments, total the plain fields
    spdy_spec_seg sg[SPDY_MAX_SPEC_SEGS];
    int ns = 0, nplain = 0;
    for (int i = 0; i < nseg; ++i) {
        if (segs[i].nb < 0) return fail(SPDY_ERR_ARG, "negative segment size");
        if (segs[i].nb == 0) continue;
        if (!segs[i].d_spec) return fail(SPDY_ERR_ARG, "null device pointer");
        sg[ns++] = segs[i];
        nplain += segs[i].nb;
    }
    if (ns <= 1)
        return spdy_inverse_batch_grad_dev(p, npairs, vor, dv, ug, vg, kcos_pairs, nplain, ns ? sg[0].d_spec : nullptr, d_kcos, kcos_all, grid,
                                           ngrad, psi, gx, gy, kcos_grad);
    RC(check_batch(p, npairs));
    RC(check_batch(p, nplain));
    RC(check_batch(p, ngrad));
    if ((npairs && (!vor || !dv || !ug || !vg)) || !grid || (ngrad && (!psi || !gx || !gy))) return fail(SPDY_ERR_ARG, "null device pointer");
    kcos_pairs = kcos_pairs == 1 ? 1 : 2;
    kcos_grad = kcos_grad == 1 ? 1 : 2;
    if (use_fused(p, npairs) && npairs > 0) {                                 // T30: one mixed launch, source array looked up per field
        spdy::PlainSegs ps{{nullptr, nullptr, nullptr}, {0x7fffffff, 0x7fffffff, 0x7fffffff}};
        for (int i = 1, first = sg[0].nb; i < ns; first += sg[i].nb, ++i) { ps.spec[i - 1] = sg[i].d_spec; ps.first[i - 1] = first; }
        return timed(p, SPDY_K_S2G_FUSED, [&] {
            return spdy::launch_s2g_fused(p->dev, npairs, vor, nullptr, kcos_pairs, ug, p->num_cu * p->wg_per_cu, p->stream, 3, dv, vg,
                                          nplain, sg[0].d_spec, d_kcos, kcos_all, grid, ngrad, psi, gx, gy, kcos_grad, p->d_zero_spec, &ps);
        });
    }
    if (use_fused63_composite(p) && npairs > 0 && npairs + ngrad <= p->max_batch) {
        // T63: U, V, the gradient pair and every source array are segments of ONE fused launch; uvspec | grad on load at model sizes
        // (row f1, round 6), otherwise as one operator launch in front
        int plain_pairs = 0;
        for (int i = 0; i < ns; ++i) plain_pairs += (sg[i].nb + 1) / 2;
        const bool derive = derive63(p, npairs, 2, plain_pairs + 2 * ((ngrad + 1) / 2));
        const size_t off = (size_t)npairs * spec_elems(p);
        spdy::T63Batch b{};
        int k = 0;
        if (derive) k = append_uv_segs(b, k, npairs, vor, dv, ug, vg, kcos_pairs);
        else {
            RC(ensure_four(p));
            if (ngrad) KERNEL(spdy::launch_uvspec_grad(p->dev, npairs, vor, dv, p->tmp_c, p->tmp_d, ngrad, psi, p->tmp_c + off, p->tmp_d + off, p->stream));
            else KERNEL(spdy::launch_uvspec(p->dev, npairs, vor, dv, p->tmp_c, p->tmp_d, p->stream));
            b.seg[k++] = spdy::T63Seg{p->tmp_c, ug, nullptr, nullptr, npairs, kcos_pairs, 0, 0};
            b.seg[k++] = spdy::T63Seg{p->tmp_d, vg, nullptr, nullptr, npairs, kcos_pairs, 0, 0};
        }
        size_t first = 0;
        for (int i = 0; i < ns; ++i) {
            b.seg[k++] = spdy::T63Seg{sg[i].d_spec, grid + first * grid_elems(p), nullptr, d_kcos ? d_kcos + first : nullptr, sg[i].nb, kcos_all, 0, 0};
            first += sg[i].nb;
        }
        if (ngrad && derive) k = append_grad_segs(b, k, ngrad, psi, gx, gy, kcos_grad);
        else if (ngrad) {
            b.seg[k++] = spdy::T63Seg{p->tmp_c + off, gx, nullptr, nullptr, ngrad, kcos_grad, 0, 0};
            b.seg[k++] = spdy::T63Seg{p->tmp_d + off, gy, nullptr, nullptr, ngrad, kcos_grad, 0, 0};
        }
        b.nseg = k;
        return timed(p, SPDY_K_S2G_FUSED, [&] { return spdy::launch_s2g_fused_t63_batch(p->dev, b, p->num_cu, p->stream); });
    }
    // any other plan: the separate calls
    if (npairs) RC(spdy_uvspec_to_grid_dev(p, npairs, vor, dv, ug, vg, kcos_pairs));
    size_t first = 0;
    for (int i = 0; i < ns; ++i) {
        RC(spdy_spec_to_grid_dev(p, sg[i].nb, sg[i].d_spec, d_kcos ? d_kcos + first : nullptr, kcos_all, grid + first * grid_elems(p)));
        first += sg[i].nb;
    }
    if (ngrad) RC(spdy_grad_to_grid_dev(p, ngrad, psi, gx, gy, kcos_grad));
    return SPDY_OK;
}

int spdy_direct_batch_dev(spdy_plan *p, int npairs, const double *ug, const double *vg, double *vorm, double *divm, int kcos,
                          int nplain, const double *grid, double *spec)
{
    NEED_DEVICE(p);
    RC(check_batch(p, npairs));
    RC(check_batch(p, nplain));
    if ((npairs && (!ug || !vg || !vorm || !divm)) || (nplain && (!grid || !spec))) return fail(SPDY_ERR_ARG, "null device pointer");
    if (use_fused(p, npairs) && npairs > 0 && nplain > 0 && one_mixed_launch(p, 2L * npairs + nplain)) {
        const double *sc = kcos == 2 ? p->dev.cosgr : p->dev.cosgr2;
        return timed(p, SPDY_K_G2S_FUSED, [&] {
            return spdy::launch_g2s_fused(p->dev, npairs, ug, sc, vorm, p->num_cu * p->wg_per_cu, p->stream, vg, divm, nplain, grid, spec);
        });
    }
    if (npairs > 0 && nplain > 0 && use_fused63_composite(p) && spdy::g2s_t63_staged(p->dev, p->num_cu, npairs + (nplain + 1) / 2)) {
        // T63, model-sized: the (u, v) pairs (vds in the contraction) and the plain grids as two segments of the staged form
        const double *sc = kcos == 2 ? p->dev.cosgr : p->dev.cosgr2;
        spdy::T63Batch b{};
        b.nseg = 2;
        b.seg[0] = spdy::T63Seg{ug, vorm, sc, nullptr, npairs, 1, 0, spdy::T63_OP_VDS};
        b.vds_src2 = vg; b.vds_dst2 = divm;
        b.seg[1] = spdy::T63Seg{grid, spec, nullptr, nullptr, nplain, 1, 0, 0};
        return timed(p, SPDY_K_G2S_FUSED, [&] { return spdy::launch_g2s_fused_t63_batch(p->dev, b, p->num_cu, p->stream); });
    }
    if (npairs > 0 && nplain > 0 && use_fused63_composite(p)) {
        // T63: the scaled u, v grids and the plain grids as three segments of ONE fused launch, then vds
        RC(ensure_four(p));
        const double *sc = kcos == 2 ? p->dev.cosgr : p->dev.cosgr2;
        spdy::T63Batch b{};
        b.nseg = 3;
        b.seg[0] = spdy::T63Seg{ug, p->tmp_c, sc, nullptr, npairs, 1, 0, 0};
        b.seg[1] = spdy::T63Seg{vg, p->tmp_d, sc, nullptr, npairs, 1, 0, 0};
        b.seg[2] = spdy::T63Seg{grid, spec, nullptr, nullptr, nplain, 1, 0, 0};
        RC(timed(p, SPDY_K_G2S_FUSED, [&] { return spdy::launch_g2s_fused_t63_batch(p->dev, b, p->num_cu, p->stream); }));
        KERNEL(spdy::launch_vds(p->dev, npairs, p->tmp_c, p->tmp_d, vorm, divm, p->stream));
        return SPDY_OK;
    }
    if (npairs) RC(spdy_vdspec_dev(p, npairs, ug, vg, vorm, divm, kcos));
    if (nplain) RC(spdy_grid_to_spec_dev(p, nplain, grid, spec));
    return SPDY_OK;
}

#define HOST_1IN_1OUT(name, devfn)                                                 \
    int name(spdy_plan *p, int nb, const double *in, double *out)                  \
    {                                                                              \
        NEED_DEVICE(p);                                                            \
        RC(check_batch(p, nb));                                                    \
        if (nb && (!in || !out)) return fail(SPDY_ERR_ARG, "null pointer");        \
        RC(ensure_staging(p, nb * spec_elems(p)));                                 \
        RC(h2d(p, p->stage_a, in, nb * spec_elems(p)));                            \
        RC(devfn(p, nb, p->stage_a, p->stage_b));                                  \
        RC(d2h(p, out, p->stage_b, nb * spec_elems(p)));                           \
        return sync(p);                                                            \
    }
HOST_1IN_1OUT(spdy_laplacian, spdy_laplacian_dev)
HOST_1IN_1OUT(spdy_inverse_laplacian, spdy_inverse_laplacian_dev)

int spdy_trunct(spdy_plan *p, int nb, double *inout)
{
    NEED_DEVICE(p);
    RC(check_batch(p, nb));
    if (nb && !inout) return fail(SPDY_ERR_ARG, "null pointer");
    RC(ensure_staging(p, nb * spec_elems(p)));
    RC(h2d(p, p->stage_a, inout, nb * spec_elems(p)));
    RC(spdy_trunct_dev(p, nb, p->stage_a));
    RC(d2h(p, inout, p->stage_a, nb * spec_elems(p)));
    return sync(p);
}

int spdy_grad(spdy_plan *p, int nb, const double *psi, double *psdx, double *psdy)
{
    NEED_DEVICE(p);
    RC(check_batch(p, nb));
    if (nb && (!psi || !psdx || !psdy)) return fail(SPDY_ERR_ARG, "null pointer");
    RC(ensure_staging(p, nb * spec_elems(p)));
    const size_t n = nb * spec_elems(p);
    RC(h2d(p, p->stage_a, psi, n));
    // rows of psdy the reference leaves untouched keep the caller's values
    RC(h2d(p, p->stage_c, psdy, n));
    RC(spdy_grad_dev(p, nb, p->stage_a, p->stage_b, p->stage_c));
    RC(d2h(p, psdx, p->stage_b, n));
    RC(d2h(p, psdy, p->stage_c, n));
    return sync(p);
}

#define HOST_2IN_2OUT(name, devfn)                                                                   \
    int name(spdy_plan *p, int nb, const double *a, const double *b, double *c, double *d)           \
    {                                                                                                \
        NEED_DEVICE(p);                                                                              \
        RC(check_batch(p, nb));                                                                      \
        if (nb && (!a || !b || !c || !d)) return fail(SPDY_ERR_ARG, "null pointer");                 \
        RC(ensure_staging(p, nb * spec_elems(p)));                                                   \
        const size_t n = nb * spec_elems(p);                                                         \
        RC(h2d(p, p->stage_a, a, n));                                                                \
        RC(h2d(p, p->stage_b, b, n));                                                                \
        RC(h2d(p, p->stage_c, c, n)); /* untouched entries keep the caller's values */              \
        RC(h2d(p, p->stage_d, d, n));                                                                \
        RC(devfn(p, nb, p->stage_a, p->stage_b, p->stage_c, p->stage_d));                            \
        RC(d2h(p, c, p->stage_c, n));                                                                \
        RC(d2h(p, d, p->stage_d, n));                                                                \
        return sync(p);                                                                              \
    }
HOST_2IN_2OUT(spdy_vds, spdy_vds_dev)
HOST_2IN_2OUT(spdy_uvspec, spdy_uvspec_dev)

int spdy_vdspec(spdy_plan *p, int nb, const double *ug, const double *vg, double *vorm, double *divm, int kcos)
{
    NEED_DEVICE(p);
    RC(check_batch(p, nb));
    if (nb && (!ug || !vg || !vorm || !divm)) return fail(SPDY_ERR_ARG, "null pointer");
    RC(ensure_staging(p, nb * grid_elems(p)));
    RC(h2d(p, p->stage_a, ug, nb * grid_elems(p)));
    RC(h2d(p, p->stage_b, vg, nb * grid_elems(p)));
    // (the one-pass kernel writes vorticity/divergence while other workgroups still read grids: outputs never alias inputs)
    RC(spdy_vdspec_dev(p, nb, p->stage_a, p->stage_b, p->stage_c, p->stage_d, kcos));
    RC(d2h(p, vorm, p->stage_c, nb * spec_elems(p)));
    RC(d2h(p, divm, p->stage_d, nb * spec_elems(p)));
    return sync(p);
}

/* ---------------------------------------------------------------- HIP graphs */
int spdy_graph_begin(spdy_plan *p)
{
    NEED_DEVICE(p);
    if (p->capturing) return fail(SPDY_ERR_STATE, "a graph capture is already open on this plan");
    if (!p->stream) return fail(SPDY_ERR_STATE, "graph capture needs a non-default stream (spdy_plan_set_stream)");
    HIP_TRY(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
    p->capturing = true;
    return SPDY_OK;
}

int spdy_graph_end(spdy_plan *p, spdy_graph **graph)
{
    NEED_DEVICE(p);
    if (!graph) return fail(SPDY_ERR_ARG, "graph == NULL");
    if (!p->capturing) return fail(SPDY_ERR_STATE, "no graph capture is open on this plan");
    p->capturing = false;
    spdy_graph *g = new spdy_graph;
    g->plan = p;
    hipError_t e = hipStreamEndCapture(p->stream, &g->graph);
    if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        if (g->graph) (void)hipGraphDestroy(g->graph);
        delete g;
        return fail(SPDY_ERR_HIP, "graph capture failed: %s", hipGetErrorString(e));
    }
    p->graphs.push_back(g);
    *graph = g;
    return SPDY_OK;
}

int spdy_graph_launch(spdy_graph *g)
{
    if (!g) return fail(SPDY_ERR_ARG, "graph == NULL");
    if (!g->plan || !g->exec) return fail(SPDY_ERR_STATE, "the plan this graph was captured from has been destroyed");
    HIP_TRY(hipSetDevice(g->plan->device));
    if (g->plan->capturing) return fail(SPDY_ERR_STATE, "cannot launch a graph while a capture is open");
    HIP_TRY(hipGraphLaunch(g->exec, g->plan->stream));
    return SPDY_OK;
}

int spdy_graph_num_nodes(spdy_graph *g, int *nodes)
{
    if (!g || !nodes) return fail(SPDY_ERR_ARG, "null graph / output");
    if (!g->graph) return fail(SPDY_ERR_STATE, "the graph has been destroyed");
    size_t n = 0;
    HIP_TRY(hipGraphGetNodes(g->graph, nullptr, &n));
    *nodes = (int)n;
    return SPDY_OK;
}

int spdy_graph_destroy(spdy_graph *g)
{
    if (!g) return SPDY_OK;
    if (g->plan) {
        auto &v = g->plan->graphs;
        for (size_t i = 0; i < v.size(); ++i) if (v[i] == g) { v.erase(v.begin() + i); break; }
    }
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return SPDY_OK;
}

}  // extern "C"
