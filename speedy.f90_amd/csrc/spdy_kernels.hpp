// Launch interface between the C-ABI layer (spdy_api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace spdy {

// Launch-policy switches of a plan (host-side use: which kernel form a launcher picks; none changes results beyond the
// rounding-level path choices documented in include/spdy.h).  Read from the environment ONCE, at plan creation
// (spdy_plan_create), and changed per plan with spdy_plan_set_option -- no launcher calls getenv.
struct LaunchOpts {
    int t30_nopart = 0;     // small T30 inverse launches walk whole tiles              ($SPDY_T30_NOPART,  "t30_part" = 0)
    int t30_nosplit = 0;    // small T30 direct launches as whole tiles                 ($SPDY_T30_NOSPLIT, "t30_split" = 0)
    int t63_nosplit = 0;    // one workgroup per pair in small fused T63 direct launches ($SPDY_T63_NOSPLIT, "t63_split" = 0)
    int t63_nostage = 0;    // small T63 direct batches fused instead of staged         ($SPDY_T63_NOSTAGE, "t63_stage" = 0)
    int t63_np2_from = 40;  // pairs from which the staged contraction takes two pairs per workgroup ($SPDY_T63_NP2_FROM)
    int wt_min_mb = 6;      // output MB from which a model-sized launch writes through ($SPDY_WT_MIN_MB; 0 = never)
};

// Device-side view of a plan: dimensions + table pointers (all device memory).
struct DevPlan {
    int trunc, ix, iy, il, kx, nx, mx;
    int num_cu;    // compute units of the plan's device (sizes the persistent launches; host-side use)
    int fs;        // row stride (doubles) of the Fourier workspace: 2*mx rounded up to 16
    int ks_inv;    // k-steps (of 4 n's) per parity in the inverse-Legendre A table
    int jt;        // 16-latitude tiles per hemisphere (ceil(iy/16))
    int nt_dir;    // 16-row n tiles per parity in the direct-Legendre A table
    int js_dir;    // k-steps (of 4 latitudes) in the direct transform (iy/4)
    const double *pa_inv;   // [mx][2][ks_inv][jt][64]      P(m,n,j)            MFMA A fragments
    const double *pa_dir;   // [mx][2][nt_dir][js_dir][64]  P(m,n,j)*wt(j)      MFMA A fragments
    // 4x4x4-block fragments with both parities packed (fused T30 kernels; nullptr otherwise):
    const double *pa_inv2;  // [mx][ks_inv][64]  blocks 0,1: latitudes 16..23 of even n; blocks 2,3: of odd n
    const double *pa_dir2;  // [mx][js_dir][64]  blocks 0,1: n rows 0..7 of even n; blocks 2,3: of odd n
    // the same fragments in the order the fused T30 kernels keep them in registers, two per 16-byte load:
    const double *img_s2g;  // [4 Legendre waves][30][64 lanes][2]
    const double *img_g2s;  // [4 Legendre waves][36][64 lanes][2]
    // the inverse kernel's small-batch form (work items = (tile, third of the latitudes)): per part the packed fragments of
    // latitudes 8 part .. 8 part + 7 (blocks 0, 1: even n, blocks 2, 3: odd n), in slot order
    const double *img_s2g3; // [3 parts][4 Legendre waves][10][64 lanes][2]
    // A-operand images of the fused T63 kernels: [4 Legendre waves][38 slots][6 chunks][64 lanes][2] (spdy_t63_sched.hpp)
    const double *img_g2s63, *img_s2g63;
    // row workspace of the staged form of small T63 direct batches: [pair][chunk][field][16 rows][128] Fourier rows
    double *rows_ws;
    int rows_ws_fields;     // capacity in fields (0: none -- the fused split form runs instead)
    const double *cosgr;    // [il]
    const double *cosgr2;   // [il]
    // spectral operator tables, each [nx][mx] (gradx: [mx])
    const double *el2, *elm2, *trfilt, *gradx, *gradym, *gradyp, *uvdx, *uvdym, *uvdyp, *vddym, *vddyp;
    const double *gradx_e;  // gradx expanded to [nx][mx] (gradient tiles of the mixed inverse kernel index it like uvdx)
    // implicit tables
    const double *xd, *xc, *xj, *tref1, *dhsx, *elz;
    // the same matrices row-major with rows padded to kxp = kx rounded up to even (16-byte rows): xdt, xct [kx][kxp],
    // xjt [mx+nx+1][kx][kxp] -- a thread's mat-vec row is one contiguous run instead of kx strided elements
    const double *xdt, *xct, *xjt;
    int kxp;
    // per-level tables [kx]: sigma-level functions (geometry.f90:51-60, geopotential.f90:22-30,53,
    // horizontal_diffusion.f90:70-82) and the reference temperature profile (implicit.f90:62-67);
    // rgtref = rgas*tref.  tref* are filled by spdy_implicit_init, the others whenever sigma levels exist.
    const double *dhs, *dhsr, *fsgr, *tref, *tref2, *tref3, *rgtref, *xgeop1, *xgeop2, *corf, *tcorv, *qcorv;
    const double *dmp_t[6]; // dmp, dmpd, dmps, dmp1, dmp1d, dmp1s [nx][mx] (dmp1* valid after spdy_implicit_init)
    const double *coriol;   // [il] 2*omega*sin(lat) (geometry.f90:89), southernmost row first
    double rgas, akap;      // physical_constants.f90:22-24 (float32 literals widened)
    LaunchOpts lo;          // (host-side use, like num_cu)
};
constexpr int LEVTAB_COUNT = 12;   // number of per-level tables above (one device allocation of LEVTAB_COUNT*kx doubles)

// FFTPACK twiddles / constants for one resolution; copied to __constant__ memory once.
struct FftConstants {
    double first[144];  // stage with ido=48: 1 block (N=96, radix 2) or 3 blocks (N=192, radix 4)
    double a[36];       // radix-4 stage, ido=12: 3 blocks of 12
    double b[9];        // radix-4 stage, ido=3 : 3 blocks of 3
    double taui, sqrt2, hsqt2, scale;
};
hipError_t upload_fft_constants(int ix, const FftConstants &c);

// four: [nb][il][fs] workspace layout.  All launches are asynchronous on `s`.
hipError_t launch_legendre_inv(const DevPlan &p, int nb, const double *spec, double *four, hipStream_t s);
hipError_t launch_legendre_dir(const DevPlan &p, int nb, const double *four, double *spec, hipStream_t s);
hipError_t launch_fourier_inv(const DevPlan &p, int nb, const double *four, const int *d_kcos, int kcos_all,
                              double *grid, hipStream_t s);
// gscale: nullptr, or a per-latitude factor applied to the grid on load (vdspec's cosgr/cosgr2)
hipError_t launch_fourier_dir(const DevPlan &p, int nb, const double *grid, const double *gscale, double *four,
                              hipStream_t s);

// Fused persistent T30 kernels (whole transform in one pass through LDS; at most max_wg workgroups)
// mode 0: plain.  mode 1: uvspec fused -- tile i = (vor[i], div[i]) = (spec, spec2) -> (ug, vg) = (grid, grid2).
// mode 2: grad fused -- tile i = psi[i] = spec -> (d/dx, d/dy) = (grid, grid2).  In modes 1/2 nb counts tiles and
// kcos_all applies to both outputs.
// mode 3: a model step's whole inverse batch in one launch -- nb (vor, div) pairs as in mode 1 plus nplain ordinary
// fields spec_p -> grid_p with their own kcos (kcos_p per field, or kcos_all_p)
// mode 3: further source arrays of the plain spectra (S2gMixed::seg_spec / seg_first; first = 0x7fffffff: unused)
struct PlainSegs { const double *spec[3]; int first[3]; };
hipError_t launch_s2g_fused(const DevPlan &p, int nb, const double *spec, const int *d_kcos, int kcos_all, double *grid,
                            int max_wg, hipStream_t s, int mode = 0, const double *spec2 = nullptr, double *grid2 = nullptr,
                            int nplain = 0, const double *spec_p = nullptr, const int *kcos_p = nullptr, int kcos_all_p = 1,
                            double *grid_p = nullptr, int ngrad = 0, const double *psi = nullptr, double *gx = nullptr,
                            double *gy = nullptr, int kcos_grad = 2, const double *zero = nullptr, const PlainSegs *segs = nullptr);
// (mode 3, ngrad > 0: gradient tiles psi[i] -> gx[i], gy[i] ride along as uvspec tiles with vor = `zero` and the grad tables)
// grid2 / spec2 non-null: vdspec in one pass -- tile i is the pair (grid[i], grid2[i]) scaled by gscale, the
// outputs are vds of the pair's spectra: vorticity -> spec, divergence -> spec2 (nb pairs)
// nplain > 0: a model step's whole direct batch in one launch -- nb (u,v) pairs as above plus nplain ordinary fields
// grid_p -> spec_p (unscaled)
hipError_t launch_g2s_fused(const DevPlan &p, int nb, const double *grid, const double *gscale, double *spec, int max_wg,
                            hipStream_t s, const double *grid2 = nullptr, double *spec2 = nullptr, int nplain = 0,
                            const double *grid_p = nullptr, double *spec_p = nullptr, bool allow_split = true);
// (allow_split = false: never the three-workgroups-per-tile form -- its workgroups read every tile three times, which is the
// wrong trade when the rows are host-mapped staging memory read across the link)

// SIMD of each of the eight waves of nwg workgroups shaped like the fused T63 kernels' (d_out: 8 ints per workgroup)
hipError_t launch_wave_placement(int *d_out, int nwg, hipStream_t s);
// Fused T63 kernels: a pair of fields per tile, six latitude chunks, accumulators / B operands resident in VGPRs
hipError_t launch_s2g_fused_t63(const DevPlan &p, int nb, const double *spec, const int *d_kcos, int kcos_all, double *grid, int max_wg,
                                hipStream_t s);
// One fused T63 launch over up to six independent sub-batches (own arrays, own scale / kcos policy); pair0 and npairs are
// filled in by the launcher
constexpr int T63_MAX_SEG = 8;
struct T63Seg {
    const double *src;
    double *dst;
    const double *scale;   // direct: per-latitude factor applied on load, or nullptr
    const int *kcos;       // inverse: per-field kcos (device), or nullptr for kcos_all
    int nb, kcos_all, pair0;
    int op;                // inverse: 0 = src holds the spectra; T63_OP_* = they are derived from src / (const double *)scale on load
                           // direct: T63_OP_VDS = vdspec pairs -- pair i of the segment is (src[i], T63Batch::vds_src2[i]) = (u, v) grids,
                           // nb pairs; the contraction applies vds (spectral.f90:146-171) to the pair's spectra in registers:
                           // vorticity -> dst[i], divergence -> T63Batch::vds_dst2[i]
};
// spectral operators folded into the inverse kernel's operand load (spectral.f90:124-196): the segment's spectra are
//   U / V of uvspec(vor = src, div = scale)  or  d/dlambda / d/dmu of grad(psi = src)
enum { T63_OP_NONE = 0, T63_OP_U = 1, T63_OP_V = 2, T63_OP_GX = 3, T63_OP_GY = 4, T63_OP_VDS = 5 };
struct T63Batch {
    int nseg, npairs;
    int by_chunk, wt;      // inverse, small batches: work items are (pair, chunk) instead of whole pairs; model-sized launches with
                           // several MB of output: stores written through (both set by the launcher)
    int nop_items, ipw;    // by-chunk walk (set by the launcher): items of the derived segments (one per workgroup), items per
                           // workgroup behind them
    const double *vds_src2; // direct launches: the second arrays of the (one) T63_OP_VDS segment -- in the header, not in the
    double *vds_dst2;       // segment, whose size decides whether the compiler indexes the by-value argument or copies it to scratch
    T63Seg seg[T63_MAX_SEG];
};
hipError_t launch_s2g_fused_t63_batch(const DevPlan &p, T63Batch b, int max_wg, hipStream_t s);
// whether an inverse launch of `pairs` field pairs may carry `op_pairs` derived ones (model-sized launches: the by-chunk form)
bool s2g_t63_derives(int max_wg, int pairs, int op_pairs);
hipError_t launch_g2s_fused_t63_batch(const DevPlan &p, T63Batch b, int max_wg, hipStream_t s);
// whether a direct launch of `pairs` field pairs takes the STAGED form (rows launch + contraction launch: model-sized batches) --
// the form that can apply vds to T63_OP_VDS segments (callers otherwise run vds as a kernel behind the launch)
bool g2s_t63_staged(const DevPlan &p, int max_wg, int pairs);
hipError_t launch_g2s_fused_t63(const DevPlan &p, int nb, const double *grid, const double *gscale, double *spec, int max_wg, hipStream_t s);

enum SpecOp { OP_LAPLACIAN = 0, OP_INV_LAPLACIAN = 1, OP_TRUNCT = 2 };
hipError_t launch_scale_op(const DevPlan &p, int op, int nb, const double *in, double *out, hipStream_t s);
hipError_t launch_grad(const DevPlan &p, int nb, const double *psi, double *psdx, double *psdy, hipStream_t s);
hipError_t launch_vds(const DevPlan &p, int nb, const double *u, const double *v, double *vor, double *div, hipStream_t s);
hipError_t launch_uvspec(const DevPlan &p, int nb, const double *vor, const double *div, double *u, double *v, hipStream_t s);
hipError_t launch_uvspec_grad(const DevPlan &p, int nuv, const double *vor, const double *div, double *u, double *v, int ngr,
                              const double *psi, double *psdx, double *psdy, hipStream_t s);
hipError_t launch_hdiff(const DevPlan &p, int nlev, const double *field, const double *fdt, const double *dmp,
                        const double *dmp1, double *out, hipStream_t s);
struct HdiffOps {   // up to 8 independent diffusion operations, passed by value as one kernel argument
    int nops, nlev[8];
    const double *field[8], *fdt[8], *dmp[8], *dmp1[8];
    double *out[8];
};
hipError_t launch_hdiff_multi(const DevPlan &p, const HdiffOps &ops, hipStream_t s);
hipError_t launch_implicit(const DevPlan &p, double *divdt, double *tdt, double *psdt, hipStream_t s);

// ---- spectral side of a time step (spdy_step.hip) ----
// step_field_2d/3d (time_stepping.f90:121-167): up to 8 prognostic arrays [2][nlev][nx][mx] + their tendencies in one launch
struct StepOps {
    int nops, nlev[8];
    double *field[8], *fdt[8];
};
hipError_t launch_step_fields(const DevPlan &p, const StepOps &ops, int j1, double dt, double eps, double wil, int do_trunct,
                              hipStream_t s);
// the diffusion block of `step` (time_stepping.f90:62-96) incl. the orographic corrections and the stratospheric drag
struct HdiffStep {
    const double *vor, *div, *t, *tr;            // time level 1 of the prognostics, [kx][nx][mx] complex
    double *vordt, *divdt, *tdt, *trdt;          // tendencies, in place
    const double *tcorh, *qcorh;                 // [nx][mx] complex (horizontal_diffusion.f90:31-32)
    const double *dmp, *dmpd, *dmps, *dmp1, *dmp1d, *dmp1s;
    double sdrag;
};
hipError_t launch_hdiff_step(const DevPlan &p, const HdiffStep &h, hipStream_t s);
// get_geopotential (geopotential.f90:33-57)
hipError_t launch_geopotential(const DevPlan &p, const double *t, const double *phis, double *phi, hipStream_t s);
// get_spectral_tendencies (tendencies.f90:242-293); phi is written as the reference's module variable is
hipError_t launch_spectral_tendencies(const DevPlan &p, const double *div, const double *t, const double *ps, const double *phis,
                                      double *divdt, double *tdt, double *psdt, double *phi, hipStream_t s);
// Level-block layout of a level-sharded step (spdy_api_shard.hip).  Rank r of R owns the levels [kx r / R, kx (r + 1) / R)
// = [lo_r, hi_r), nl_r of them.  A stack of F fields x kx levels that the ranks fill and exchange is stored block by block,
// block r = the rank's own contiguous launch operands [F][nl_r] (+ X level-free fields behind them): its slab offset is
// F lo_r + X r, and field f of level k lives at slab  F lo_r + X r + f nl_r + (k - lo_r)  of the owner r of k.  So ONE
// contiguous block per rank travels in the exchange and every transform launch reads/writes plain contiguous stacks; only
// the column kernels (grid tendencies, spectral step) index through the blocks.  nranks = 0: not sharded (the plain [F][kx]
// stacks of separate pointers); with one rank the block layout coincides with the plain one.
struct LevelShard { int nranks, rank; };
// grid-space dynamical tendencies (tendencies.f90:105-197) and their spectral-space combination (:125-126, 218-233)
struct GridTend {
    const double *ug, *vg, *tg, *vorg, *divg, *trg;   // [kx] grids each (vorg WITHOUT the Coriolis term)
    const double *px, *py;                            // grad(ps) on the grid
    double *u, *v, *plain;                            // [3 kx], [3 kx], [3 kx + 1] grids: operands of the direct batch
    // sh.nranks >= 1: the six inputs are ONE level-block stack `ug` (F = 6, X = 0, field order ug, vg, vorg, divg, tg, trg:
    // the other five pointers are ignored) holding all levels; the outputs are this rank's own launch operands only --
    // u, v [3 nl], plain [3 nl + 1] with nl = the rank's level count (every rank gets the level-free last field)
    LevelShard sh;
    // TRANSPOSED form of the level-sharded step (sh.nranks >= 1 and tr_out non-null): this rank holds ALL levels of the points
    // [pt0, pt0 + npts) only.  `ug` is the F = 6 level-block stack with slabs of npts doubles (local point index); px, py stay
    // whole grids (every rank computes them itself) and are read at pt0 + i; the outputs of ALL levels go to tr_out, an F = 9,
    // X = 1 level-block stack with slabs of npts doubles -- block q = rank q's direct-batch operands u [3 nl_q] | v [3 nl_q] |
    // plain [3 nl_q] | the level-free field, restricted to these points (u, v, plain are ignored).
    int npts, pt0;
    double *tr_out;
};
hipError_t launch_grid_tendencies(const DevPlan &p, const GridTend &g, hipStream_t s);
// Whether a fused launch with this many grid-side bytes streams them (non-temporal loads / stores; >= 16 MB): the size from which
// a launch is throughput-bound rather than latency-bound
bool streams(long grid_bytes);
// Write-through policy of a model-sized launch (the step's kernels): outputs of at least p.lo.wt_min_mb MB (default 6; 0 = never)
// leave the L2s as they are produced instead of waiting, dirty, for the end-of-kernel release.
bool write_through_policy(const DevPlan &p, long output_bytes);
hipError_t launch_tendency_combine(const DevPlan &p, double *pdiv, double *pspec, hipStream_t s);
// the whole spectral-space tail of a step in one launch (kx <= 16)
struct SpecStep {
    double *pvor, *pdiv, *pspec;                 // direct-batch outputs [3kx], [3kx], [3kx+1]; tendencies are left in them
    double *vor, *div, *t, *tr, *ps;             // prognostics, both time levels ([2][kx] / [2])
    const double *phis, *tcorh, *qcorh;
    double *phi;
    double sdrag, dt, eps, wil;
    int j1, do_trunct;
    // non-null: the vdspec pairs' spectra have NOT been through vds yet -- raw_u, raw_v [3kx] are grid_to_spec of the scaled
    // (u, v) grids and the kernel applies vds (spectral.f90:146-171) where it reads them (pvor / pdiv are outputs only)
    const double *raw_u, *raw_v;
    // sh.nranks >= 1: the direct batches' outputs of all ranks are ONE level-block stack `pvor` (F = 9: the rank's pvor | pdiv
    // | pspec stacks of 3 nl each, X = 1: its copy of the level-free psdt; pdiv / pspec / raw_u / raw_v are then only flags:
    // raw_u non-null = the first six groups are the raw pairs' spectra) and the final tendencies go to tend_out
    // [vordt | divdt | tdt | trdt] (kx each) | psdt in the plain layout instead of back into the operands
    LevelShard sh;
    double *tend_out;
    // TRANSPOSED form (sh.nranks >= 1 and ne > 0): this rank holds ALL levels of the coefficients [e0, e0 + ne) only (e0 a
    // multiple of the kernel's 16-coefficient blocks).  The level-block stack `pvor` has slabs of ne complex values (local
    // coefficient index); the prognostics, phi and tend_out are the whole arrays as ever and are read / written at these
    // coefficients only.  No raw pairs (vds needs the neighbouring rows): raw_u must be null.
    int e0, ne;
};
hipError_t launch_spectral_step(const DevPlan &p, const SpecStep &a, hipStream_t s);
// output path (input_output.f90:184-206)
struct GatherOps { int nops, nfld[8]; const double *src[8]; double *dst[8]; };
hipError_t launch_gather_spectra(const DevPlan &p, const GatherOps &g, hipStream_t s);
struct OutputCast { int nops, nfld[8], kind[8]; double factor[8]; const double *src[8]; float *dst[8]; };
hipError_t launch_output_cast(const DevPlan &p, const OutputCast &c, hipStream_t s);
// once per device, before the first launch: raises the dynamic-LDS limit of every kernel that needs > 64 KB
hipError_t prepare_device_kernels();
hipError_t prepare_device_step_kernels(int kx);

}  // namespace spdy
