/*
 * spdy.h -- C ABI of the MI355X-native spectral transform path for speedy.f90.
 *
 * This is the drop-in boundary: every entry point below replaces one public procedure of the
 * reference's Fortran modules `spectral`, `fourier`, `legendre`, `horizontal_diffusion` and
 * `implicit` (file:line cited per function, relative to /root/reference/source).  The
 * reference-side binding (an ISO_C_BINDING module that keeps the reference's names and
 * signatures) is speedy.f90_amd/fortran/spectral.f90; see INTEGRATION.md.
 *
 * Conventions
 *  - All data is FP64.  Arrays are Fortran column-major exactly as the reference declares them:
 *        grid  g(ix,il)            ix*il doubles, longitude fastest, j=1 southernmost
 *        spec  s(mx,nx) complex    mx*nx (re,im) pairs, zonal wavenumber index fastest
 *        four  f(2*mx,il)          re/im interleaved Fourier coefficients per latitude
 *    A batch of nb fields is nb such arrays back to back (e.g. a (mx,nx,kx) level stack).
 *  - Functions without suffix take HOST pointers: they copy in, run the HIP kernels, copy out
 *    and synchronise (signature-compatible with the reference, PCIe-bound).
 *    Functions ending in _dev take DEVICE pointers, are asynchronous on the plan's stream and
 *    never touch the host -- this is the throughput path (state stays resident in HBM).
 *  - Every function returns 0 on success or a negative SPDY_ERR_* code; spdy_last_error()
 *    gives the message for the calling thread.  There is no CPU fallback: without a usable
 *    HIP device every compute entry point fails with SPDY_ERR_NO_DEVICE.
 *  - A plan is immutable after creation (except spdy_implicit_init / spdy_plan_set_stream);
 *    calls on one plan are serialised by its stream; distinct plans are independent.
 */
#ifndef SPDY_H
#define SPDY_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spdy_plan spdy_plan;

enum {
    SPDY_OK = 0,
    SPDY_ERR_ARG = -1,          /* bad argument (null pointer, nb > max_batch, bad kcos ...)   */
    SPDY_ERR_UNSUPPORTED = -2,  /* resolution the kernels are not built for                   */
    SPDY_ERR_NO_DEVICE = -3,    /* no HIP device / host-only plan used for compute             */
    SPDY_ERR_HIP = -4,          /* a HIP runtime call failed                                  */
    SPDY_ERR_STATE = -5,        /* e.g. implicit_terms before implicit_init                    */
    SPDY_ERR_TABLE = -6,        /* generated table failed its pinned-value self check          */
    SPDY_ERR_COMM = -7          /* RCCL could not be loaded / a collective failed              */
};

/* ---- plan ------------------------------------------------------------------------------
 * Replaces the private module state filled by initialize_geometry (geometry.f90:35),
 * initialize_fourier (fourier.f90:18), initialize_legendre (legendre.f90:23),
 * initialize_spectral (spectral.f90:20) and initialize_horizontal_diffusion
 * (horizontal_diffusion.f90:36).  Supported: (trunc,ix,iy) = (30,96,24) and (63,192,48); 1 <= kx <= SPDY_MAX_KX.
 * Transforms, operators and do_horizontal_diffusion are level-agnostic.  Everything that needs sigma levels
 * (implicit solve, geopotential, spectral tendencies, the diffusion block's orographic correction) works out of the
 * box for kx in {5,7,8} -- the only sets the reference defines (geometry.f90:42-48) -- and for any other kx after
 * spdy_plan_set_sigma.
 * device >= 0 selects a HIP device; device = SPDY_DEVICE_AUTO takes $SPDY_DEVICE, else the process's local rank
 * ($LOCAL_RANK, $OMPI_COMM_WORLD_LOCAL_RANK, $SLURM_LOCALID) modulo the visible devices, else device 0;
 * device = SPDY_DEVICE_NONE builds a host-only plan (tables only: lets CPU-side tests inspect tables; every compute
 * call on it returns SPDY_ERR_NO_DEVICE).
 * max_batch bounds nb of every batched call.  Device memory beyond the tables (< 10 MB) is allocated on demand:
 * the host-pointer entry points stage through 4 x max_batch grids (+ 4 x 512 KB of pinned host memory for small calls,
 * $SPDY_HOST_STAGE_KB below) from their first call on; the four-kernel path
 * (spdy_plan_set_fused(0)) and the T63 operator+transform sequences
 * keep a workspace of max_batch x (il x 2mx + 2 mx nx complex) doubles, allocated at plan creation when it is <= 64 MB
 * (model-shaped plans: a graph capture then needs no warm-up) and at the first call that needs it otherwise (allocation
 * inside an open capture is refused with SPDY_ERR_STATE).  Destroying a plan also invalidates the graphs captured
 * from it (spdy_graph_launch then returns SPDY_ERR_STATE) and shuts down its communicators (spdy_comm_*: the handles
 * stay valid for spdy_comm_destroy, every other call on them returns SPDY_ERR_STATE).                              */
/* Environment read by the library (measurement and debugging aids; none changes results beyond rounding-level path choices).
 * The launch-policy switches (SPDY_T30_*, SPDY_T63_*, SPDY_WT_MIN_MB) are read ONCE per plan, in spdy_plan_create; a plan's
 * values are changed afterwards with spdy_plan_set_option.
 *   SPDY_DEVICE        device index for SPDY_DEVICE_AUTO              SPDY_FUSED = 0 | 1   initial spdy_plan_set_fused mode
 *   SPDY_WG_PER_CU     persistent workgroups per CU of the T30 kernels (default 1)
 *   SPDY_COMM_FORCE=1  issue the collectives even at world size 1
 *   SPDY_SHARD_TRANSPOSE=1  communicators created under it run the level-sharded step in its transposed form (spdy_comm_set_option)
 *   SPDY_T63_NOSPLIT   one workgroup per pair in small T63 direct launches instead of two (same bits)
 *   SPDY_T63_NODERIVE  T63 model-sized inverse batches run uvspec / grad as an operator kernel in front of the transform launch instead
 *                      of evaluating them where the transform loads its operands (agree to rounding)
 *   SPDY_T63_NOSTAGE   small T63 direct batches run fused (row FFTs inside the contraction launch) instead of staged (same bits)
 *   SPDY_T63_NP2_FROM  pairs from which the staged contraction takes two pairs per workgroup (default 40; same bits)
 *   SPDY_T30_NOSPLIT   small T30 direct batches as whole tiles instead of three workgroups per tile (same bits)
 *   SPDY_WT_MIN_MB     output size (MB, default 6) from which a model-sized launch writes its output through the L2s instead of
 *                      leaving it dirty for the end-of-kernel release (0 = never; same bits either way)
 *   SPDY_COMM_TIMEOUT_S  seconds (> 0; default 120; read when a group is created) an in-process collective waits for its missing ranks before it breaks the group
 *   SPDY_T30_NOPART    small T30 inverse launches walk whole tiles instead of (tile, third of the latitudes) items (same bits)
 *   SPDY_COMM_DRY=1    RCCL communicators created under it skip their collectives (timing a sharded step without its
 *                      exchanges; results are then wrong)
 *   SPDY_HOST_SPIN=0   host-staged calls (below) end in hipStreamSynchronize instead of spinning on a host-mapped completion stamp that
 *                      a one-thread kernel behind the call's kernels writes (the spin saves the runtime's wake-up: 1.3x the rate of
 *                      one-field calls at T30; it gives up after 2 s and falls back to the runtime)
 *   SPDY_HOST_STAGE_KB host-pointer calls whose largest array is at most this many KB (default 512; 0 = never) stage through
 *                      pinned host memory mapped into the device: the caller's thread copies in and out, the kernels read and
 *                      write the staging buffers across the link themselves (no copy-engine round trips: 1.4-1.6x the rate
 *                      of one-field calls); larger calls use device staging and hipMemcpyAsync.  Same bits either way.      */
enum { SPDY_MAX_KX = 32, SPDY_DEVICE_NONE = -1, SPDY_DEVICE_AUTO = -2 };
int spdy_plan_create(int trunc, int ix, int iy, int kx, int max_batch, int device, spdy_plan **plan);
int spdy_plan_destroy(spdy_plan *plan);
/* Half levels hsg[kx+1] (strictly increasing within [0,1]) for a level count the reference has no set for -- or
 * to override its set.  Rebuilds dhs, fsg, dhsr, fsgr (geometry.f90:51-60) and the tables derived from them
 * (geopotential.f90:22-30, horizontal_diffusion.f90:70-82); spdy_implicit_init must be called (again) afterwards. */
int spdy_plan_set_sigma(spdy_plan *plan, const double *hsg);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream); NULL restores the plan's own. */
int spdy_plan_set_stream(spdy_plan *plan, void *hip_stream);
int spdy_plan_synchronize(spdy_plan *plan);
/* Device memory for a host that has no HIP binding of its own (the Fortran model: fortran/time_stepping.f90 keeps the
 * reference's prognostics -- prognostics.f90:16-24 -- in HBM with these).  spdy_dev_alloc returns zero-filled memory on
 * the plan's device; upload/download are ordered on the plan's stream and return when the copy is complete; none of the
 * four may be called while a graph capture is open (SPDY_ERR_STATE).  A host with HIP (hipfort, torch) does not need them:
 * every *_dev entry point takes any device pointer.                                                                   */
int spdy_dev_alloc(spdy_plan *plan, size_t bytes, void **d_ptr);
int spdy_dev_free(spdy_plan *plan, void *d_ptr);
int spdy_dev_upload(spdy_plan *plan, void *d_dst, const void *src, size_t bytes);
int spdy_dev_download(spdy_plan *plan, void *dst, const void *d_src, size_t bytes);

/* Per-kernel timing for the roofline report: while on, every transform kernel launched by the
 * *_dev entry points is bracketed by HIP events on its own stream.  spdy_plan_get_profile
 * synchronises, adds up milliseconds and launch counts per kernel kind (arrays of
 * SPDY_K_COUNT) and clears the record.                                                       */
enum { SPDY_K_LEGENDRE_INV = 0, SPDY_K_FOURIER_INV = 1, SPDY_K_FOURIER_DIR = 2, SPDY_K_LEGENDRE_DIR = 3,
       SPDY_K_S2G_FUSED = 4, SPDY_K_G2S_FUSED = 5, SPDY_K_COUNT = 6 };
int spdy_plan_set_profiling(spdy_plan *plan, int on);
/* Kernel selection for the transforms: 1 and -1 (default) = fused single-pass kernels (T30, T63) at every batch size,
 * 0 = the four-kernel path (any resolution; the only path for other resolutions).  The two paths agree to rounding, not
 * bitwise; within one setting a field's bits never depend on the size or composition of the batch it travels in
 * (tests/test_gpu_determinism.py).  Fused launches whose grid-side array is >= 16 MB stream it with
 * non-temporal loads/stores (the data passes through the caches once); smaller ones leave it cached for their
 * consumer.                                                                                        */
int spdy_plan_set_fused(spdy_plan *plan, int mode);
/* Launch-policy switches of one plan (measurement and test aids; same results, see the environment list above -- the
 * environment gives a plan its initial values when it is created and is not read again).  name / value:
 *   "t30_part", "t30_split", "t63_split", "t63_stage", "t63_derive"   0 = the form named by $SPDY_T30_NOPART, $SPDY_T30_NOSPLIT,
 *                       $SPDY_T63_NOSPLIT, $SPDY_T63_NOSTAGE, $SPDY_T63_NODERIVE; 1 = the default form
 *   "t63_np2_from" >= 1, "wt_min_mb" >= 0                              as $SPDY_T63_NP2_FROM, $SPDY_WT_MIN_MB
 * Not while a graph capture is open (SPDY_ERR_STATE); captured graphs keep the forms they were captured with.       */
int spdy_plan_set_option(spdy_plan *plan, const char *name, int value);
int spdy_plan_get_profile(spdy_plan *plan, double *ms, int *launches);
/* Diagnostic: where the dispatcher places the eight waves of a workgroup shaped like the fused T63 kernels' (512 threads,
 * their LDS footprint, one workgroup per CU).  simd_of_wave[0..7] = SIMD of waves 0..7 of workgroup 0; violations = how many of
 * the launch's workgroups (one per CU) do NOT have waves w and w + 4 on one SIMD and waves 0..3 on four different ones.  The
 * T63 kernels assign their two wave roles to SIMDs on that assumption (csrc/spdy_fused_t63.inc: a tuning choice per kernel,
 * results do not depend on it); tests/test_gpu_determinism.py checks violations == 0.                                      */
int spdy_wave_placement(spdy_plan *plan, int *simd_of_wave, int *violations);
/* dims[0..7] = trunc, ix, iy, il, kx, nx, mx, max_batch */
int spdy_plan_dims(const spdy_plan *plan, int *dims);
const char *spdy_last_error(void);
/* Copy a named host table into buf (up to cap doubles); returns the element count or <0.
 * Names: sia_half coa_half cosgr cosgr2 hsg dhs fsg dhsr fsgr work ifac epsi wt poly nsh2
 *        el2 elm2 el4 trfilt gradx gradym gradyp uvdx uvdym uvdyp vddym vddyp
 *        dmp dmpd dmps dmp1 dmp1d dmp1s tref tref1 tref2 tref3 xc xd xj dhsx elz
 *        xgeop1 xgeop2 corf tcorv qcorv coriol                                             */
int spdy_get_table(const spdy_plan *plan, const char *name, double *buf, int cap);

/* ---- grid <-> spectral transforms --------------------------------------------------------
 * spec_to_grid(vorm,kcos)  spectral.f90:98-110  = fourier_inv(legendre_inv(.),kcos)
 * grid_to_spec(vorg)       spectral.f90:112-122 = legendre_dir(fourier_dir(.))
 * kcos semantics as fourier.f90:47-51: 1 = plain, anything else = multiply row j by cosgr(j). */
int spdy_spec_to_grid(spdy_plan *plan, const double *spec, int kcos, double *grid);
int spdy_grid_to_spec(spdy_plan *plan, const double *grid, double *spec);
/* nb independent fields; kcos[nb] per field (NULL = all 1). */
int spdy_spec_to_grid_batch(spdy_plan *plan, int nb, const double *spec, const int *kcos, double *grid);
int spdy_grid_to_spec_batch(spdy_plan *plan, int nb, const double *grid, double *spec);
/* Device-resident batch.  d_kcos: device int[nb] or NULL (then kcos_all applies to every field). */
int spdy_spec_to_grid_dev(spdy_plan *plan, int nb, const double *d_spec, const int *d_kcos, int kcos_all,
                          double *d_grid);
int spdy_grid_to_spec_dev(spdy_plan *plan, int nb, const double *d_grid, double *d_spec);

/* ---- transform stages (host pointers, batched) -------------------------------------------
 * legendre_inv legendre.f90:74-111 ; legendre_dir legendre.f90:114-155
 * fourier_inv  fourier.f90:23-53   ; fourier_dir  fourier.f90:56-82                          */
int spdy_legendre_inv(spdy_plan *plan, int nb, const double *spec, double *four);
int spdy_legendre_dir(spdy_plan *plan, int nb, const double *four, double *spec);
int spdy_fourier_inv(spdy_plan *plan, int nb, const double *four, int kcos, double *grid);
int spdy_fourier_dir(spdy_plan *plan, int nb, const double *grid, double *four);

/* ---- spectral-space operators (nb fields each) -------------------------------------------
 * laplacian spectral.f90:84 ; inverse_laplacian :91 ; trunct :229 (in place)
 * grad :124 ; vds :146 ; uvspec :173 ; vdspec :198 (kcos==2: *cosgr, else *cosgr2)            */
int spdy_laplacian(spdy_plan *plan, int nb, const double *in, double *out);
int spdy_inverse_laplacian(spdy_plan *plan, int nb, const double *in, double *out);
int spdy_trunct(spdy_plan *plan, int nb, double *inout);
int spdy_grad(spdy_plan *plan, int nb, const double *psi, double *psdx, double *psdy);
int spdy_vds(spdy_plan *plan, int nb, const double *ucosm, const double *vcosm, double *vorm, double *divm);
int spdy_uvspec(spdy_plan *plan, int nb, const double *vorm, const double *divm, double *ucosm, double *vcosm);
int spdy_vdspec(spdy_plan *plan, int nb, const double *ug, const double *vg, double *vorm, double *divm, int kcos);
int spdy_laplacian_dev(spdy_plan *plan, int nb, const double *in, double *out);
int spdy_inverse_laplacian_dev(spdy_plan *plan, int nb, const double *in, double *out);
int spdy_trunct_dev(spdy_plan *plan, int nb, double *inout);
int spdy_grad_dev(spdy_plan *plan, int nb, const double *psi, double *psdx, double *psdy);
int spdy_vds_dev(spdy_plan *plan, int nb, const double *ucosm, const double *vcosm, double *vorm, double *divm);
int spdy_uvspec_dev(spdy_plan *plan, int nb, const double *vorm, const double *divm, double *ucosm, double *vcosm);
/* device-pointer operators: outputs must not overlap inputs (except spdy_trunct_dev, which is in place);
 * vdspec at T30 is one kernel and writes its outputs while other inputs are still being read */
int spdy_vdspec_dev(spdy_plan *plan, int nb, const double *ug, const double *vg, double *vorm, double *divm, int kcos);

/* ---- spectral-space tail ------------------------------------------------------------------
 * do_horizontal_diffusion(field,fdt_in,dmp,dmp1) horizontal_diffusion.f90:86-105:
 *     fdt_out = (fdt_in - dmp*field)*dmp1, nlev levels sharing the two (mx,nx) real tables.
 * initialize_implicit(dt) implicit.f90:36-165 ; implicit_terms(divdt,tdt,psdt) :168-217
 * (in place; divdt,tdt are (mx,nx,kx), psdt is (mx,nx)).                                      */
int spdy_hdiff(spdy_plan *plan, int nlev, const double *field, const double *fdt_in, const double *dmp,
               const double *dmp1, double *fdt_out);
int spdy_hdiff_dev(spdy_plan *plan, int nlev, const double *field, const double *fdt_in, const double *d_dmp,
                   const double *d_dmp1, double *fdt_out);
/* The seven diffusion calls of one time step (time_stepping.f90:63-96) in one launch: nops <= SPDY_HDIFF_MAX_OPS
 * independent operations, each as spdy_hdiff_dev would do it (device pointers; ops is a host array).          */
enum { SPDY_HDIFF_MAX_OPS = 8 };
typedef struct {
    int nlev;
    const double *field, *fdt_in, *d_dmp, *d_dmp1;
    double *fdt_out;
} spdy_hdiff_op;
int spdy_hdiff_multi_dev(spdy_plan *plan, int nops, const spdy_hdiff_op *ops);
int spdy_implicit_init(spdy_plan *plan, double dt);
int spdy_implicit_terms(spdy_plan *plan, double *divdt, double *tdt, double *psdt);
int spdy_implicit_terms_dev(spdy_plan *plan, double *divdt, double *tdt, double *psdt);
/* Device copies of the plan's damping tables for spdy_hdiff_dev: name in
 * {dmp,dmpd,dmps,dmp1,dmp1d,dmp1s}; *d_ptr stays valid until the next spdy_implicit_init.     */
int spdy_device_table(spdy_plan *plan, const char *name, const double **d_ptr);

/* ---- spectral side of a time step (device-resident prognostics) ---------------------------------------------
 * With these a host keeps vor, div, t, ps, tr in HBM across steps: everything between the direct transforms of one
 * step and the inverse transforms of the next runs on the device (and can be captured into one graph).
 *   get_geopotential(t, phis) -> phi                         geopotential.f90:33-57
 *   get_spectral_tendencies(divdt, tdt, psdt, j2)            tendencies.f90:242-293: div, t, ps are the time-level-j2
 *       slabs of the prognostics, phis the surface geopotential; phi is written like the reference's module variable
 *   the diffusion block of step()                            time_stepping.f90:62-96: seven do_horizontal_diffusion
 *       calls with the orographic corrections ctmp = t + tcorh*tcorv(k), tr + qcorh*qcorv(k) and the stratospheric
 *       zonal-wind drag sdrag folded in; vor..tr are time level 1; tcorh/qcorh are device (mx,nx) complex arrays the
 *       host owns (forcing.f90 computes them); tr/trdt/qcorh may all be NULL (no tracer; the reference has ntr = 1)
 *   step_field_2d/3d(j1, dt, eps, input, fdt)                time_stepping.f90:121-167: leapfrog + Robert-Asselin-
 *       Williams filter; field is (mx,nx,nlev,2) with both time levels, fdt (mx,nx,nlev) is truncated in place as the
 *       reference does when ix == 4*iy; wil is the caller's params%wil.  Up to SPDY_STEP_MAX_OPS arrays per launch.   */
int spdy_geopotential(spdy_plan *plan, const double *t, const double *phis, double *phi);
int spdy_geopotential_dev(spdy_plan *plan, const double *t, const double *phis, double *phi);
int spdy_spectral_tendencies_dev(spdy_plan *plan, const double *div, const double *t, const double *ps, const double *phis,
                                 double *divdt, double *tdt, double *psdt, double *phi);
int spdy_hdiff_step_dev(spdy_plan *plan, const double *vor, const double *div, const double *t, const double *tr,
                        const double *d_tcorh, const double *d_qcorh, double sdrag,
                        double *vordt, double *divdt, double *tdt, double *trdt);
enum { SPDY_STEP_MAX_OPS = 8 };
typedef struct {
    int nlev;
    double *field, *fdt;
} spdy_step_op;
int spdy_step_fields_dev(spdy_plan *plan, int nops, const spdy_step_op *ops, int j1, double dt, double eps, double wil);
int spdy_step_field(spdy_plan *plan, int nlev, int j1, double dt, double eps, double wil, double *field, double *fdt);

/* ---- grid-space dynamical tendencies (tendencies.f90:105-197) --------------------------------------
 * Closes the loop inverse batch -> nonlinear terms -> direct batch on the device: from the gridded prognostics of time
 * level j2 (ug, vg, tg, vorg, divg, trg: (ix,il,kx) each, exactly what spdy_inverse_batch_dev produced -- vorg without
 * the Coriolis term, it is added here) and px, py = spec_to_grid(grad(ps), 2) it computes the vertical means, the
 * sigma-dot velocities and every grid tendency, laid out as the operands of ONE spdy_direct_batch_dev launch:
 *     u_out, v_out [3 kx] : (utend, vtend) | (-ug*tgg, -vg*tgg) | (-ug*trg, -vg*trg)
 *     plain_out [3 kx+1]  : 0.5*(ug^2+vg^2) | ttend | trtend | -umean*px - vmean*py
 * A host with physics adds its tendencies to utend, vtend, ttend, trtend in place before the direct batch
 * (tendencies.f90:203-206).  After the direct batch (pvor, pdiv [3 kx] from the pairs, pspec [3 kx + 1] from the plain
 * fields) spdy_tendency_combine_dev finishes tendencies.f90:125-126, 218-233 in place: pdiv[0:kx] -= laplacian(pspec[0:kx]),
 * pdiv[kx:3kx] += pspec[kx:3kx], pspec[3kx](1,1) = 0 -- so vordt = pvor[0:kx], divdt = pdiv[0:kx], tdt = pdiv[kx:2kx],
 * trdt = pdiv[2kx:3kx], psdt = pspec[3kx] are views, no copies.                                                     */
int spdy_grid_tendencies_dev(spdy_plan *plan, const double *ug, const double *vg, const double *tg, const double *vorg,
                             const double *divg, const double *trg, const double *px, const double *py, double *u_out,
                             double *v_out, double *plain_out);
int spdy_tendency_combine_dev(spdy_plan *plan, double *pdiv, double *pspec);
/* Everything after the direct batch in ONE launch (kx <= 16; the separate kernels otherwise): spdy_tendency_combine_dev,
 * spdy_spectral_tendencies_dev, spdy_implicit_terms_dev, spdy_hdiff_step_dev and spdy_step_fields_dev of ps, vor, div, t, tr.
 * Same expressions in the same order as the separate kernels; the results agree with them to rounding (the compiler is free
 * to contract a*b+c differently in the two translation contexts), not necessarily bit for bit.  pvor/pdiv/pspec: the direct batch's outputs (the final, truncated tendencies are left in
 * them); vor, div, t, tr: (mx,nx,kx,2), ps: (mx,nx,2) -- time level 1 feeds the spectral tendencies and the diffusion, as
 * in the reference's leapfrog step (tendencies.f90:34-38 with alph >= 0.5, time_stepping.f90:63-96).                    */
int spdy_spectral_step_dev(spdy_plan *plan, double *pvor, double *pdiv, double *pspec, double *vor, double *div, double *t,
                           double *tr, double *ps, const double *phis, const double *d_tcorh, const double *d_qcorh, double sdrag,
                           int j1, double dt, double eps, double wil, double *phi);
/* spdy_direct_batch_dev(3 kx pairs ug, vg -> pvor, pdiv; 3 kx + 1 plain fields d_grid -> pspec) followed by
 * spdy_spectral_step_dev, as ONE call: the second half of a time step (tendencies.f90:212-234, :242-293,
 * implicit.f90:168-217, time_stepping.f90:62-167).  At T30 exactly those two calls; at T63, where vds is not part of the
 * transform kernel, the pairs' spectra stay in the plan's temporaries and vds (spectral.f90:146-171) is applied where the
 * spectral step reads them: 2 launches instead of 3.  pvor, pdiv, pspec receive the truncated tendencies as with the
 * separate calls; results agree with them to rounding.                                                              */
int spdy_direct_batch_spectral_step_dev(spdy_plan *plan, const double *d_ug, const double *d_vg, const double *d_grid, int kcos,
                                        double *pvor, double *pdiv, double *pspec, double *vor, double *div, double *t, double *tr,
                                        double *ps, const double *phis, const double *d_tcorh, const double *d_qcorh, double sdrag,
                                        int j1, double dt, double eps, double wil, double *phi);

/* ---- multi-GPU: level-sharded time steps (one process per GPU with RCCL over xGMI, or ranks inside one process) -------
 * The transform batch shards over (field x level) with no communication.  What couples levels in a step is exchanged:
 * implicit_terms' operands (implicit.f90:174-216) with spdy_implicit_terms_sharded_dev, or -- the complete adiabatic step --
 * everything the two column kernels read, with spdy_sharded_step_dev below.
 * Rank r of n owns levels [nlev*r/n, nlev*(r+1)/n) (spdy_comm_level_range).  RCCL ranks: rank 0 obtains an id with
 * spdy_comm_unique_id and hands the SPDY_COMM_ID_BYTES bytes to the other ranks by whatever means the host has
 * (MPI_Bcast, a file, torch.distributed); then every rank calls spdy_comm_create.  Collectives run on the plan's
 * stream and can be captured into a graph.  RCCL is loaded on first use.                                        */
typedef struct spdy_comm spdy_comm;
enum { SPDY_COMM_ID_BYTES = 128, SPDY_COMM_MAX_ARRAYS = 16 };
int spdy_comm_unique_id(char *id);
int spdy_comm_create(spdy_plan *plan, int nranks, int rank, const char *id, spdy_comm **comm);
int spdy_comm_destroy(spdy_comm *comm);
int spdy_comm_level_range(const spdy_comm *comm, int nlev, int *lo, int *hi);
/* Ranks INSIDE one process -- a single-process host (the reference is one: speedy.f90:24-54) that drives several GPUs, one
 * host thread and one plan per GPU; also how the multi-rank paths are tested on a 1-GPU box (several plans on one device).
 * No RCCL: a collective blocks its calling thread until every rank of the group has entered it and moves the blocks with
 * device-to-device / peer copies on the ranks' own streams.  The calls of different ranks must come from different threads;
 * a rank missing for $SPDY_COMM_TIMEOUT_S (default 120) seconds breaks the group (SPDY_ERR_COMM) instead of hanging it.
 * Not graph-capturable (SPDY_ERR_STATE inside a capture).  Everything else -- level ranges, spdy_allgather_levels_dev, the
 * sharded step -- is the same code as with RCCL ranks.                                                                 */
typedef struct spdy_comm_group spdy_comm_group;
int spdy_comm_group_create(int nranks, spdy_comm_group **group);
int spdy_comm_group_destroy(spdy_comm_group *group);      /* after the group's communicators (SPDY_ERR_STATE otherwise) */
int spdy_comm_create_local(spdy_plan *plan, spdy_comm_group *group, int rank, spdy_comm **comm);
/* in place: d_full[i] are narr <= SPDY_COMM_MAX_ARRAYS full (mx,nx,nlev) stacks in which this rank has filled its own level block */
int spdy_allgather_levels_dev(spdy_comm *comm, int nlev, int narr, double *const *d_full);
/* all-gather of the level blocks of divdt and tdt, then implicit_terms on the full columns (every rank).  psdt must already be
 * the complete surface-pressure tendency on every rank -- true for implicit_terms called on its own, NOT inside a level-sharded
 * step, where get_spectral_tendencies (tendencies.f90:256-263) first sums the divergence of all levels into it.        */
int spdy_implicit_terms_sharded_dev(spdy_comm *comm, double *divdt, double *tdt, double *psdt);

/* The COMPLETE adiabatic time step with the transforms sharded by level (BASELINE config 3).  Lines of the reference that couple
 * levels: get_grid_point_tendencies' vertical means, sigma-dot sums and half-level fluxes (tendencies.f90:109-197),
 * get_spectral_tendencies' dmean / sigma-dot sums and psdt -= dmean (:256-285), the hydrostatic integration
 * (geopotential.f90:33-57), implicit_terms (implicit.f90:174-216).  Rank r runs the inverse and the direct transforms of ITS
 * levels only; the two column kernels run on full columns on every rank, each behind ONE in-place all-gather:
 *     inverse batch (own levels of time level j2; + grad(ps), level-free and replicated)  ->  all-gather of the gridded
 *     prognostics (6 kx grids)  ->  grid tendencies (full columns in, own levels' direct-batch operands out)  ->  direct batch
 *     (own levels; + the level-free ps tendency)  ->  all-gather of the spectral tendencies (9 kx + nranks spectra)  ->
 *     spectral step on full columns: tendency combination, get_spectral_tendencies, implicit_terms, diffusion, leapfrog.
 * Every rank holds the full prognostic state (vor, div, t, tr: (mx,nx,kx,2); ps: (mx,nx,2)) before and after; no rank reads a
 * level of an intermediate field that it neither computed nor received.  The exchanged stacks are stored block by block, one
 * contiguous block per rank (csrc/spdy_kernels.hpp: LevelShard), in workspace the communicator owns
 * (spdy_sharded_step_workspace: allocate before a graph capture).  kx <= 16, nranks <= kx, max_batch >= 4 kx + 2.
 * Results: the transforms are position-independent and the column kernels evaluate the unsharded kernels' expressions, so the
 * step agrees with spdy_inverse_batch_segs_dev + spdy_grid_tendencies_dev + spdy_direct_batch_spectral_step_dev to rounding
 * (in practice bit for bit, tests/test_gpu_sharded_step.py) for every rank count.
 *   spdy_sharded_step_dev          the whole step; tend_out (4 kx + 1 spectra: vordt | divdt | tdt | trdt | psdt, truncated, as
 *                                  step_field_* leaves them) may be NULL
 *   spdy_sharded_step_grid_dev     first half, up to the grid tendencies ...
 *   spdy_sharded_step_operands     ... whose results -- this rank's levels [lo, hi): u, v [3 nl], plain [3 nl + 1] grids laid
 *                                  out as spdy_grid_tendencies_dev documents, with nl for kx -- a host with physics updates in
 *                                  place (tendencies.f90:203-206) before ...
 *   spdy_sharded_step_spectral_dev ... the second half
 *   spdy_sharded_step_stacks       the two exchanged stacks (tests poison them between steps)                              */
int spdy_sharded_step_workspace(spdy_comm *comm);
int spdy_sharded_step_dev(spdy_comm *comm, double *vor, double *div, double *t, double *tr, double *ps, const double *phis,
                          const double *d_tcorh, const double *d_qcorh, double sdrag, int j1, int j2, double dt, double eps, double wil,
                          double *phi, double *tend_out);
int spdy_sharded_step_grid_dev(spdy_comm *comm, const double *vor, const double *div, const double *t, const double *tr,
                               const double *ps, int j2);
int spdy_sharded_step_operands(spdy_comm *comm, double **u, double **v, double **plain, int *lo, int *hi);
int spdy_sharded_step_spectral_dev(spdy_comm *comm, double *vor, double *div, double *t, double *tr, double *ps, const double *phis,
                                   const double *d_tcorh, const double *d_qcorh, double sdrag, int j1, double dt, double eps, double wil,
                                   double *phi, double *tend_out);
int spdy_sharded_step_stacks(spdy_comm *comm, double **grid_stack, size_t *grid_doubles, double **spec_stack, size_t *spec_doubles);
/* The TRANSPOSED form of the same step (spdy_comm_set_option(comm, "transpose", 1), or $SPDY_SHARD_TRANSPOSE=1 when the
 * communicator is created; the all-gather form above stays the default).  Instead of gathering whole stacks and running the
 * two column kernels redundantly on every rank, the ranks TRANSPOSE: the grid-space column kernel is coupled in the vertical
 * but independent in the horizontal (tendencies.f90:109-197), the spectral step independent per coefficient
 * (tendencies.f90:242-293, implicit.f90:168-217) -- so rank r runs them for ALL levels of ITS share of the grid points /
 * coefficients (contiguous ranges in units of 16):
 *     inverse batch (own levels)  ->  exchange 1: levels -> point ranges (6 kx slabs)  ->  grid tendencies, all levels of the own
 *     points  ->  exchange 2 back: every rank's direct-batch operands come home (9 nl + 1 grids)  ->  direct batch incl. vds (own
 *     levels)  ->  exchange 3: levels -> coefficient ranges (9 kx + nranks slabs)  ->  spectral step, all levels of the own
 *     coefficients  ->  exchange 4 back (time level j2 of the own levels; ps whole), issued at the START of the next grid half.
 * Each exchange is one grouped ncclSend / ncclRecv per peer on the plan's stream (in-process groups: 2-D peer copies).  Per rank
 * and step it receives (R - 1) / R of 6 nl + 9 nl + 1 grids and 9 nl + 1 + 4 nl + 1 spectra -- a third of the all-gather form's
 * bytes at 8 ranks (spdy_comm_describe prints both) -- and does 1 / R of the column kernels' work.  Same kernels' expressions
 * on the same values: bit-identical to the all-gather form and to the unsharded step (tests/test_gpu_sharded_step.py).
 * STATE: nothing is replicated any more.  After a transposed step the caller's prognostic arrays are current on (all levels x
 * own coefficients) only; the next spdy_sharded_step_grid_dev completes what it reads IN PLACE in the caller's arrays (which is
 * why they are not really const there), and spdy_sharded_state_gather_dev makes them whole on every rank (output, restart,
 * switching the form).  phi and tend_out are written on the own coefficients; spdy_sharded_gather_ranges_dev completes any
 * such array (rows of mx nx complex values; <= 4 kx + 2 rows).  The physics hook (spdy_sharded_step_operands between the two
 * halves) is unchanged.                                                                                                      */
int spdy_comm_set_option(spdy_comm *comm, const char *name, int value);      /* "transpose" 0 | 1; "force", "dry" as $SPDY_COMM_FORCE / _DRY */
int spdy_sharded_state_gather_dev(spdy_comm *comm, double *vor, double *div, double *t, double *tr, double *ps);
int spdy_sharded_gather_ranges_dev(spdy_comm *comm, int narr, double *const *arrays, const int *nrows);
/* One line of JSON about the communicator: route ("rccl" with the path of the librccl the symbols were bound from -- a host
 * process that carries two RCCL builds is a known source of hangs -- or the in-process group), rank / ranks, form, this rank's
 * levels / points / coefficients, bytes received per step in either form.                                                   */
int spdy_comm_describe(spdy_comm *comm, char *buf, int len);

/* ---- fused operator + transform sequences (device-resident; extensions of the reference interface) ------
 * The reference's callers always follow uvspec by two spec_to_grid(.,2) (tendencies.f90:98-100, physics.f90:96-98)
 * and grad by two spec_to_grid(.,2) (tendencies.f90:121-123).  These entry points do each sequence in one pass:
 * the operator is applied while the spectra are staged for the inverse transform, so the intermediate spectra
 * never exist in memory.  Results equal spdy_uvspec_dev / spdy_grad_dev followed by spdy_spec_to_grid_dev.
 *   spdy_uvspec_to_grid_dev : (vor, div)[nb] -> ug, vg [nb] grids     (spectral.f90:173-196 + :98-110)
 *   spdy_grad_to_grid_dev   : psi[nb]        -> gx, gy [nb] grids     (spectral.f90:124-144 + :98-110)          */
int spdy_uvspec_to_grid_dev(spdy_plan *plan, int nb, const double *d_vor, const double *d_div, double *d_ug, double *d_vg,
                            int kcos);
int spdy_grad_to_grid_dev(spdy_plan *plan, int nb, const double *d_psi, double *d_gx, double *d_gy, int kcos);
/* One model step's whole direct batch in one launch (tendencies.f90:212-234: vdspec of 3*kx (u,v) pairs next to
 * grid_to_spec of the remaining fields): at these sizes every launch costs its ~10 us pipeline latency, whatever it
 * carries.  Equals spdy_vdspec_dev(npairs, ...) followed by spdy_grid_to_spec_dev(nplain, ...); the two groups must
 * not overlap in memory.                                                                                       */
/* ... and its whole inverse batch (tendencies.f90:89-107: uvspec + two spec_to_grid(.,2) per level next to the
 * spec_to_grid of the other fields).  Equals spdy_uvspec_to_grid_dev(npairs, ..., kcos_pairs) followed by
 * spdy_spec_to_grid_dev(nplain, d_spec, d_kcos, kcos_all, d_grid).                                          */
int spdy_inverse_batch_dev(spdy_plan *plan, int npairs, const double *d_vor, const double *d_div, double *d_ug, double *d_vg,
                           int kcos_pairs, int nplain, const double *d_spec, const int *d_kcos, int kcos_all, double *d_grid);
int spdy_direct_batch_dev(spdy_plan *plan, int npairs, const double *d_ug, const double *d_vg, double *d_vorm,
                          double *d_divm, int kcos, int nplain, const double *d_grid, double *d_spec);
/* spdy_inverse_batch_dev followed by spdy_grad_to_grid_dev(ngrad, d_psi, d_gx, d_gy, kcos_grad) -- everything
 * get_grid_point_tendencies transforms to the grid before its level loops (tendencies.f90:89-107 and the grad(ps) /
 * spec_to_grid(.,2) pair of :121-123).  One launch at T30 (gradient tiles ride along in the mixed kernel); at T63 the
 * five groups of spectra are one fused launch behind the two operator kernels (a lone gradient is a one-workgroup
 * launch of a full pipeline latency).  npairs + ngrad <= max_batch for the one-launch T63 form.                                                                                                */
int spdy_inverse_batch_grad_dev(spdy_plan *plan, int npairs, const double *d_vor, const double *d_div, double *d_ug, double *d_vg,
                                int kcos_pairs, int nplain, const double *d_spec, const int *d_kcos, int kcos_all, double *d_grid,
                                int ngrad, const double *d_psi, double *d_gx, double *d_gy, int kcos_grad);
/* The same with the plain spectra taken from up to SPDY_MAX_SPEC_SEGS separate device arrays -- the reference reads vor, div,
 * t, tr of time level j2 straight from its four prognostic arrays (tendencies.f90:89-101), so a captured time step needs no
 * gather copy in front of this call.  Segment i contributes segs[i].nb fields; the grids of all segments are ONE stack
 * d_grid (sum of nb fields, in segment order); d_kcos (optional) is indexed along that stack.  Still one launch at T30 (the
 * mixed kernel looks the source array up per field) and one fused launch at T63 (one segment of the launch each).
 * ngrad may be 0.                                                                                                  */
enum { SPDY_MAX_SPEC_SEGS = 4 };
typedef struct {
    int nb;
    const double *d_spec;
} spdy_spec_seg;
int spdy_inverse_batch_segs_dev(spdy_plan *plan, int npairs, const double *d_vor, const double *d_div, double *d_ug, double *d_vg,
                                int kcos_pairs, int nseg, const spdy_spec_seg *segs, const int *d_kcos, int kcos_all, double *d_grid,
                                int ngrad, const double *d_psi, double *d_gx, double *d_gy, int kcos_grad);
/* the same with host pointers (one H2D + one D2H round trip for a whole level stack) */
int spdy_uvspec_to_grid(spdy_plan *plan, int nb, const double *vor, const double *div, double *ug, double *vg, int kcos);
int spdy_grad_to_grid(spdy_plan *plan, int nb, const double *psi, double *gx, double *gy, int kcos);

/* ---- output path: one snapshot's fields to the grid, scaled and rounded to float32 ------------------------------
 * input_output.f90:184-206: per level uvspec + spec_to_grid(.,2) of (vor, div), spec_to_grid(.,1) of t, q (= tr(:,:,:,1,1)),
 * phi, and of ps; then u, v, t unchanged, q*1.0e-3, phi/grav, p0*exp(ps), each converted with real(., sp).  The inputs are
 * time level 1 of the device-resident prognostics ((mx,nx,kx) complex; ps (mx,nx)); the outputs are float arrays
 * (ix,il,kx) / (ix,il) ready for the host's NetCDF writer (which stays on the host).  Three launches: gather of the plain
 * spectra, ONE transform launch for all 5 kx + 1 fields, float32 epilogue.  spdy_output_workspace allocates the plan's
 * scratch (5 kx + 1 grids) ahead of time, e.g. before a graph capture; max_batch must be >= 3 kx + 1.                    */
int spdy_output_workspace(spdy_plan *plan);
int spdy_output_batch_dev(spdy_plan *plan, const double *vor, const double *div, const double *t, const double *q, const double *phi,
                          const double *ps, float *u_out, float *v_out, float *t_out, float *q_out, float *phi_out, float *ps_out);

/* ---- HIP graphs: replaying a fixed sequence of device-resident calls --------------------------------
 * A model step is the same sequence of small launches every time (tendencies.f90:89-107, :212-234,
 * time_stepping.f90:56-121: ~90 inverse and ~70 direct transforms plus the spectral operators, 7 horizontal
 * diffusions and implicit_terms at T30 L8).  At those batch sizes a launch costs as much as the kernel, so the
 * sequence can be recorded once and replayed as one graph launch:
 *     spdy_graph_begin(plan);  <any *_dev calls on this plan>;  spdy_graph_end(plan, &g);  spdy_graph_launch(g); ...
 * Between begin and end nothing executes; the calls are captured from the plan's stream (which must not be the
 * legacy default stream) with the pointer arguments they were given.  Host-pointer entry points, profiling and
 * spdy_plan_synchronize are refused while a capture is open (SPDY_ERR_STATE).  spdy_graph_launch enqueues the
 * whole sequence on the plan's stream.  (The reference has no counterpart: it calls the transforms one field
 * at a time, spectral.f90:98-122.)                                                                          */
typedef struct spdy_graph spdy_graph;
int spdy_graph_begin(spdy_plan *plan);
int spdy_graph_end(spdy_plan *plan, spdy_graph **graph);
int spdy_graph_launch(spdy_graph *graph);
int spdy_graph_destroy(spdy_graph *graph);
/* Number of nodes of a captured graph (one per kernel launch / collective of the captured calls): what bench.py and the
 * step tests report as "launches per step". */
int spdy_graph_num_nodes(spdy_graph *graph, int *nodes);

#ifdef __cplusplus
}
#endif
#endif /* SPDY_H */
