// Internal: the plan object behind include/spdy.h and the helpers shared by the C-ABI translation units
// (spdy_api.hip: plan, transforms, operators, graphs; spdy_api_step.hip: time-step tail, collectives, output).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/spdy.h"
#include "spdy_kernels.hpp"
#include "spdy_tables.hpp"

struct spdy_graph {
    struct spdy_plan *plan = nullptr;   // nullptr once the plan is gone: the graph is then dead (SPDY_ERR_STATE)
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

struct spdy_plan {
    spdy::HostTables tab;
    int max_batch = 0;
    int device = -1;
    spdy::DevPlan dev{};
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::vector<void *> allocs;       // everything hipMalloc'ed for this plan
    double *four = nullptr;           // [max_batch][il][fs] Fourier workspace (four-kernel path only; allocated on demand)
    // host-pointer API staging: four buffers of max_batch grids each, allocated at the first host-pointer call
    double *stage_a = nullptr, *stage_b = nullptr, *stage_c = nullptr, *stage_d = nullptr;   // the set the current call uses
    size_t stage_elems = 0;
    // Small host-pointer calls (the reference's one-field-per-call pattern, level stacks) skip both PCIe copy engines: their
    // staging buffers are PINNED HOST memory mapped into the device's address space -- the input is memcpy'ed into it by the
    // calling thread, the kernels read and write it across the link themselves, and the result is memcpy'ed out after the
    // stream synchronisation that ends every host-pointer call (`pending`).  $SPDY_HOST_STAGE_KB (default 512; 0 = off) is
    // the largest per-buffer size that takes this route; larger calls use the device set and hipMemcpyAsync as before.
    double *dstage[4] = {nullptr, nullptr, nullptr, nullptr};   // device memory, max_batch grids each
    double *hstage[4] = {nullptr, nullptr, nullptr, nullptr};   // hipHostMalloc (mapped, coherent), hstage_elems doubles each
    int *h_kcos = nullptr;                                      // hstage-route twin of d_kcos
    unsigned *h_stamp = nullptr;                                // host-mapped completion stamp of the hstage route (sync())
    unsigned stamp_seq = 0;
    size_t hstage_elems = 0;
    struct Pending { void *dst; const void *src; size_t bytes; };
    std::vector<Pending> pending;     // device-written host-stage results still to be copied to the caller's arrays
    const double *d_zero_spec = nullptr;         // one all-zero spectrum (gradient tiles of the mixed inverse kernel)
    double *tmp_c = nullptr, *tmp_d = nullptr;   // max_batch spectra each; allocated with `four` (multi-kernel operator sequences)
    double *out_grid = nullptr, *out_spec = nullptr;   // output path: (5kx+1) grids, (3kx+1) spectra (spdy_output_workspace)
    int *d_kcos = nullptr;
    // device copies of dt-dependent tables
    double *d_dmp[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double *d_xd = nullptr, *d_xc = nullptr, *d_xj = nullptr, *d_tref1 = nullptr, *d_dhsx = nullptr, *d_elz = nullptr;
    double *d_xt = nullptr;                      // xdt | xct | xjt (row-major, padded copies for the one-launch spectral step)
    double *d_levtab = nullptr;       // [LEVTAB_COUNT][kx] per-level tables (DevPlan::dhs ... qcorv)
    int num_cu = 256;
    int wg_per_cu = 1;                // fused kernels: one 448-thread wave-specialised workgroup per CU
    int fused_mode = -1;              // -1 auto, 0 four-kernel path, 1 fused kernels (T30 only)
    bool t63_derive = true;           // T63 model-sized inverse batches evaluate uvspec / grad on load ($SPDY_T63_NODERIVE, spdy_plan_set_option)
    // optional per-kernel timing (HIP events on the launch stream)
    bool profiling = false;
    bool capturing = false;           // between spdy_graph_begin and spdy_graph_end
    struct Span { int kind; hipEvent_t t0, t1; };
    std::vector<Span> spans;
    std::vector<spdy_graph *> graphs; // graphs captured from this plan that are still alive
    std::vector<struct spdy_comm *> comms;   // communicators created on this plan that are still alive (spdy_api_step.hip)
};

namespace spdy_detail {

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int dev_alloc(spdy_plan *p, size_t bytes, void **out);
int check_batch(const spdy_plan *p, int nb);
int h2d(spdy_plan *p, double *dst, const double *src, size_t n);
int d2h(spdy_plan *p, double *dst, const double *src, size_t n);
int sync(spdy_plan *p);
// host-pointer entry points call this first; elems = the largest array (in doubles) the call puts into ONE staging buffer:
// selects the host-mapped set when it fits (see spdy_plan::hstage), the device set otherwise / by default
int ensure_staging(spdy_plan *p, size_t elems = (size_t)-1);
inline bool on_host_stage(const spdy_plan *p) { return p->hstage[0] && p->stage_a == p->hstage[0]; }
int ensure_four(spdy_plan *p);        // four-kernel path workspace
// T63 fused path: the direct batch WITHOUT vds -- the scaled (u, v) grids' spectra go to raw_u / raw_v (null: p->tmp_c /
// p->tmp_d), the plain grids' to spec (spdy_direct_batch_spectral_step_dev, the level-sharded step)
int direct_batch_raw63(spdy_plan *p, int npairs, const double *ug, const double *vg, int kcos, int nplain, const double *grid, double *spec,
                       double *raw_u = nullptr, double *raw_v = nullptr);
bool use_raw63(const spdy_plan *p, int npairs);   // whether a step's direct batch of npairs (u, v) pairs takes that route
int upload_level_tables(spdy_plan *p);
void release_comms(spdy_plan *p);     // plan teardown: RCCL communicators of this plan are shut down, their handles stay valid but dead

inline size_t spec_elems(const spdy_plan *p) { return (size_t)2 * p->tab.mx * p->tab.nx; }
inline size_t grid_elems(const spdy_plan *p) { return (size_t)p->tab.ix * p->tab.il; }
inline size_t four_elems(const spdy_plan *p) { return (size_t)2 * p->tab.mx * p->tab.il; }

}  // namespace spdy_detail

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return spdy_detail::fail(SPDY_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define NEED_PLAN(p)                                                                   \
    do {                                                                               \
        if (!(p)) return spdy_detail::fail(SPDY_ERR_ARG, "null plan");                 \
    } while (0)
#define NEED_DEVICE(p)                                                                                       \
    do {                                                                                                     \
        NEED_PLAN(p);                                                                                        \
        if ((p)->device < 0) return spdy_detail::fail(SPDY_ERR_NO_DEVICE, "host-only plan: no HIP device, no CPU fallback"); \
        HIP_TRY(hipSetDevice((p)->device));                                                                  \
    } while (0)
#define NOT_CAPTURING(p, what)                                                                               \
    do {                                                                                                     \
        if ((p)->capturing) return spdy_detail::fail(SPDY_ERR_STATE, "%s is not possible while a graph capture is open", what); \
    } while (0)
#define RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)
#define KERNEL(expr) HIP_TRY(expr)
