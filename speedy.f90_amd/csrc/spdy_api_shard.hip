// C ABI (include/spdy.h), third part: multi-GPU.  One process per GPU with RCCL over xGMI (spdy_comm_create), or -- for a
// single-process host that drives several GPUs from one thread each, and for the multi-rank tests on a 1-GPU box -- ranks
// inside one process that exchange by peer copies (spdy_comm_create_local).  On top of either: the level all-gather for
// implicit_terms and the COMPLETE level-sharded time step (both exchanges a level-sharded adiabatic step needs).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "spdy_plan.hpp"

using namespace spdy_detail;

// RCCL is loaded on first use (dlopen by soname): single-GPU hosts never load it, and inside a process that already
// carries a librccl.so.1 (e.g. PyTorch's) the same instance is shared instead of a second copy being mapped.
namespace {
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl &rccl()
{
    static Rccl r;
    if (r.handle || !r.error.empty()) return r;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!r.handle) { r.error = std::string("cannot load librccl: ") + dlerror(); return r; }
#define SYM(field, name)                                                                       \
    if (!(r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name)))) { r.error = std::string("librccl lacks ") + name; r.handle = nullptr; return r; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather") SYM(Broadcast, "ncclBroadcast") SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    return r;
}
}  // namespace


// Ranks inside one process, one host thread per rank (each with its own plan; same or different devices).  A collective
// blocks its caller until every rank of the group has entered it -- the calls of different ranks must therefore come from
// different threads -- and moves the data with device-to-device (peer) copies on the callers' own streams:
//   rank r: record ready[r] behind the producer of its block; host barrier; for every peer q: wait for ready[q] on r's
//   stream and PULL q's block out of q's array; record done[r]; host barrier; wait for every done[q] (so that no rank
//   overwrites its block while a peer is still reading it).
// Eager only (cross-stream event waits of different captures cannot be recorded into one graph).
struct spdy_comm_group {
    int nranks = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long generation = 0;
    bool broken = false;
    int attached = 0;
    double timeout_s = 120.0;         // $SPDY_COMM_TIMEOUT_S at spdy_comm_group_create (a value that is not a positive number keeps the default)
    std::vector<spdy_comm *> member;
    std::vector<hipEvent_t> ready, done;
    std::vector<int> device;
    std::vector<double *> arr[SPDY_COMM_MAX_ARRAYS];      // the running collective's arrays, by rank
};

struct spdy_comm {
    spdy_plan *plan = nullptr;        // nullptr once the plan is gone: the communicator is then dead (SPDY_ERR_STATE)
    ncclComm_t comm = nullptr;        // RCCL (one process per GPU) ...
    spdy_comm_group *grp = nullptr;   // ... or ranks inside this process
    int nranks = 1, rank = 0;
    int force = 0;                    // $SPDY_COMM_FORCE (debug): 1 = issue the RCCL collectives even with one rank,
                                      // 2 = ... and take the ragged (per-rank broadcast) route for equal blocks too
    int dry = 0;                      // $SPDY_COMM_DRY=1 (measurement): collectives return at once -- what a sharded step costs
                                      // WITHOUT its exchanges (results are then wrong wherever another rank's block is read)
    // workspace of the level-sharded step (spdy_sharded_step_workspace; plan-owned device memory)
    double *G = nullptr;              // level-block stack of the gridded prognostics: 6 kx grids
    double *px = nullptr, *py = nullptr;
    double *U = nullptr, *V = nullptr, *PL = nullptr;   // this rank's direct-batch operands: 3 nl, 3 nl, 3 nl + 1 grids
    double *T = nullptr;              // level-block stack of the direct batches' outputs: 9 kx + nranks spectra
    double *tend = nullptr;           // final tendencies when the caller passes no array: 4 kx + 1 spectra
};

#define NCCL_TRY(expr)                                                                                     \
    do {                                                                                                   \
        ncclResult_t r_ = (expr);                                                                          \
        if (r_ != ncclSuccess) return fail(SPDY_ERR_COMM, "%s failed: %s", #expr, rccl().GetErrorString(r_)); \
    } while (0)
// inside ncclGroupStart/End: the group is closed before the error is returned (an open group would swallow or hang every
// later collective of the process)
#define NCCL_GROUP_TRY(expr)                                                                               \
    do {                                                                                                   \
        ncclResult_t r_ = (expr);                                                                          \
        if (r_ != ncclSuccess) {                                                                           \
            (void)rccl().GroupEnd();                                                                       \
            return fail(SPDY_ERR_COMM, "%s failed: %s", #expr, rccl().GetErrorString(r_));                 \
        }                                                                                                  \
    } while (0)

namespace spdy_detail {
// Plan teardown (spdy_plan_destroy, with the plan's stream still alive and idle): the plan's communicators are shut down and
// detached; their handles stay valid for spdy_comm_destroy, every other call on them fails with SPDY_ERR_STATE.
void release_comms(spdy_plan *p)
{
    for (spdy_comm *c : p->comms) {
        if (c->comm && rccl().handle) (void)rccl().CommDestroy(c->comm);
        c->comm = nullptr;
        c->plan = nullptr;
        c->G = c->px = c->py = c->U = c->V = c->PL = c->T = c->tend = nullptr;     // plan-owned memory: gone with the plan
    }
    p->comms.clear();
}
}  // namespace spdy_detail

#define NEED_COMM(c)                                                                                       \
    do {                                                                                                   \
        if (!(c)) return fail(SPDY_ERR_ARG, "null comm");                                                  \
        if (!(c)->plan) return fail(SPDY_ERR_STATE, "the communicator's plan has been destroyed");         \
    } while (0)

namespace {

inline void level_range(int nlev, int rank, int nranks, int *lo, int *hi)
{
    *lo = (int)(((long)nlev * rank) / nranks);
    *hi = (int)(((long)nlev * (rank + 1)) / nranks);
}

// host barrier of an in-process group; a rank that never arrives breaks the group instead of hanging the others for ever
int group_barrier(spdy_comm_group *g)
{
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->broken) return fail(SPDY_ERR_COMM, "in-process communicator group is broken (an earlier collective timed out)");
    const unsigned long gen = g->generation;
    if (++g->arrived == g->nranks) {
        g->arrived = 0;
        ++g->generation;
        g->cv.notify_all();
        return SPDY_OK;
    }
    const double limit = g->timeout_s;
    if (!g->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return g->generation != gen || g->broken; })) {
        const int arrived = g->arrived;
        g->broken = true;
        g->arrived = 0;
        g->cv.notify_all();
        return fail(SPDY_ERR_COMM, "in-process collective: %d of %d ranks arrived within %.0f s (each rank must call from its own thread)",
                    arrived, g->nranks, limit);
    }
    if (g->broken) return fail(SPDY_ERR_COMM, "in-process communicator group is broken");
    return SPDY_OK;
}

/* In place: each of the narr arrays d[a] is partitioned into nranks blocks, block r = doubles [off[r], off[r] + cnt[r]); this
 * rank has filled block `rank` of every array, afterwards every rank holds all blocks.  RCCL: ONE grouped operation on the
 * plan's stream (graph-capturable) -- equal, densely packed blocks: one in-place ncclAllGather per array (each rank's block
 * travels over its own xGMI link); anything else: one ncclBroadcast per rank and array.                              */
void group_break(spdy_comm_group *g)
{
    std::lock_guard<std::mutex> lk(g->mu);
    g->broken = true;
    g->cv.notify_all();
}
// (inside an in-process collective: a HIP error on this rank must not leave the peers waiting at the next barrier)
#define HIP_TRY_GROUP(g_, expr)                                                                            \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            group_break(g_);                                                                               \
            return fail(SPDY_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));                      \
        }                                                                                                  \
    } while (0)

int allgather_blocks(spdy_comm *c, int narr, double *const *d, const size_t *off, const size_t *cnt)
{
    spdy_plan *p = c->plan;
    if (narr == 0 || c->dry) return SPDY_OK;
    if (c->grp) {
        spdy_comm_group *g = c->grp;
        NOT_CAPTURING(p, "a collective of an in-process communicator (peer copies ordered by events of other ranks' streams)");
        for (int a = 0; a < narr; ++a) g->arr[a][c->rank] = d[a];
        HIP_TRY_GROUP(g, hipEventRecord(g->ready[c->rank], p->stream));
        RC(group_barrier(g));
        for (int q = 0; q < c->nranks; ++q) {
            if (q == c->rank || cnt[q] == 0) continue;
            HIP_TRY_GROUP(g, hipStreamWaitEvent(p->stream, g->ready[q], 0));
            for (int a = 0; a < narr; ++a) {
                if (g->device[q] == p->device)
                    HIP_TRY_GROUP(g, hipMemcpyAsync(d[a] + off[q], g->arr[a][q] + off[q], cnt[q] * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
                else
                    HIP_TRY_GROUP(g, hipMemcpyPeerAsync(d[a] + off[q], p->device, g->arr[a][q] + off[q], g->device[q], cnt[q] * sizeof(double), p->stream));
            }
        }
        HIP_TRY_GROUP(g, hipEventRecord(g->done[c->rank], p->stream));
        RC(group_barrier(g));
        // (the events belong to the GROUP and live until spdy_comm_group_destroy: a peer that has already finished its step and
        // destroyed its communicator cannot pull done[q] away from under this wait)
        for (int q = 0; q < c->nranks; ++q)
            if (q != c->rank) HIP_TRY_GROUP(g, hipStreamWaitEvent(p->stream, g->done[q], 0));
        return SPDY_OK;
    }
    if (c->nranks == 1 && !c->force) return SPDY_OK;
    bool even = c->force < 2;
    for (int r = 0; r < c->nranks; ++r) even = even && cnt[r] == cnt[0] && off[r] == off[0] + (size_t)r * cnt[0];
    NCCL_TRY(rccl().GroupStart());
    for (int a = 0; a < narr; ++a) {
        if (even) {
            if (cnt[0]) NCCL_GROUP_TRY(rccl().AllGather(d[a] + off[c->rank], d[a] + off[0], cnt[0], ncclDouble, c->comm, p->stream));
        } else {
            for (int r = 0; r < c->nranks; ++r)
                if (cnt[r]) NCCL_GROUP_TRY(rccl().Broadcast(d[a] + off[r], d[a] + off[r], cnt[r], ncclDouble, r, c->comm, p->stream));
        }
    }
    NCCL_TRY(rccl().GroupEnd());
    return SPDY_OK;
}

}  // namespace

extern "C" {

int spdy_comm_unique_id(char *id)
{
    static_assert(sizeof(ncclUniqueId) == SPDY_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id) return fail(SPDY_ERR_ARG, "null id");
    if (!rccl().handle) return fail(SPDY_ERR_COMM, "%s", rccl().error.c_str());
    ncclUniqueId u;
    NCCL_TRY(rccl().GetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return SPDY_OK;
}

int spdy_comm_create(spdy_plan *p, int nranks, int rank, const char *id, spdy_comm **comm)
{
    NEED_DEVICE(p);
    if (!comm) return fail(SPDY_ERR_ARG, "null comm pointer");
    *comm = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks || !id) return fail(SPDY_ERR_ARG, "bad rank %d of %d / null id", rank, nranks);
    NOT_CAPTURING(p, "spdy_comm_create");
    if (!rccl().handle) return fail(SPDY_ERR_COMM, "%s", rccl().error.c_str());
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    spdy_comm *c = new spdy_comm;
    c->plan = p; c->nranks = nranks; c->rank = rank;
    if (const char *env = getenv("SPDY_COMM_FORCE")) c->force = atoi(env);
    if (const char *env = getenv("SPDY_COMM_DRY")) c->dry = atoi(env);
    ncclResult_t r = rccl().CommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(SPDY_ERR_COMM, "ncclCommInitRank failed: %s", rccl().GetErrorString(r));
    }
    p->comms.push_back(c);
    *comm = c;
    return SPDY_OK;
}

int spdy_comm_group_create(int nranks, spdy_comm_group **grp)
{
    if (!grp) return fail(SPDY_ERR_ARG, "null group pointer");
    *grp = nullptr;
    if (nranks < 1 || nranks > 64) return fail(SPDY_ERR_ARG, "1..64 ranks per in-process group, not %d", nranks);
    spdy_comm_group *g = new spdy_comm_group;
    g->nranks = nranks;
    if (const char *env = getenv("SPDY_COMM_TIMEOUT_S")) { const double v = atof(env); if (v > 0.0) g->timeout_s = v; }   // once per group
    g->member.assign(nranks, nullptr);
    g->ready.assign(nranks, nullptr);
    g->done.assign(nranks, nullptr);
    g->device.assign(nranks, -1);
    for (auto &a : g->arr) a.assign(nranks, nullptr);
    *grp = g;
    return SPDY_OK;
}

int spdy_comm_group_destroy(spdy_comm_group *g)
{
    if (!g) return SPDY_OK;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        if (g->attached) return fail(SPDY_ERR_STATE, "%d communicators of the group are still alive", g->attached);
    }
    int caller_device = -1;
    const bool have_device = hipGetDevice(&caller_device) == hipSuccess;
    for (int q = 0; q < g->nranks; ++q) {                    // the ranks' events: created on first attach, owned by the group
        if (g->device[q] >= 0) (void)hipSetDevice(g->device[q]);
        if (g->ready[q]) (void)hipEventDestroy(g->ready[q]);
        if (g->done[q]) (void)hipEventDestroy(g->done[q]);
    }
    if (have_device) (void)hipSetDevice(caller_device);      // the calling thread keeps its device
    else (void)hipGetLastError();
    delete g;
    return SPDY_OK;
}

int spdy_comm_create_local(spdy_plan *p, spdy_comm_group *g, int rank, spdy_comm **comm)
{
    NEED_DEVICE(p);
    if (!comm) return fail(SPDY_ERR_ARG, "null comm pointer");
    *comm = nullptr;
    if (!g || rank < 0 || rank >= g->nranks) return fail(SPDY_ERR_ARG, "null group / rank %d outside it", rank);
    NOT_CAPTURING(p, "spdy_comm_create_local");
    std::lock_guard<std::mutex> lk(g->mu);
    if (g->member[rank]) return fail(SPDY_ERR_STATE, "rank %d of the group already exists", rank);
    if (g->ready[rank] && g->device[rank] != p->device) {       // the slot's events were made on another device: start afresh
        (void)hipSetDevice(g->device[rank]);
        (void)hipEventDestroy(g->ready[rank]); (void)hipEventDestroy(g->done[rank]);
        g->ready[rank] = g->done[rank] = nullptr;
        (void)hipSetDevice(p->device);
    }
    if (!g->ready[rank]) {
        // The events are the GROUP's (destroyed with it, not with the communicator): after the second barrier of a collective a
        // peer still issues hipStreamWaitEvent on done[rank] while this rank may already be tearing its communicator down.
        hipEvent_t ev[2] = {nullptr, nullptr};
        for (auto &e : ev) {
            const hipError_t er = hipEventCreateWithFlags(&e, hipEventDisableTiming);
            if (er != hipSuccess) {
                if (ev[0]) (void)hipEventDestroy(ev[0]);
                return fail(SPDY_ERR_HIP, "hipEventCreateWithFlags failed: %s", hipGetErrorString(er));
            }
        }
        g->ready[rank] = ev[0]; g->done[rank] = ev[1];
    }
    for (int q = 0; q < g->nranks; ++q)
        if (g->member[q] && g->device[q] != p->device) {        // peers on other devices: let the copy engines reach them, BOTH ways
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, p->device, g->device[q]);
            if (can) { (void)hipDeviceEnablePeerAccess(g->device[q], 0); (void)hipGetLastError(); }
            can = 0;
            (void)hipDeviceCanAccessPeer(&can, g->device[q], p->device);
            if (can) {                                          // (an earlier rank pulls from this one as well)
                (void)hipSetDevice(g->device[q]);
                (void)hipDeviceEnablePeerAccess(p->device, 0); (void)hipGetLastError();
                (void)hipSetDevice(p->device);
            }
        }
    spdy_comm *c = new spdy_comm;
    c->plan = p; c->grp = g; c->nranks = g->nranks; c->rank = rank;
    g->member[rank] = c; g->device[rank] = p->device;
    ++g->attached;
    p->comms.push_back(c);
    *comm = c;
    return SPDY_OK;
}

int spdy_comm_destroy(spdy_comm *c)
{
    if (!c) return SPDY_OK;
    if (spdy_plan *p = c->plan) {                           // still attached: drain the plan's stream, then shut RCCL down
        for (auto it = p->comms.begin(); it != p->comms.end(); ++it)
            if (*it == c) { p->comms.erase(it); break; }
        (void)hipSetDevice(p->device);
        if (!p->capturing) (void)hipStreamSynchronize(p->stream);
        if (c->comm && rccl().handle) (void)rccl().CommDestroy(c->comm);
    }
    if (spdy_comm_group *g = c->grp) {
        std::lock_guard<std::mutex> lk(g->mu);
        if (g->member[c->rank] == c) {                          // (its events stay with the group: peers may still wait on them)
            g->member[c->rank] = nullptr;
            --g->attached;
        }
    }
    delete c;
    return SPDY_OK;
}

int spdy_comm_level_range(const spdy_comm *c, int nlev, int *lo, int *hi)
{
    if (!c || !lo || !hi || nlev < 0) return fail(SPDY_ERR_ARG, "bad argument");
    level_range(nlev, c->rank, c->nranks, lo, hi);
    return SPDY_OK;
}

/* In place: every array d_full[i] is a full (mx,nx,nlev) stack of which this rank has filled its own level block
 * [lo, hi) (spdy_comm_level_range); afterwards every rank holds all levels.                                       */
int spdy_allgather_levels_dev(spdy_comm *c, int nlev, int narr, double *const *d_full)
{
    NEED_COMM(c);
    spdy_plan *p = c->plan;
    NEED_DEVICE(p);
    if (narr < 0 || narr > SPDY_COMM_MAX_ARRAYS || (narr && !d_full) || nlev < 0) return fail(SPDY_ERR_ARG, "bad argument");
    for (int a = 0; a < narr; ++a)
        if (!d_full[a]) return fail(SPDY_ERR_ARG, "null array %d", a);
    if (narr == 0 || nlev == 0) return SPDY_OK;
    std::vector<size_t> off(c->nranks), cnt(c->nranks);
    for (int r = 0; r < c->nranks; ++r) {
        int lo, hi;
        level_range(nlev, r, c->nranks, &lo, &hi);
        off[r] = (size_t)lo * spec_elems(p); cnt[r] = (size_t)(hi - lo) * spec_elems(p);
    }
    return allgather_blocks(c, narr, d_full, off.data(), cnt.data());
}

/* implicit_terms alone with the levels sharded over the ranks of `c` (implicit.f90:168-217 couples all levels of a
 * coefficient): gather the level blocks of divdt and tdt, then the solve on the full columns -- redundant on every rank, it is
 * a few microseconds.  psdt must be the complete surface-pressure tendency on every rank: in a level-sharded STEP it is not
 * (tendencies.f90:256-263 sums the divergence over all levels into it first) -- use spdy_sharded_step_dev there, which
 * exchanges everything get_spectral_tendencies and implicit_terms read.                                                  */
int spdy_implicit_terms_sharded_dev(spdy_comm *c, double *divdt, double *tdt, double *psdt)
{
    NEED_COMM(c);
    double *arr[2] = {divdt, tdt};
    RC(spdy_allgather_levels_dev(c, c->plan->tab.kx, 2, arr));
    return spdy_implicit_terms_dev(c->plan, divdt, tdt, psdt);
}

/* ---------------------------------------------------------------- the complete level-sharded time step
 * What a level-sharded adiabatic step exchanges (every line that couples levels in the reference):
 *   tendencies.f90:109-197  get_grid_point_tendencies: vertical means of u, v, div, the sigma-dot prefix sums and the
 *                           half-level fluxes read ALL levels of a column of ug, vg, tg, divg, trg
 *   tendencies.f90:256-285  get_spectral_tendencies: dmean and the sigma-dot sums read all levels of div; psdt -= dmean
 *   geopotential.f90:33-57  the hydrostatic integration reads all levels of t
 *   implicit.f90:174-216    three kx x kx mat-vecs per coefficient on divdt, tdt; psdt -= sum_k dhsx(k) divdt(k)
 * Design: the TRANSFORMS are sharded -- rank r runs the inverse and direct batches of its own levels only (that is where a
 * step's work is: 123 transforms against two column kernels of 5 us) -- and the two column kernels run redundantly on full
 * columns on every rank, each behind ONE in-place all-gather of the level-block stack its transform batch has filled:
 *   1 inverse batch, own levels of time level j2: (vor, div) -> ug, vg; vor, div, t, tr -> grid; + grad(ps) -> px, py (the
 *     level-free ps is replicated; every rank transforms it itself -- one more tile in its launch, no broadcast)
 *   2 all-gather of the grid stack G (6 kx grids; 221 KB per rank and level at T30)
 *   3 grid tendencies on full columns, writing this rank's direct-batch operands only
 *   4 direct batch, own levels: 3 nl (u, v) pairs + 3 nl + 1 plain fields (the +1: the level-free ps tendency, as in 1)
 *   5 all-gather of the spectral stack T (9 kx + nranks spectra)
 *   6 the one-launch spectral step on full columns: every rank ends the step with the complete new prognostic state
 * No rank ever reads a level it neither computed nor received in 2 or 5.                                          */
}  // extern "C"
namespace {
struct Shard { int lo, hi, nl; size_t gs, ss; };
Shard shard_of(const spdy_comm *c)
{
    Shard s;
    level_range(c->plan->tab.kx, c->rank, c->nranks, &s.lo, &s.hi);
    s.nl = s.hi - s.lo; s.gs = grid_elems(c->plan); s.ss = spec_elems(c->plan);
    return s;
}
int need_sharded(spdy_comm *c)
{
    spdy_plan *p = c->plan;
    const int kx = p->tab.kx;
    if (kx > 16) return fail(SPDY_ERR_UNSUPPORTED, "the level-sharded step needs kx <= 16 (one-launch column kernels), not %d", kx);
    if (c->nranks > kx) return fail(SPDY_ERR_ARG, "%d ranks for %d levels: every rank must own at least one level", c->nranks, kx);
    if (!p->tab.implicit_ready) return fail(SPDY_ERR_STATE, "the sharded step needs spdy_implicit_init first");
    if (!p->tab.sigma_ready) return fail(SPDY_ERR_STATE, "the sharded step needs sigma levels");
    if (p->max_batch < 4 * kx + 2) return fail(SPDY_ERR_ARG, "max_batch must be >= 4 kx + 2 for the sharded step");
    return SPDY_OK;
}
}  // namespace
extern "C" {

int spdy_sharded_step_workspace(spdy_comm *c)
{
    NEED_COMM(c);
    spdy_plan *p = c->plan;
    NEED_DEVICE(p);
    RC(need_sharded(c));
    if (c->T) return SPDY_OK;
    NOT_CAPTURING(p, "allocating the sharded step's workspace (call spdy_sharded_step_workspace before the capture)");
    const Shard s = shard_of(c);
    const int kx = p->tab.kx;
    const size_t P = (size_t)3 * s.nl;
    struct { double **dst; size_t n; } want[8] = {
        {&c->G, (size_t)6 * kx * s.gs}, {&c->px, s.gs}, {&c->py, s.gs}, {&c->U, P * s.gs}, {&c->V, P * s.gs}, {&c->PL, (P + 1) * s.gs},
        {&c->tend, (size_t)(4 * kx + 1) * s.ss}, {&c->T, (size_t)(9 * kx + c->nranks) * s.ss}};
    for (auto &w : want) {
        void *ptr;
        RC(dev_alloc(p, w.n * sizeof(double), &ptr));
        HIP_TRY(hipMemsetAsync(ptr, 0, w.n * sizeof(double), p->stream));
        *w.dst = static_cast<double *>(ptr);
    }
    return SPDY_OK;
}

int spdy_sharded_step_operands(spdy_comm *c, double **u, double **v, double **plain, int *lo, int *hi)
{
    NEED_COMM(c);
    RC(spdy_sharded_step_workspace(c));
    const Shard s = shard_of(c);
    if (u) *u = c->U;
    if (v) *v = c->V;
    if (plain) *plain = c->PL;
    if (lo) *lo = s.lo;
    if (hi) *hi = s.hi;
    return SPDY_OK;
}

int spdy_sharded_step_stacks(spdy_comm *c, double **grid_stack, size_t *grid_doubles, double **spec_stack, size_t *spec_doubles)
{
    NEED_COMM(c);
    RC(spdy_sharded_step_workspace(c));
    const Shard s = shard_of(c);
    if (grid_stack) *grid_stack = c->G;
    if (grid_doubles) *grid_doubles = (size_t)6 * c->plan->tab.kx * s.gs;
    if (spec_stack) *spec_stack = c->T;
    if (spec_doubles) *spec_doubles = (size_t)(9 * c->plan->tab.kx + c->nranks) * s.ss;
    return SPDY_OK;
}

int spdy_sharded_step_grid_dev(spdy_comm *c, const double *vor, const double *div, const double *t, const double *tr, const double *ps, int j2)
{
    NEED_COMM(c);
    spdy_plan *p = c->plan;
    NEED_DEVICE(p);
    RC(need_sharded(c));
    if (!vor || !div || !t || !tr || !ps) return fail(SPDY_ERR_ARG, "null device pointer");
    if (j2 != 1 && j2 != 2) return fail(SPDY_ERR_ARG, "j2 must be 1 or 2");
    RC(spdy_sharded_step_workspace(c));
    const Shard s = shard_of(c);
    const int kx = p->tab.kx;
    const size_t lev = (size_t)(j2 - 1) * kx * s.ss + (size_t)s.lo * s.ss;     // this rank's levels of time level j2
    double *Gb = c->G + (size_t)6 * s.lo * s.gs;                                // its block: ug | vg | vorg | divg | tg | trg, nl each
    const spdy_spec_seg segs[4] = {{s.nl, vor + lev}, {s.nl, div + lev}, {s.nl, t + lev}, {s.nl, tr + lev}};
    RC(spdy_inverse_batch_segs_dev(p, s.nl, vor + lev, div + lev, Gb, Gb + (size_t)s.nl * s.gs, 2, 4, segs, nullptr, 1,
                                   Gb + (size_t)2 * s.nl * s.gs, 1, ps + (size_t)(j2 - 1) * s.ss, c->px, c->py, 2));
    std::vector<size_t> off(c->nranks), cnt(c->nranks);
    for (int r = 0; r < c->nranks; ++r) {
        int lo, hi;
        level_range(kx, r, c->nranks, &lo, &hi);
        off[r] = (size_t)6 * lo * s.gs; cnt[r] = (size_t)6 * (hi - lo) * s.gs;
    }
    double *arr[1] = {c->G};
    RC(allgather_blocks(c, 1, arr, off.data(), cnt.data()));
    spdy::GridTend g{c->G, c->G, c->G, c->G, c->G, c->G, c->px, c->py, c->U, c->V, c->PL, spdy::LevelShard{c->nranks, c->rank}};
    KERNEL(spdy::launch_grid_tendencies(p->dev, g, p->stream));
    return SPDY_OK;
}

int spdy_sharded_step_spectral_dev(spdy_comm *c, double *vor, double *div, double *t, double *tr, double *ps, const double *phis,
                                   const double *d_tcorh, const double *d_qcorh, double sdrag, int j1, double dt, double eps, double wil,
                                   double *phi, double *tend_out)
{
    NEED_COMM(c);
    spdy_plan *p = c->plan;
    NEED_DEVICE(p);
    RC(need_sharded(c));
    if (!vor || !div || !t || !tr || !ps || !phis || !d_tcorh || !d_qcorh || !phi) return fail(SPDY_ERR_ARG, "null device pointer");
    if (j1 != 1 && j1 != 2) return fail(SPDY_ERR_ARG, "j1 must be 1 or 2");
    RC(spdy_sharded_step_workspace(c));
    const Shard s = shard_of(c);
    const int kx = p->tab.kx, P = 3 * s.nl;
    double *Tb = c->T + ((size_t)9 * s.lo + c->rank) * s.ss;                    // this rank's block: A | B | C (3 nl each) | psdt
    double *A = Tb, *B = Tb + (size_t)P * s.ss, *C = Tb + (size_t)2 * P * s.ss;
    const bool raw = use_raw63(p, P);
    if (raw) RC(direct_batch_raw63(p, P, c->U, c->V, 2, P + 1, c->PL, C, A, B));
    else RC(spdy_direct_batch_dev(p, P, c->U, c->V, A, B, 2, P + 1, c->PL, C));
    std::vector<size_t> off(c->nranks), cnt(c->nranks);
    for (int r = 0; r < c->nranks; ++r) {
        int lo, hi;
        level_range(kx, r, c->nranks, &lo, &hi);
        off[r] = ((size_t)9 * lo + r) * s.ss; cnt[r] = ((size_t)9 * (hi - lo) + 1) * s.ss;
    }
    double *arr[1] = {c->T};
    RC(allgather_blocks(c, 1, arr, off.data(), cnt.data()));
    if (!tend_out) tend_out = c->tend;
    spdy::SpecStep a{c->T, c->T, c->T, vor, div, t, tr, ps, phis, d_tcorh, d_qcorh, phi, sdrag, dt, eps, wil, j1,
                     p->tab.ix == 4 * p->tab.iy, raw ? c->T : nullptr, raw ? c->T : nullptr, spdy::LevelShard{c->nranks, c->rank}, tend_out};
    KERNEL(spdy::launch_spectral_step(p->dev, a, p->stream));
    return SPDY_OK;
}

int spdy_sharded_step_dev(spdy_comm *c, double *vor, double *div, double *t, double *tr, double *ps, const double *phis,
                          const double *d_tcorh, const double *d_qcorh, double sdrag, int j1, int j2, double dt, double eps, double wil,
                          double *phi, double *tend_out)
{
    RC(spdy_sharded_step_grid_dev(c, vor, div, t, tr, ps, j2));
    return spdy_sharded_step_spectral_dev(c, vor, div, t, tr, ps, phis, d_tcorh, d_qcorh, sdrag, j1, dt, eps, wil, phi, tend_out);
}

}  // extern "C"
