// C ABI (include/spdy.h), third part: multi-GPU.  One process per GPU with RCCL over xGMI (spdy_comm_create), or -- for a
// single-process host that drives several GPUs from one thread each, and for the multi-rank tests on a 1-GPU box -- ranks
// inside one process that exchange by peer copies (spdy_comm_create_local).  On top of either: the level all-gather for
// implicit_terms and the COMPLETE level-sharded time step (both exchanges a level-sharded adiabatic step needs).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
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
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    std::string path;                 // the file the symbols were bound from (dladdr): two RCCL builds in one job is a known source of hangs
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
    SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
#undef SYM
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(r.AllGather), &info) && info.dli_fname) r.path = info.dli_fname;
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
    // the TRANSPOSED form of the step (spdy_comm_set_option "transpose"): levels <-> point ranges on the grid side, levels <->
    // coefficient ranges on the spectral side (see transpose_blocks)
    int transpose = 0;
    bool ranges_valid = false;        // the caller's prognostics are current only on (all levels x own coefficients): the next grid
                                      // half, or spdy_sharded_state_gather_dev, brings in what it reads
    double *Gb = nullptr;             // gridded prognostics, all levels of the own points: 6 kx slabs of npts
    double *Ob = nullptr;             // grid tendencies' results, all levels of the own points: 9 kx + nranks slabs of npts
    double *Tb = nullptr;             // direct batches' outputs, all levels of the own coefficients: 9 kx + nranks slabs of ne
    double *stage = nullptr;          // RCCL route: pack / unpack staging, (9 nl + 1) grids
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
        c->Gb = c->Ob = c->Tb = c->stage = nullptr;
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

/* Transposition between the two shardings of a level-block stack of F fields x levels (+ X level-free slabs per block;
 * LevelShard, csrc/spdy_kernels.hpp):
 *   by level : rank r holds ITS block -- nslab_r = F nl_r + X slabs -- over the whole horizontal domain (H doubles per slab);
 *   by range : rank r holds ALL blocks over ITS horizontal range [h0[r], h0[r] + len[r]) of every slab.
 * to_ranges: by level -> by range; otherwise back.  `lev` is the by-level array addressed like the whole stack (block r at slab
 * slab0_r, pitch H); `rng` the by-range array with pitch rp doubles per slab: len[rank] for the compact range stacks, H for a
 * whole array updated in place on a range (then rng == lev is allowed: what is read and what is written never overlap).
 * In-process groups: every rank PULLS its pieces with 2-D device copies (the protocol of allgather_blocks).  RCCL: one grouped
 * ncclSend / ncclRecv per peer on the plan's stream (graph-capturable) -- each rank's piece travels over its own xGMI link --
 * with the strided side packed / unpacked through c->stage by 2-D copies; force >= 1 sends the rank's own piece through RCCL too
 * (so that a 1-GPU box exercises the route).                                                                          */
int transpose_blocks(spdy_comm *c, bool to_ranges, int F, int X, size_t H, const size_t *h0, const size_t *len, double *lev, double *rng, size_t rp)
{
    spdy_plan *p = c->plan;
    if (c->dry) return SPDY_OK;
    const int R = c->nranks, me = c->rank, kx = p->tab.kx;
    std::vector<size_t> slab0(R), nslab(R);
    for (int r = 0; r < R; ++r) {
        int lo, hi;
        level_range(kx, r, R, &lo, &hi);
        slab0[r] = (size_t)F * lo + (size_t)X * r; nslab[r] = (size_t)F * (hi - lo) + X;
    }
    const size_t D = sizeof(double);
    auto copy2d = [&](double *dst, size_t dpitch, const double *src, size_t spitch, size_t width, size_t height, int src_dev) -> hipError_t {
        if (!width || !height) return hipSuccess;
        (void)src_dev;      // (peer access is enabled both ways when a rank joins its group; unified addressing finds the device)
        return hipMemcpy2DAsync(dst, dpitch * D, src, spitch * D, width * D, height, hipMemcpyDeviceToDevice, p->stream);
    };
    if (c->grp) {
        spdy_comm_group *g = c->grp;
        NOT_CAPTURING(p, "a collective of an in-process communicator (peer copies ordered by events of other ranks' streams)");
        g->arr[0][me] = to_ranges ? lev : rng;
        HIP_TRY_GROUP(g, hipEventRecord(g->ready[me], p->stream));
        RC(group_barrier(g));
        for (int q = 0; q < R; ++q) {
            if (q != me) HIP_TRY_GROUP(g, hipStreamWaitEvent(p->stream, g->ready[q], 0));
            const double *src = g->arr[0][q];
            if (to_ranges)      // block q, my range: out of q's by-level array
                HIP_TRY_GROUP(g, copy2d(rng + slab0[q] * rp, rp, src + slab0[q] * H + h0[me], H, len[me], nslab[q], g->device[q]));
            else {              // my block, range q: out of q's by-range array (its pitch: compact = len[q], in place = H)
                const size_t qp = rp == H ? H : len[q], qoff = rp == H ? h0[q] : 0;
                if (q == me && rp == H && rng == lev) continue;             // in place: the own piece is where it belongs
                HIP_TRY_GROUP(g, copy2d(lev + slab0[me] * H + h0[q], H, src + slab0[me] * qp + qoff, qp, len[q], nslab[me], g->device[q]));
            }
        }
        HIP_TRY_GROUP(g, hipEventRecord(g->done[me], p->stream));
        RC(group_barrier(g));
        for (int q = 0; q < R; ++q)
            if (q != me) HIP_TRY_GROUP(g, hipStreamWaitEvent(p->stream, g->done[q], 0));
        return SPDY_OK;
    }
    // RCCL (one process per GPU)
    const bool self_rccl = c->force >= 1;
    if (R > 1 || self_rccl) {
        if (!c->stage) return fail(SPDY_ERR_STATE, "no staging buffer (spdy_sharded_step_workspace)");
    }
    double *st = c->stage;
    if (to_ranges) {
        // pack: for every peer q the own block's slabs restricted to range q, contiguous [nslab_me][len_q] at st + nslab_me h0[q]
        for (int q = 0; q < R; ++q) {
            if (q == me && !self_rccl) HIP_TRY(copy2d(rng + slab0[me] * rp, rp, lev + slab0[me] * H + h0[me], H, len[me], nslab[me], p->device));
            else HIP_TRY(copy2d(st + nslab[me] * h0[q], len[q], lev + slab0[me] * H + h0[q], H, len[q], nslab[me], p->device));
        }
        if (R > 1 || self_rccl) {
            if (rp != len[me]) return fail(SPDY_ERR_ARG, "RCCL transposition to ranges needs a compact range stack");
            NCCL_TRY(rccl().GroupStart());
            for (int q = 0; q < R; ++q) {
                if (q == me && !self_rccl) continue;
                if (nslab[me] * len[q]) NCCL_GROUP_TRY(rccl().Send(st + nslab[me] * h0[q], nslab[me] * len[q], ncclDouble, q, c->comm, p->stream));
                if (nslab[q] * len[me]) NCCL_GROUP_TRY(rccl().Recv(rng + slab0[q] * rp, nslab[q] * len[me], ncclDouble, q, c->comm, p->stream));
            }
            NCCL_TRY(rccl().GroupEnd());
        }
    } else {
        const bool inplace = rp == H;
        if (R > 1 || self_rccl) {
            if (inplace)            // the by-range side is strided too: pack what every peer q gets (its block, my range) behind the receive area
                for (int q = 0; q < R; ++q)
                    if (q != me || self_rccl) HIP_TRY(copy2d(st + nslab[me] * H + slab0[q] * len[me], len[me], rng + slab0[q] * H + h0[me], H, len[me], nslab[q], p->device));
            NCCL_TRY(rccl().GroupStart());
            for (int q = 0; q < R; ++q) {
                if (q == me && !self_rccl) continue;
                const double *sendp = inplace ? st + nslab[me] * H + slab0[q] * len[me] : rng + slab0[q] * rp;
                if (nslab[q] * len[me]) NCCL_GROUP_TRY(rccl().Send(sendp, nslab[q] * len[me], ncclDouble, q, c->comm, p->stream));
                if (nslab[me] * len[q]) NCCL_GROUP_TRY(rccl().Recv(st + nslab[me] * h0[q], nslab[me] * len[q], ncclDouble, q, c->comm, p->stream));
            }
            NCCL_TRY(rccl().GroupEnd());
        }
        for (int q = 0; q < R; ++q) {       // unpack (own piece: straight from the by-range array unless it went through RCCL)
            if (q == me && !self_rccl) {
                if (!(inplace && rng == lev)) HIP_TRY(copy2d(lev + slab0[me] * H + h0[me], H, rng + slab0[me] * rp + (inplace ? h0[me] : 0), rp, len[me], nslab[me], p->device));
            } else HIP_TRY(copy2d(lev + slab0[me] * H + h0[q], H, st + nslab[me] * h0[q], len[q], len[q], nslab[me], p->device));
        }
    }
    return SPDY_OK;
}

// horizontal ranges of the transposed form, in units of the column kernels' 16-element blocks: [units r / R, units (r + 1) / R) x 16
void block_ranges(size_t total, int R, size_t mult, std::vector<size_t> &h0, std::vector<size_t> &len)
{
    const size_t nblk = (total + 15) / 16;
    h0.resize(R); len.resize(R);
    for (int r = 0; r < R; ++r) {
        const size_t b0 = nblk * r / R, b1 = nblk * (r + 1) / R;
        const size_t e0 = std::min(total, b0 * 16), e1 = std::min(total, b1 * 16);
        h0[r] = e0 * mult; len[r] = (e1 - e0) * mult;
    }
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
    if (const char *env = getenv("SPDY_SHARD_TRANSPOSE")) c->transpose = atoi(env) != 0;
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
    if (const char *env = getenv("SPDY_SHARD_TRANSPOSE")) c->transpose = atoi(env) != 0;
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
// the transposed form's horizontal ranges: grid points (doubles) and spectral coefficients (in doubles: 2 per coefficient)
struct Ranges { std::vector<size_t> g0, gl, s0, sl; };
Ranges ranges_of(const spdy_comm *c)
{
    Ranges r;
    block_ranges(grid_elems(c->plan), c->nranks, 1, r.g0, r.gl);
    block_ranges(spec_elems(c->plan) / 2, c->nranks, 2, r.s0, r.sl);
    return r;
}
/* All ranks end up with all ranges of the rows of `arr` ([nrows][H], every rank current on its own range [h0, h0 + len) of every
 * row): the gather that makes a range-sharded array whole again.                                                          */
int allgather_ranges(spdy_comm *c, double *arr, size_t nrows, size_t H, const size_t *h0, const size_t *len)
{
    spdy_plan *p = c->plan;
    if (c->dry || !nrows) return SPDY_OK;
    const int R = c->nranks, me = c->rank;
    const size_t D = sizeof(double);
    if (c->grp) {
        spdy_comm_group *g = c->grp;
        NOT_CAPTURING(p, "a collective of an in-process communicator (peer copies ordered by events of other ranks' streams)");
        g->arr[0][me] = arr;
        HIP_TRY_GROUP(g, hipEventRecord(g->ready[me], p->stream));
        RC(group_barrier(g));
        for (int q = 0; q < R; ++q) {
            if (q == me || !len[q]) continue;
            HIP_TRY_GROUP(g, hipStreamWaitEvent(p->stream, g->ready[q], 0));
            HIP_TRY_GROUP(g, hipMemcpy2DAsync(arr + h0[q], H * D, g->arr[0][q] + h0[q], H * D, len[q] * D, nrows, hipMemcpyDeviceToDevice, p->stream));
        }
        HIP_TRY_GROUP(g, hipEventRecord(g->done[me], p->stream));
        RC(group_barrier(g));
        for (int q = 0; q < R; ++q)
            if (q != me) HIP_TRY_GROUP(g, hipStreamWaitEvent(p->stream, g->done[q], 0));
        return SPDY_OK;
    }
    if (R == 1 && !c->force) return SPDY_OK;
    if (!c->stage) return fail(SPDY_ERR_STATE, "no staging buffer (spdy_sharded_step_workspace)");
    // compact [nrows][len_q] pieces side by side in the staging buffer: piece q at nrows h0[q]; one broadcast per rank
    double *st = c->stage;
    HIP_TRY(hipMemcpy2DAsync(st + nrows * h0[me], len[me] * D, arr + h0[me], H * D, len[me] * D, nrows, hipMemcpyDeviceToDevice, p->stream));
    NCCL_TRY(rccl().GroupStart());
    for (int q = 0; q < R; ++q)
        if (len[q]) NCCL_GROUP_TRY(rccl().Broadcast(st + nrows * h0[q], st + nrows * h0[q], nrows * len[q], ncclDouble, q, c->comm, p->stream));
    NCCL_TRY(rccl().GroupEnd());
    for (int q = 0; q < R; ++q)
        if (q != me && len[q])
            HIP_TRY(hipMemcpy2DAsync(arr + h0[q], H * D, st + nrows * h0[q], len[q] * D, len[q] * D, nrows, hipMemcpyDeviceToDevice, p->stream));
    return SPDY_OK;
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
    if (c->T && (!c->transpose || c->Tb)) return SPDY_OK;
    NOT_CAPTURING(p, "allocating the sharded step's workspace (call spdy_sharded_step_workspace before the capture)");
    const Shard s = shard_of(c);
    const int kx = p->tab.kx;
    const size_t P = (size_t)3 * s.nl;
    auto alloc0 = [&](double **dst, size_t n) -> int {         // zero-filled plan-owned device memory
        void *ptr;
        RC(dev_alloc(p, std::max<size_t>(n, 2) * sizeof(double), &ptr));
        HIP_TRY(hipMemsetAsync(ptr, 0, std::max<size_t>(n, 2) * sizeof(double), p->stream));
        *dst = static_cast<double *>(ptr);
        return SPDY_OK;
    };
    if (!c->T) {
        // (U | V | PL are ONE allocation: the rank's block of the F = 9, X = 1 operand stack of the transposed form)
        RC(alloc0(&c->G, (size_t)6 * kx * s.gs)); RC(alloc0(&c->px, s.gs)); RC(alloc0(&c->py, s.gs)); RC(alloc0(&c->U, (3 * P + 1) * s.gs));
        RC(alloc0(&c->tend, (size_t)(4 * kx + 1) * s.ss)); RC(alloc0(&c->T, (size_t)(9 * kx + c->nranks) * s.ss));
        c->V = c->U + P * s.gs;
        c->PL = c->V + P * s.gs;
    }
    if (c->transpose && !c->Tb) {
        // the transposed form's range stacks (all levels of the own points / coefficients) and the RCCL route's staging buffer
        const Ranges rg = ranges_of(c);
        const size_t npts = rg.gl[c->rank], ned = rg.sl[c->rank];
        RC(alloc0(&c->Gb, (size_t)6 * kx * npts)); RC(alloc0(&c->Ob, (size_t)(9 * kx + c->nranks) * npts));
        RC(alloc0(&c->stage, std::max((3 * P + 1) * s.gs, (size_t)(4 * kx + 2) * s.ss) + (size_t)(2 * kx + 2) * s.ss
                                 + (size_t)(9 * kx + c->nranks) * std::max(npts, ned)));
        RC(alloc0(&c->Tb, (size_t)(9 * kx + c->nranks) * ned));
    }
    return SPDY_OK;
}

int spdy_comm_set_option(spdy_comm *c, const char *name, int value)
{
    NEED_COMM(c);
    if (!name) return fail(SPDY_ERR_ARG, "null option name");
    if (c->plan->capturing) return fail(SPDY_ERR_STATE, "communicator options cannot change while a graph capture is open");
    const std::string n(name);
    if (n == "transpose") {
        if (c->ranges_valid && !value) return fail(SPDY_ERR_STATE, "the state is range-sharded: spdy_sharded_state_gather_dev first");
        c->transpose = value != 0;
    } else if (n == "force") c->force = value;
    else if (n == "dry") c->dry = value;
    else return fail(SPDY_ERR_ARG, "unknown communicator option '%s'", name);
    return SPDY_OK;
}

int spdy_comm_describe(spdy_comm *c, char *buf, int len)
{
    NEED_COMM(c);
    if (!buf || len < 1) return fail(SPDY_ERR_ARG, "null / empty buffer");
    const Shard s = shard_of(c);
    const Ranges rg = ranges_of(c);
    const int kx = c->plan->tab.kx, R = c->nranks, me = c->rank;
    // bytes this rank RECEIVES per step: the two all-gathers of the default form, the four exchanges of the transposed form
    const double ag = (double)(R - 1) / R * 8.0 * ((double)6 * kx * s.gs + (double)(9 * kx + R) * s.ss) * (R > 1);   // (blocks of the other ranks; ~)
    const double nl = s.nl;
    const double tr = 8.0 * ((6.0 * (kx - nl)) * rg.gl[me] + (9 * nl + 1) * (double)(s.gs - rg.gl[me]) + (9.0 * (kx - nl) + (R - 1)) * rg.sl[me]
                             + 4 * nl * (double)(s.ss - rg.sl[me]) + (double)(s.ss - rg.sl[me]));
    snprintf(buf, (size_t)len, "{\"route\": \"%s\", \"librccl\": \"%s\", \"nranks\": %d, \"rank\": %d, \"form\": \"%s\", "
             "\"levels\": [%d, %d], \"points\": [%zu, %zu], \"coefficients\": [%zu, %zu], "
             "\"bytes_received_per_step_allgather_form\": %.0f, \"bytes_received_per_step_transposed_form\": %.0f}",
             c->grp ? "in-process group (peer copies)" : "rccl", c->grp ? "" : rccl().path.c_str(), R, me, c->transpose ? "transpose" : "allgather",
             s.lo, s.hi, rg.g0[me], rg.g0[me] + rg.gl[me], rg.s0[me] / 2, (rg.s0[me] + rg.sl[me]) / 2, ag, tr);
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
    const Ranges rg = ranges_of(c);
    // (inside a graph capture the exchange is always recorded: a replayed step follows a transposed step; on a whole state it
    // moves values that are already there)
    if (c->transpose && (c->ranges_valid || p->capturing) && (c->nranks > 1 || c->force)) {
        // Transposed form, exchange 4 (coefficient ranges -> levels), done where its result is first read: the previous step
        // left the new prognostics on (all levels x own coefficients); this rank's inverse batch reads time level j2 of ITS
        // levels at all coefficients, and the level-free ps whole.  In place in the caller's arrays (the part written here is
        // exactly what this rank did not compute itself).
        const size_t slot = (size_t)(j2 - 1) * kx * s.ss;
        for (const double *arr : {vor, div, t, tr}) {
            double *a2 = const_cast<double *>(arr) + slot;
            RC(transpose_blocks(c, false, 1, 0, s.ss, rg.s0.data(), rg.sl.data(), a2, a2, s.ss));
        }
        RC(allgather_ranges(c, const_cast<double *>(ps) + (size_t)(j2 - 1) * s.ss, 1, s.ss, rg.s0.data(), rg.sl.data()));
    }
    const size_t lev = (size_t)(j2 - 1) * kx * s.ss + (size_t)s.lo * s.ss;     // this rank's levels of time level j2
    double *Gb = c->G + (size_t)6 * s.lo * s.gs;                                // its block: ug | vg | vorg | divg | tg | trg, nl each
    const spdy_spec_seg segs[4] = {{s.nl, vor + lev}, {s.nl, div + lev}, {s.nl, t + lev}, {s.nl, tr + lev}};
    RC(spdy_inverse_batch_segs_dev(p, s.nl, vor + lev, div + lev, Gb, Gb + (size_t)s.nl * s.gs, 2, 4, segs, nullptr, 1,
                                   Gb + (size_t)2 * s.nl * s.gs, 1, ps + (size_t)(j2 - 1) * s.ss, c->px, c->py, 2));
    if (c->transpose) {
        // exchange 1 (levels -> point ranges), the grid tendencies of ALL levels on the own points, exchange 2 (back: every
        // rank's direct-batch operands come home).  1 / R of the column kernel's work per rank, nothing replicated.
        const int me = c->rank;
        RC(transpose_blocks(c, true, 6, 0, s.gs, rg.g0.data(), rg.gl.data(), c->G, c->Gb, rg.gl[me]));
        spdy::GridTend gt{c->Gb, c->Gb, c->Gb, c->Gb, c->Gb, c->Gb, c->px, c->py, c->U, c->V, c->PL, spdy::LevelShard{c->nranks, c->rank},
                          (int)rg.gl[me], (int)rg.g0[me], c->Ob};
        if (rg.gl[me]) KERNEL(spdy::launch_grid_tendencies(p->dev, gt, p->stream));
        double *olev = c->U - ((size_t)9 * s.lo + me) * s.gs;                   // addressed like the whole operand stack: only the own block exists
        RC(transpose_blocks(c, false, 9, 1, s.gs, rg.g0.data(), rg.gl.data(), olev, c->Ob, rg.gl[me]));
        return SPDY_OK;
    }
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
    const bool raw = !c->transpose && use_raw63(p, P);     // (transposed form: vds needs whole rows -- applied here, before the exchange)
    if (raw) RC(direct_batch_raw63(p, P, c->U, c->V, 2, P + 1, c->PL, C, A, B));
    else RC(spdy_direct_batch_dev(p, P, c->U, c->V, A, B, 2, P + 1, c->PL, C));
    if (c->transpose) {
        // exchange 3 (levels -> coefficient ranges), then the spectral step of ALL levels on the own coefficients.  The new state
        // stays range-sharded: exchange 4 runs at the start of the next grid half (or spdy_sharded_state_gather_dev).
        const Ranges rg = ranges_of(c);
        const int me = c->rank;
        RC(transpose_blocks(c, true, 9, 1, s.ss, rg.s0.data(), rg.sl.data(), c->T, c->Tb, rg.sl[me]));
        if (!tend_out) tend_out = c->tend;
        spdy::SpecStep a{c->Tb, c->Tb, c->Tb, vor, div, t, tr, ps, phis, d_tcorh, d_qcorh, phi, sdrag, dt, eps, wil, j1,
                         p->tab.ix == 4 * p->tab.iy, nullptr, nullptr, spdy::LevelShard{c->nranks, c->rank}, tend_out,
                         (int)(rg.s0[me] / 2), (int)(rg.sl[me] / 2)};
        if (rg.sl[me]) KERNEL(spdy::launch_spectral_step(p->dev, a, p->stream));
        c->ranges_valid = c->nranks > 1 || c->force;
        return SPDY_OK;
    }
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

/* Every rank ends up with the complete arrays again: rows x (mx nx complex) arrays that the transposed form's spectral half
 * left current on (all rows x own coefficients) only -- the prognostics (2 kx rows each; ps: 2), phi (kx), the final tendencies
 * (4 kx + 1).  No-op for the all-gather form and for one rank.  Marks the state whole: the next grid half skips exchange 4.   */
int spdy_sharded_gather_ranges_dev(spdy_comm *c, int narr, double *const *arr, const int *nrows)
{
    NEED_COMM(c);
    spdy_plan *p = c->plan;
    NEED_DEVICE(p);
    if (narr < 0 || (narr && (!arr || !nrows))) return fail(SPDY_ERR_ARG, "bad argument");
    if (!c->transpose || (c->nranks == 1 && !c->force)) return SPDY_OK;
    RC(spdy_sharded_step_workspace(c));
    const Shard s = shard_of(c);
    const Ranges rg = ranges_of(c);
    for (int a = 0; a < narr; ++a) {
        if (!arr[a] || nrows[a] < 0 || nrows[a] > 4 * p->tab.kx + 2) return fail(SPDY_ERR_ARG, "array %d: null or more than 4 kx + 2 rows", a);
        RC(allgather_ranges(c, arr[a], (size_t)nrows[a], s.ss, rg.s0.data(), rg.sl.data()));
    }
    return SPDY_OK;
}

int spdy_sharded_state_gather_dev(spdy_comm *c, double *vor, double *div, double *t, double *tr, double *ps)
{
    NEED_COMM(c);
    if (!vor || !div || !t || !tr || !ps) return fail(SPDY_ERR_ARG, "null device pointer");
    const int kx = c->plan->tab.kx;
    double *arr[5] = {vor, div, t, tr, ps};
    const int rows[5] = {2 * kx, 2 * kx, 2 * kx, 2 * kx, 2};
    RC(spdy_sharded_gather_ranges_dev(c, 5, arr, rows));
    c->ranges_valid = false;
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
