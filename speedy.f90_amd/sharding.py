"""Multi-GPU host logic: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests).

The per-time-step transform batch is indexed by b = (field, level) and every 2-D transform is
independent, so the batch is partitioned contiguously -- rank r of R owns
b in [r*B/R, (r+1)*B/R) -- with all tables replicated and NO data-path collective.  The only
spectral-space step that couples levels is implicit_terms (implicit.f90:174-216): when tendencies
are sharded by level, its inputs are completed with one all-gather of (divdt, tdt) level slabs
(psdt is level-free and replicated).
"""
import torch
import torch.distributed as dist


def shard_range(nitems, rank, world):
    """Contiguous block partition; ranks differ by at most one item."""
    lo = (nitems * rank) // world
    hi = (nitems * (rank + 1)) // world
    return lo, hi


def shard_sizes(nitems, world):
    return [shard_range(nitems, r, world)[1] - shard_range(nitems, r, world)[0] for r in range(world)]


def max_over_ranks(seconds, device=None):
    """bench.py contract: the job's time is the slowest rank's time."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_levels(local, kx):
    """local: [k_local, nx, mx] complex tensor holding this rank's contiguous level block (shard_range
    over kx).  Returns the full [kx, nx, mx] tensor on every rank.  Ragged blocks (kx % world != 0) are
    padded to the largest block for the collective and trimmed afterwards."""
    world = dist.get_world_size()
    sizes = shard_sizes(kx, world)
    kmax = max(sizes)
    pad = torch.zeros((kmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * kmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    # view_as_real: NCCL/RCCL has no complex dtype; the (re,im) pairs travel as doubles
    if local.is_complex():
        dist.all_gather_into_tensor(torch.view_as_real(out), torch.view_as_real(pad))
    else:
        dist.all_gather_into_tensor(out, pad)
    parts = [out[r * kmax: r * kmax + sizes[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


class LevelComm:
    """The C-ABI communicator (include/spdy.h: spdy_comm_*): direct RCCL collectives on the plan's stream, no torch
    ops, graph-capturable.  One per process/GPU; the RCCL unique id travels over the already initialised
    torch.distributed group (any backend) -- a Fortran/MPI host would MPI_Bcast it instead."""

    def __init__(self, sp):
        import ctypes
        from ._lib import check
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
        ident = ctypes.create_string_buffer(128)
        if rank == 0:
            check(sp.lib.spdy_comm_unique_id(ident))
        if world > 1:
            box = [bytes(ident.raw)]
            dist.broadcast_object_list(box, src=0)
            ident = ctypes.create_string_buffer(box[0], 128)
        h = ctypes.c_void_p()
        check(sp.lib.spdy_comm_create(sp.h, world, rank, ident, ctypes.byref(h)))
        self.sp, self.h, self.rank, self.world = sp, h, rank, world

    def level_range(self, nlev):
        return shard_range(nlev, self.rank, self.world)

    def allgather_levels_(self, *full):
        """In place: each tensor is a full [nlev, nx, mx] stack in which this rank has filled its own level block."""
        import ctypes
        from ._lib import check
        self.sp._sync_stream()
        arr = (ctypes.c_void_p * len(full))(*[t.data_ptr() for t in full])
        check(self.sp.lib.spdy_allgather_levels_dev(self.h, full[0].shape[0], len(full), arr))

    def implicit_terms_sharded_(self, divdt_full, tdt_full, psdt):
        from ._lib import check
        self.sp._sync_stream()
        check(self.sp.lib.spdy_implicit_terms_sharded_dev(self.h, divdt_full.data_ptr(), tdt_full.data_ptr(), psdt.data_ptr()))

    def close(self):
        if self.h:
            self.sp.lib.spdy_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sharded_implicit_terms(sp, divdt_local, tdt_local, psdt, comm=None):
    """Level-sharded semi-implicit correction (implicit.f90:168-217 couples all levels of a coefficient): complete
    the level stacks with one all-gather, run implicit_terms on the full columns (independent per spectral
    coefficient, a few microseconds, done redundantly on every rank), keep this rank's levels.
    sp: speedy_f90_amd.Spectral with initialize_implicit() done; tensors live on sp's device.
    comm: a LevelComm -> the gather is one grouped RCCL call on the same stream as the solve (include/spdy.h);
    without it the gather goes through torch.distributed.  Either way every kernel and collective is ordered on
    torch's current stream (Spectral follows it), so the returned tensors are safe to use from torch."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(sp.kx, rank, world)
    if comm is not None:
        shape = (sp.kx,) + tuple(divdt_local.shape[1:])
        div = torch.empty(shape, dtype=divdt_local.dtype, device=divdt_local.device)
        t = torch.empty(shape, dtype=tdt_local.dtype, device=tdt_local.device)
        div[lo:hi].copy_(divdt_local)
        t[lo:hi].copy_(tdt_local)
        ps = psdt.clone()
        comm.implicit_terms_sharded_(div, t, ps)
        return div[lo:hi], t[lo:hi], ps
    if world > 1:
        div = allgather_levels(divdt_local, sp.kx).contiguous()
        t = allgather_levels(tdt_local, sp.kx).contiguous()
    else:
        div, t = divdt_local.clone(), tdt_local.clone()
    ps = psdt.clone()
    sp.implicit_terms_dev(div, t, ps)
    return div[lo:hi], t[lo:hi], ps
