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


def sharded_implicit_terms(sp, divdt_local, tdt_local, psdt):
    """Level-sharded semi-implicit correction: all-gather the level slabs, run implicit_terms on the
    full columns (it is independent per spectral coefficient, cheap), keep this rank's levels.
    sp: speedy_f90_amd.Spectral with initialize_implicit() done; tensors live on sp's device."""
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(sp.kx, rank, world)
    div = allgather_levels(divdt_local, sp.kx).contiguous()
    t = allgather_levels(tdt_local, sp.kx).contiguous()
    ps = psdt.clone()
    sp.implicit_terms_dev(div, t, ps)
    return div[lo:hi], t[lo:hi], ps
