"""CPU, world_size 2 (gloo): the N>1 host logic of bench.py / a level-sharded model step.

The transforms themselves need the GPU; here the C oracle stands in as the executor (tests may use it)
so that the *sharding* logic is what is verified: contiguous (field x level) partition with no
collective reproduces the unsharded batch, the level all-gather reassembles implicit_terms' inputs
in the right order (also for ragged splits), and the timing reduction takes the slowest rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, kx, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import speedy_f90_amd as s
    from oracle.pyoracle import Oracle
    from golden.make_golden import tail_inputs
    sh = s.sharding
    o = Oracle(30, 96, 24, kx)
    # (1) batch sharding of the transform metric: B = 6 fields x kx levels, no collective
    B = 6 * kx
    lo, hi = sh.shard_range(B, rank, world)
    G = synth.grids(B, 96, 48, first=500)
    mine = np.stack([o.grid_to_spec(G[b]) for b in range(lo, hi)]) if hi > lo else np.zeros((0, 32, 31), complex)
    np.save(os.path.join(outdir, "spec_%d.npy" % rank), mine)
    # (2) level all-gather for the semi-implicit solve
    div, t, ps = tail_inputs(kx, 32, 31)
    klo, khi = sh.shard_range(kx, rank, world)
    full_div = sh.allgather_levels(torch.from_numpy(div[klo:khi].copy()), kx).numpy()
    full_t = sh.allgather_levels(torch.from_numpy(t[klo:khi].copy()), kx).numpy()
    assert np.array_equal(full_div, div) and np.array_equal(full_t, t)
    real = sh.allgather_levels(torch.from_numpy(div[klo:khi].real.copy()), kx).numpy()
    assert np.array_equal(real, div.real)
    o.tail_init(4800.0)
    d2, t2, p2 = o.implicit_terms(full_div, full_t, ps)
    np.save(os.path.join(outdir, "imp_%d.npy" % rank), np.concatenate([d2[klo:khi].ravel(), t2[klo:khi].ravel()]))
    # (3) timing contract: max over ranks
    assert sh.max_over_ranks(1.0 + rank) == float(world)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kx", [(2, 8), (2, 7)])
def test_two_rank_sharding(world, kx, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, kx, str(tmp_path)), nprocs=world, join=True)
    import speedy_f90_amd as s
    from oracle.pyoracle import Oracle
    from golden.make_golden import tail_inputs
    o = Oracle(30, 96, 24, kx)
    B = 6 * kx
    G = synth.grids(B, 96, 48, first=500)
    got = np.concatenate([np.load(tmp_path / ("spec_%d.npy" % r)) for r in range(world)])
    ref = np.stack([o.grid_to_spec(G[b]) for b in range(B)])
    assert got.shape == ref.shape and np.array_equal(got, ref)
    div, t, ps = tail_inputs(kx, 32, 31)
    o.tail_init(4800.0)
    d2, t2, _ = o.implicit_terms(div, t, ps)
    chunks = []
    for r in range(world):
        lo, hi = s.sharding.shard_range(kx, r, world)
        chunks.append(np.concatenate([d2[lo:hi].ravel(), t2[lo:hi].ravel()]))
        assert np.array_equal(np.load(tmp_path / ("imp_%d.npy" % r)), chunks[-1])


def test_shard_ranges_partition():
    import speedy_f90_amd as s
    for n in (0, 1, 7, 8, 48, 73, 91, 6144):
        for w in (1, 2, 4, 8):
            rs = [s.sharding.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs[:-1], rs[1:]))
            sizes = s.sharding.shard_sizes(n, w)
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n
