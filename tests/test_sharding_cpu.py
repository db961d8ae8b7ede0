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


def _step_worker(rank, world, port, kx, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import speedy_f90_amd as s
    from oracle.pyoracle import Oracle
    from dynstep import ROB, SDRAG, WIL, state
    o = Oracle(30, 96, 24, kx)
    o.tail_init(2400.0)
    st = state(o, 8000)                       # the full prognostic state on every rank (the tail is replicated) ...
    for step in range(2):                     # ... of which a rank only ever TRANSFORMS its own levels
        st, fin = s.sharding.sharded_step_host(o, st, rank, world, 2, 2, 2400.0, ROB, WIL, SDRAG)
    np.savez(os.path.join(outdir, "step_%d.npz" % rank), **{n: st[n] for n in ("vor", "div", "t", "tr", "ps")},
             **{n: fin[n] for n in ("vordt", "divdt", "tdt", "trdt", "psdt", "phi", "U", "V", "PL")})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kx", [(2, 8), (3, 8), (8, 8)])      # 8 ranks x 8 levels: BASELINE config 3's own shape, one level per rank
def test_multi_rank_sharded_step(world, kx, tmp_path):
    """The COMPLETE level-sharded adiabatic step (speedy_f90_amd.sharding.sharded_step_host: the data flow of
    spdy_sharded_step_dev with the oracle as the executor of the single procedures) over gloo: each rank transforms only its own
    levels; the two level-block stacks are completed by one gather each; get_grid_point_tendencies, get_spectral_tendencies,
    implicit_terms, the diffusion and the leapfrog run on full columns.  After two chained steps every rank must hold the
    UNSHARDED oracle step's prognostics and tendencies bit for bit -- equal (world 2) and ragged (world 3, 8 levels) blocks.
    This is the test that catches a level coupling left out of the exchange (round 3: psdt / get_spectral_tendencies)."""
    port = _free_port()
    mp.spawn(_step_worker, args=(world, port, kx, str(tmp_path)), nprocs=world, join=True)
    import speedy_f90_amd as s
    from oracle.pyoracle import Oracle
    from dynstep import ROB, SDRAG, WIL, oracle_dynamics_step, state
    o = Oracle(30, 96, 24, kx)
    o.tail_init(2400.0)
    ref = state(o, 8000)
    for step in range(2):
        ref, out = oracle_dynamics_step(o, ref, 2, 2400.0, ROB)
    for r in range(world):
        z = np.load(tmp_path / ("step_%d.npz" % r))
        for n in ("vor", "div", "t", "tr", "ps"):
            assert np.array_equal(z[n], ref[n]), (r, n)
        for n in ("vordt", "divdt", "tdt", "trdt", "psdt", "phi"):
            assert np.array_equal(z[n], out[n]), (r, n)
        lo, hi = s.sharding.shard_range(kx, r, world)
        own = [g * kx + k for g in range(3) for k in range(lo, hi)]
        assert np.array_equal(z["U"], out["U"][own]) and np.array_equal(z["V"], out["V"][own])
        assert np.array_equal(z["PL"], np.concatenate([out["PL"][own], out["PL"][3 * kx:]]))


def _transposed_worker(rank, world, port, kx, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import speedy_f90_amd as s
    from oracle.pyoracle import Oracle
    from dynstep import ROB, SDRAG, WIL, state
    o = Oracle(30, 96, 24, kx)
    o.tail_init(2400.0)
    st = state(o, 8000)
    for step in range(2):                     # the second step starts with exchange 4 in the rank's own arrays
        st, fin = s.sharding.sharded_step_host_transposed(o, st, rank, world, 2, 2, 2400.0, ROB, WIL, SDRAG, ranges_valid=step > 0)
    np.savez(os.path.join(outdir, "tstep_%d.npz" % rank), c0=fin["range"][0], c1=fin["range"][1],
             **{n: st[n] for n in ("vor", "div", "t", "tr", "ps")},
             **{n: fin[n] for n in ("vordt", "divdt", "tdt", "trdt", "psdt", "phi", "U", "V", "PL")})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kx", [(2, 8), (3, 8), (8, 8)])
def test_multi_rank_transposed_step(world, kx, tmp_path):
    """The TRANSPOSED form of the level-sharded step (sharding.sharded_step_host_transposed: the data flow of spdy_sharded_step_dev
    with spdy_comm_set_option "transpose", the oracle as executor) over gloo: levels <-> point ranges around
    get_grid_point_tendencies (tendencies.f90:109-197), levels <-> coefficient ranges around get_spectral_tendencies /
    implicit_terms / diffusion / leapfrog (tendencies.f90:242-293, implicit.f90:168-217, time_stepping.f90:56-167).  After two
    chained steps every rank must hold, on ITS coefficients, the unsharded oracle step's prognostics and tendencies bit for bit,
    the ranges must tile the spectrum, and its levels' direct-batch operands must have come home complete."""
    port = _free_port()
    mp.spawn(_transposed_worker, args=(world, port, kx, str(tmp_path)), nprocs=world, join=True)
    import speedy_f90_amd as s
    from oracle.pyoracle import Oracle
    from dynstep import ROB, SDRAG, WIL, oracle_dynamics_step, state
    o = Oracle(30, 96, 24, kx)
    o.tail_init(2400.0)
    ref = state(o, 8000)
    for step in range(2):
        ref, out = oracle_dynamics_step(o, ref, 2, 2400.0, ROB)
    flat = lambda a: a.reshape(a.shape[:-2] + (-1,))
    covered = 0
    for r in range(world):
        z = np.load(tmp_path / ("tstep_%d.npz" % r))
        c0, c1 = int(z["c0"]), int(z["c1"])
        assert (c0, c1) == s.sharding.block_ranges(32 * 31, world)[r]
        covered += c1 - c0
        for n in ("vor", "div", "t", "tr", "ps"):
            assert np.array_equal(flat(z[n])[..., c0:c1], flat(ref[n])[..., c0:c1]), (r, n)
        for n in ("vordt", "divdt", "tdt", "trdt", "psdt", "phi"):
            assert np.array_equal(flat(z[n])[..., c0:c1], flat(out[n])[..., c0:c1]), (r, n)
        lo, hi = s.sharding.shard_range(kx, r, world)
        own = [g * kx + k for g in range(3) for k in range(lo, hi)]
        assert np.array_equal(z["U"], out["U"][own]) and np.array_equal(z["V"], out["V"][own])
        assert np.array_equal(z["PL"], np.concatenate([out["PL"][own], out["PL"][3 * kx:]]))
    assert covered == 32 * 31


def test_level_block_layout():
    """The level-block stacks of the sharded step (csrc/spdy_kernels.hpp: LevelShard; mirrored by sharding.block_slab): every
    (field, level) -- and every rank's level-free extra slab -- has exactly one slab, a rank's slabs are contiguous, and the
    owner formula the kernels use agrees with the block partition."""
    import speedy_f90_amd as s
    sh = s.sharding
    for kx in (5, 7, 8, 16):
        for w in range(1, kx + 1):
            for k in range(kx):
                lo, hi = sh.shard_range(kx, sh.level_owner(k, kx, w), w)
                assert lo <= k < hi
            for F, X in ((6, 0), (9, 1)):
                seen = {}
                for f in range(F):
                    for k in range(kx):
                        seen[sh.block_slab(f, k, kx, w, F, X)] = sh.level_owner(k, kx, w)
                for r in range(w):
                    lo, hi = sh.shard_range(kx, r, w)
                    mine = sorted(i for i, o in seen.items() if o == r)
                    assert mine == list(range(F * lo + X * r, F * hi + X * r)), (kx, w, F, r)
                assert len(seen) == F * kx


def test_shard_ranges_partition():
    import speedy_f90_amd as s
    for n in (0, 1, 7, 8, 48, 73, 91, 6144):
        for w in (1, 2, 4, 8):
            rs = [s.sharding.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs[:-1], rs[1:]))
            sizes = s.sharding.shard_sizes(n, w)
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n
