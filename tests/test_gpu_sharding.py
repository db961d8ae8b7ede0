"""GPU: the level-sharded implicit solve on the HIP path with RCCL (BASELINE.json config 3) as far as a 1-GPU box
allows: torch.distributed `nccl` backend at world_size 1, the C-ABI communicator (spdy_comm_*, direct RCCL) and the
torch.distributed gather, both on a NON-default torch stream (the ordering hazard of round 1), against the oracle."""
import os
import socket

import numpy as np
import pytest

import synth
from conftest import TOL

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def nccl_world1():
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("use_comm", [True, False])
def test_sharded_implicit_terms_hip(use_comm, nccl_world1, oracle_factory):
    import torch
    import speedy_f90_amd as s
    o = oracle_factory("t30")
    sp = s.Spectral("t30", kx=8, max_batch=16, device=0)
    sp.initialize_implicit(4800.0); o.tail_init(4800.0)
    d, t, p = synth.tail_inputs(8, sp.nx, sp.mx)
    rd, rt, rp = o.implicit_terms(d, t, p)
    comm = s.sharding.LevelComm(sp) if use_comm else None
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                          # everything below is ordered on `side`, not the default stream
        dd, dt_, dp = (torch.from_numpy(x).cuda(non_blocking=True) for x in (d, t, p))
        lo, hi = s.sharding.shard_range(sp.kx, 0, 1)
        for _ in range(3):                                 # repeated: a race would show as a stale/partial result
            gd, gt, gp = s.sharding.sharded_implicit_terms(sp, dd[lo:hi], dt_[lo:hi], dp, comm=comm)
            hd, ht, hp = gd.cpu(), gt.cpu(), gp.cpu()      # D2H on `side` right behind the kernel
        side.synchronize()
    for a, b in ((hd, rd), (ht, rt), (hp, rp)):
        assert synth.relerr(a.numpy(), b) <= TOL
    if comm:
        # the in-place gather primitive itself (at world 1 it returns before any RCCL call; the collectives are exercised by
        # test_collectives_really_issued_at_world1 and, on multi-GPU boxes, test_multi_rank_sharded_implicit)
        full = torch.from_numpy(d).cuda()
        comm.allgather_levels_(full)
        torch.cuda.synchronize()
        assert np.array_equal(full.cpu().numpy(), d)
        assert comm.level_range(8) == (0, 8)
        comm.close()
    sp.close()


def test_collectives_really_issued_at_world1(nccl_world1, oracle_factory, monkeypatch):
    """With one rank spdy_allgather_levels_dev returns early; SPDY_COMM_FORCE makes it issue the RCCL calls anyway -- 1: the
    in-place ncclAllGather route, 2: the per-rank ncclBroadcast route of ragged level blocks -- so both code paths (group
    start/end, in-place pointers, the plan's stream) execute on this 1-GPU box, eagerly and inside a captured graph.  A true
    multi-rank run is test_two_rank_sharded_implicit below (needs >= 2 GPUs)."""
    import torch
    import speedy_f90_amd as s
    o = oracle_factory("t30")
    o.tail_init(2400.0)
    d, t, p = synth.tail_inputs(8, o.nx, o.mx)
    rd, rt, rp = o.implicit_terms(d, t, p)
    for force in ("1", "2"):
        monkeypatch.setenv("SPDY_COMM_FORCE", force)
        sp = s.Spectral("t30", kx=8, max_batch=16, device=0)
        sp.initialize_implicit(2400.0)
        sp.use_own_stream()
        comm = s.sharding.LevelComm(sp)
        dd, dt_, dp = (torch.from_numpy(x).cuda() for x in (d, t, p))
        torch.cuda.synchronize()
        comm.allgather_levels_(dd, dt_)                      # in place: must leave the data untouched at one rank
        sp.synchronize()
        assert np.array_equal(dd.cpu().numpy(), d) and np.array_equal(dt_.cpu().numpy(), t)
        with sp.graph_capture() as g:                        # the collective + the solve as graph nodes
            comm.implicit_terms_sharded_(dd, dt_, dp)
        g.launch(); sp.synchronize()
        for a, b in ((dd, rd), (dt_, rt), (dp, rp)):
            assert synth.relerr(a.cpu().numpy(), b) <= TOL
        # teardown in the "wrong" order: the plan first, then the communicator and the graph (use-after-free)
        sp.close()
        from speedy_f90_amd._lib import SpdyError
        with pytest.raises(SpdyError):
            comm.allgather_levels_(dd, dt_)                  # dead communicator: SPDY_ERR_STATE, not a crash
        comm.close(); g.close()


def _two_rank_worker(rank, world, port, kx, q):
    import torch
    import torch.distributed as dist
    import speedy_f90_amd as s
    from oracle.pyoracle import Oracle, RESOLUTIONS
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        o = Oracle(RESOLUTIONS["t30"][0], RESOLUTIONS["t30"][1], RESOLUTIONS["t30"][2], kx)
        o.tail_init(2400.0)
        d, t, p = synth.tail_inputs(kx, o.nx, o.mx)
        rd, rt, rp = o.implicit_terms(d, t, p)
        sp = s.Spectral("t30", kx=kx, max_batch=16, device=rank)
        sp.initialize_implicit(2400.0)
        comm = s.sharding.LevelComm(sp)
        lo, hi = comm.level_range(kx)
        dd, dt_ = torch.zeros((kx, o.nx, o.mx), dtype=torch.complex128, device="cuda"), torch.zeros((kx, o.nx, o.mx), dtype=torch.complex128, device="cuda")
        dd[lo:hi] = torch.from_numpy(d[lo:hi]).cuda(); dt_[lo:hi] = torch.from_numpy(t[lo:hi]).cuda()      # own level block only
        dp = torch.from_numpy(p).cuda()
        comm.implicit_terms_sharded_(dd, dt_, dp)
        torch.cuda.synchronize()
        err = max(synth.relerr(a.cpu().numpy(), b) for a, b in ((dd, rd), (dt_, rt), (dp, rp)))
        comm.close(); sp.close()
        q.put((rank, err))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kx", [(2, 8), (3, 8), (4, 8), (8, 8)])
def test_multi_rank_sharded_implicit(world, kx):
    """BASELINE config 3 on real ranks: each rank fills its own level block, spdy_implicit_terms_sharded_dev completes the
    stacks over RCCL (world 2, 4, 8: in-place ncclAllGather -- 8 is config 3's own rank count, one level per rank; world 3 with 8
    levels: ragged blocks -> grouped ncclBroadcast) and
    every rank must hold the oracle's result.  Skipped on boxes with fewer GPUs."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs, %d visible" % (world, torch.cuda.device_count()))
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, kx, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0
    got = dict(q.get(timeout=10) for _ in range(world))
    assert sorted(got) == list(range(world)) and max(got.values()) <= TOL


def test_dev_calls_follow_torch_stream(oracle_factory):
    """`*_dev` methods run on torch's CURRENT stream (a side stream here), so torch ops before and after them need no
    extra synchronisation."""
    import torch
    import speedy_f90_amd as s
    o = oracle_factory("t30")
    sp = s.Spectral("t30", kx=8, max_batch=64, device=0)
    G = synth.grids(48, sp.ix, sp.il, first=123)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(5):
            dG = torch.from_numpy(G).cuda(non_blocking=True) * 2.0        # producer on `side`
            dS = torch.empty((48, sp.nx, sp.mx), dtype=torch.complex128, device="cuda")
            sp.grid_to_spec_dev(dG, dS)
            hS = (dS * 0.5).cpu()                                          # consumer on `side`
    side.synchronize()
    for b in (0, 17, 47):
        assert synth.relerr(hS[b].numpy(), o.grid_to_spec(G[b])) <= TOL
    sp.close()
