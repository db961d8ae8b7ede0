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
        # the in-place gather primitive itself (trivial at world 1, but the RCCL path is loaded and executed)
        full = torch.from_numpy(d).cuda()
        comm.allgather_levels_(full)
        torch.cuda.synchronize()
        assert np.array_equal(full.cpu().numpy(), d)
        assert comm.level_range(8) == (0, 8)
        comm.close()
    sp.close()


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
