"""GPU: the COMPLETE level-sharded adiabatic time step (BASELINE.json config 3; include/spdy.h: spdy_sharded_step_dev).

Every rank holds the full prognostic state, transforms only its own levels and receives every other level of the two
intermediate stacks (gridded prognostics, spectral tendencies) through ONE all-gather each; the two column kernels
(get_grid_point_tendencies; tendency combination + get_spectral_tendencies + implicit_terms + diffusion + leapfrog) then run on
full columns.  Reference lines that couple levels: tendencies.f90:109-197, :256-285, geopotential.f90:33-57,
implicit.f90:174-216.

What runs where:
  * ranks INSIDE this process (spdy_comm_create_local: one thread + one plan per rank, all on device 0) -- real multi-rank
    runs of the very entry point an RCCL rank calls, on a 1-GPU box: world 1, 2 (equal blocks), 3 (ragged blocks), T30 L8,
    T30 L5 and T63 L16 (the raw-pairs route of the T63 direct batch); world 8 -- config 3's rank count: one level per rank at
    T30 L8, two at T63 L16.  The exchanged stacks are NaN-poisoned between steps:
    a level that a rank neither computed nor received would poison its result.
  * RCCL at world size 1 with SPDY_COMM_FORCE (the collectives are really issued, both routes), eager and captured into a graph.
  * RCCL ranks in separate processes (world 2 and 3) where the box has the GPUs (skipped otherwise).
Checked against the oracle's call-by-call step (1e-12) AND against the unsharded device step (bit for bit: the transforms are
position-independent and the column kernels evaluate the same expressions)."""
import os
import socket
import threading

import numpy as np
import pytest

import synth
from conftest import TOL, VARIANTS
from dynstep import ROB, SDRAG, WIL, oracle_dynamics_step, state, wave_relerr

pytestmark = pytest.mark.gpu
DT = 2400.0
PROGS = ("vor", "div", "t", "tr", "ps")


def make_plan(tag, device=0):
    import speedy_f90_amd as s
    trunc, ix, iy, kx = VARIANTS[tag]
    sp = s.Spectral((trunc, ix, iy), kx=kx, max_batch=4 * kx + 4, device=device)
    if tag in synth.SIGMA_SETS:
        sp.set_sigma(synth.SIGMA_SETS[tag])
    sp.initialize_implicit(DT)
    return sp


def unsharded_device_steps(sp, st, nsteps):
    """The three-call device step (tests/test_gpu_step.py) on one plan: prognostics, tendencies and direct-batch operands."""
    import torch
    kx, nx, mx, il, ix = sp.kx, sp.nx, sp.mx, sp.il, sp.ix
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    D = {n: dev(st[n]) for n in st}
    P = 3 * kx
    c128 = lambda *shape: torch.zeros(shape, dtype=torch.complex128, device="cuda")
    f64 = lambda *shape: torch.zeros(shape, dtype=torch.float64, device="cuda")
    ug, vg, plain_g, px, py = f64(kx, il, ix), f64(kx, il, ix), f64(4 * kx, il, ix), f64(1, il, ix), f64(1, il, ix)
    U, V, PL = f64(P, il, ix), f64(P, il, ix), f64(P + 1, il, ix)
    pvor, pdiv, pspec, phi = c128(P, nx, mx), c128(P, nx, mx), c128(P + 1, nx, mx), c128(kx, nx, mx)
    for _ in range(nsteps):
        sp.inverse_batch_segs_dev(D["vor"][1], D["div"][1], ug, vg, [D[n][1] for n in ("vor", "div", "t", "tr")], plain_g, D["ps"][1:2], px, py,
                                  kcos_pairs=2, kcos=1)
        sp.grid_tendencies_dev(ug, vg, plain_g[2 * kx:3 * kx], plain_g[:kx], plain_g[kx:2 * kx], plain_g[3 * kx:], px, py, U, V, PL)
        sp.direct_batch_spectral_step_dev(U, V, PL, pvor, pdiv, pspec, D["vor"], D["div"], D["t"], D["tr"], D["ps"], D["phis"], D["tcorh"],
                                          D["qcorh"], SDRAG, 2, DT, ROB, WIL, phi, kcos=2)
    torch.cuda.synchronize()
    out = {n: D[n].cpu().numpy() for n in PROGS}
    out["phi"] = phi.cpu().numpy()
    out["tend"] = np.concatenate([pvor[:kx].cpu().numpy(), pdiv.cpu().numpy(), pspec[P:].cpu().numpy()])   # vordt | divdt | tdt | trdt | psdt
    out["U"], out["V"], out["PL"] = U.cpu().numpy(), V.cpu().numpy(), PL.cpu().numpy()
    return out


def _rank_thread(rank, group, tag, st, nsteps, results, errors, device=0, transpose=False):
    import torch
    import speedy_f90_amd as s
    try:
        torch.cuda.set_device(device)          # (per thread: the tensors below live on the rank's device)
        sp = make_plan(tag, device)
        sp.use_own_stream()
        comm = s.sharding.LevelComm(sp, group=group, rank=rank)
        if transpose:
            comm.set_option("transpose", 1)
        comm.sharded_step_workspace()
        kx, nx, mx = sp.kx, sp.nx, sp.mx
        D = {n: torch.from_numpy(np.ascontiguousarray(st[n])).cuda() for n in st}
        phi = torch.zeros((kx, nx, mx), dtype=torch.complex128, device="cuda")
        tend = torch.zeros((4 * kx + 1, nx, mx), dtype=torch.complex128, device="cuda")
        G, T = comm.sharded_step_stacks()
        torch.cuda.synchronize()
        for step in range(nsteps):
            G.fill_(float("nan")); T.fill_(float("nan"))          # whatever a rank does not compute must ARRIVE, or it poisons the step
            torch.cuda.synchronize()
            if step == 0:                                         # the one call ...
                comm.sharded_step_(D["vor"], D["div"], D["t"], D["tr"], D["ps"], D["phis"], D["tcorh"], D["qcorh"], SDRAG, 2, 2, DT, ROB, WIL,
                                   phi, tend)
            else:                                                 # ... and its two halves (the physics hook sits between them)
                comm.sharded_step_grid_(D["vor"], D["div"], D["t"], D["tr"], D["ps"], 2)
                comm.sharded_step_spectral_(D["vor"], D["div"], D["t"], D["tr"], D["ps"], D["phis"], D["tcorh"], D["qcorh"], SDRAG, 2, DT, ROB,
                                            WIL, phi, tend)
            sp.synchronize()
        U, V, PL, lo, hi = comm.sharded_step_operands()
        out = {}
        if transpose:
            # what the rank holds BEFORE the gather: its coefficient range of every level is current (both time levels)
            desc = comm.describe()
            e0, e1 = desc["coefficients"]
            out["own_range"] = (e0, e1)
            out["before"] = {n: D[n].reshape(D[n].shape[:-2] + (-1,))[..., e0:e1].cpu().numpy() for n in PROGS}
            out["describe"] = desc
            comm.state_gather_(D["vor"], D["div"], D["t"], D["tr"], D["ps"])
            comm.gather_ranges_(phi, tend)
            sp.synchronize()
        out.update({n: D[n].cpu().numpy() for n in PROGS})
        out.update(phi=phi.cpu().numpy(), tend=tend.cpu().numpy(), U=U.cpu().numpy(), V=V.cpu().numpy(), PL=PL.cpu().numpy(), lo=lo, hi=hi)
        results[rank] = out
        comm.close(); sp.close()
    except Exception as e:          # a missing rank would leave the others waiting for the group's timeout
        errors[rank] = e


@pytest.mark.parametrize("tag,world", [(t, w) for t in ("t30", "t30k5", "t63k16") for w in (1, 2, 3)] + [("t30", 8), ("t63k16", 8)])
def test_sharded_step_in_process_ranks(tag, world, oracle_factory, monkeypatch):
    import speedy_f90_amd as s
    monkeypatch.setenv("SPDY_COMM_TIMEOUT_S", "60")
    kx = VARIANTS[tag][3]
    o = oracle_factory(tag)
    o.tail_init(DT)
    sp0 = make_plan(tag)
    st = state(sp0, 8000)
    nsteps = 2
    whole = unsharded_device_steps(sp0, st, nsteps)
    group = s.sharding.LocalGroup(sp0.lib, world)
    results, errors = {}, {}
    threads = [threading.Thread(target=_rank_thread, args=(r, group, tag, st, nsteps, results, errors)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
    assert sorted(results) == list(range(world))
    group.close(); sp0.close()
    ref = st
    for _ in range(nsteps):
        ref, out = oracle_dynamics_step(o, ref, 2, DT, ROB)
    ref_tend = np.concatenate([out["vordt"], out["divdt"], out["tdt"], out["trdt"], out["psdt"][None]])
    worst = 0.0
    for r in range(world):
        got = results[r]
        assert (got["lo"], got["hi"]) == s.sharding.shard_range(kx, r, world)
        own = [g * kx + k for g in range(3) for k in range(got["lo"], got["hi"])]
        # vs the oracle's call-by-call step: the north star's bar
        for n in PROGS:
            worst = max(worst, synth.relerr(got[n], ref[n]), wave_relerr(got[n], ref[n]))
        worst = max(worst, wave_relerr(got["phi"], out["phi"]))
        for a in range(4):
            worst = max(worst, wave_relerr(got["tend"][a * kx:(a + 1) * kx], ref_tend[a * kx:(a + 1) * kx]))
        worst = max(worst, synth.relerr(got["tend"][4 * kx], ref_tend[4 * kx]))
        worst = max(worst, synth.relerr(got["U"], out["U"][own]), synth.relerr(got["V"], out["V"][own]),
                    synth.relerr(got["PL"], np.concatenate([out["PL"][own], out["PL"][3 * kx:]])))
        # vs the unsharded device step: same bits, whatever the rank count
        for n in PROGS + ("phi", "tend"):
            assert np.array_equal(got[n], whole[n]), (tag, world, r, n, synth.relerr(got[n], whole[n]))
        assert np.array_equal(got["U"], whole["U"][own]) and np.array_equal(got["V"], whole["V"][own])
        assert np.array_equal(got["PL"], np.concatenate([whole["PL"][own], whole["PL"][3 * kx:]]))
    print("\n[sharded step %s, %d in-process ranks] worst relative error vs the oracle %.1e; bits equal to the unsharded device step"
          % (tag, world, worst))
    assert worst <= TOL, (tag, world, worst)


@pytest.mark.parametrize("tag,world", [(t, w) for t in ("t30", "t30k5", "t63k16") for w in (1, 2, 3)] + [("t30", 8), ("t63k16", 8)])
def test_sharded_step_transposed_in_process_ranks(tag, world, oracle_factory, monkeypatch):
    """The TRANSPOSED form of the level-sharded step (spdy_comm_set_option "transpose"; include/spdy.h): levels <-> point ranges
    around the grid-space column kernel, levels <-> coefficient ranges around the spectral step (tendencies.f90:109-197, 242-293,
    implicit.f90:168-217 are independent in the horizontal), four exchanges instead of two all-gathers, nothing replicated.
    Real multi-rank runs with in-process ranks on one GPU, two chained steps (the second one starts with exchange 4 in the
    caller's arrays), then the state gather.  Against the oracle's call-by-call step at 1e-12; against the unsharded device step
    BIT FOR BIT at T30 (same kernels' expressions on the same values, whatever the rank count) and to rounding at T63 L16 (there
    the unsharded step applies vds inside the spectral step, the transposed form before its exchange)."""
    import speedy_f90_amd as s
    monkeypatch.setenv("SPDY_COMM_TIMEOUT_S", "60")
    kx = VARIANTS[tag][3]
    o = oracle_factory(tag)
    o.tail_init(DT)
    sp0 = make_plan(tag)
    st = state(sp0, 8000)
    nsteps = 2
    whole = unsharded_device_steps(sp0, st, nsteps)
    group = s.sharding.LocalGroup(sp0.lib, world)
    results, errors = {}, {}
    threads = [threading.Thread(target=_rank_thread, args=(r, group, tag, st, nsteps, results, errors, 0, True)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
    assert sorted(results) == list(range(world))
    group.close(); sp0.close()
    ref = st
    for _ in range(nsteps):
        ref, out = oracle_dynamics_step(o, ref, 2, DT, ROB)
    ref_tend = np.concatenate([out["vordt"], out["divdt"], out["tdt"], out["trdt"], out["psdt"][None]])
    exact = tag.startswith("t30")
    worst, covered = 0.0, 0
    for r in range(world):
        got = results[r]
        own = [g * kx + k for g in range(3) for k in range(got["lo"], got["hi"])]
        for n in PROGS:
            worst = max(worst, synth.relerr(got[n], ref[n]), wave_relerr(got[n], ref[n]))
        worst = max(worst, wave_relerr(got["phi"], out["phi"]))
        for a in range(4):
            worst = max(worst, wave_relerr(got["tend"][a * kx:(a + 1) * kx], ref_tend[a * kx:(a + 1) * kx]))
        worst = max(worst, synth.relerr(got["tend"][4 * kx], ref_tend[4 * kx]))
        worst = max(worst, synth.relerr(got["U"], out["U"][own]), synth.relerr(got["V"], out["V"][own]),
                    synth.relerr(got["PL"], np.concatenate([out["PL"][own], out["PL"][3 * kx:]])))
        # the ranks' coefficient ranges tile the spectrum, and each rank's range was current before the gather
        e0, e1 = got["own_range"]
        covered += e1 - e0
        for n in PROGS:
            flat = whole[n].reshape(whole[n].shape[:-2] + (-1,))[..., e0:e1]
            if exact:
                assert np.array_equal(got["before"][n], flat), (tag, world, r, n)
            else:
                assert synth.relerr(got["before"][n], flat) <= 1e-13, (tag, world, r, n)
        for n in PROGS + ("phi", "tend"):
            if exact:
                assert np.array_equal(got[n], whole[n]), (tag, world, r, n, synth.relerr(got[n], whole[n]))
            else:
                assert synth.relerr(got[n], whole[n]) <= 1e-13, (tag, world, r, n, synth.relerr(got[n], whole[n]))
        # the direct-batch operands of the rank's levels came home complete (exchange 2)
        wpl = np.concatenate([whole["PL"][own], whole["PL"][3 * kx:]])
        if exact:
            assert np.array_equal(got["U"], whole["U"][own]) and np.array_equal(got["V"], whole["V"][own]) and np.array_equal(got["PL"], wpl)
        else:   # (second step: its inputs already differ by the first step's rounding)
            assert max(synth.relerr(got["U"], whole["U"][own]), synth.relerr(got["V"], whole["V"][own]), synth.relerr(got["PL"], wpl)) <= 1e-13
        d = got["describe"]
        assert d["form"] == "transpose" and d["nranks"] == world and d["rank"] == r
        if world >= 4:      # (at two ranks the four exchanges move about what the two all-gathers do; the form pays from four ranks on)
            assert d["bytes_received_per_step_transposed_form"] < 0.6 * d["bytes_received_per_step_allgather_form"], d
    assert covered == results[0]["vor"].shape[-1] * results[0]["vor"].shape[-2]
    print("\n[transposed sharded step %s, %d in-process ranks] worst relative error vs the oracle %.1e; %s the unsharded device step"
          % (tag, world, worst, "bits equal to" if exact else "within 1e-13 of"))
    assert worst <= TOL, (tag, world, worst)


@pytest.mark.parametrize("transpose", [False, True])
def test_sharded_step_in_process_ranks_on_several_devices(transpose, oracle_factory, monkeypatch):
    """The in-process group with one DEVICE per rank -- what a single-process host driving several GPUs runs: the exchanges are
    peer copies (hipMemcpyPeerAsync) between the ranks' streams.  Skipped on 1-GPU boxes."""
    import torch
    import speedy_f90_amd as s
    world = min(torch.cuda.device_count(), 4)
    if world < 2:
        pytest.skip("needs >= 2 GPUs, %d visible" % torch.cuda.device_count())
    monkeypatch.setenv("SPDY_COMM_TIMEOUT_S", "60")
    tag, nsteps = "t30", 2
    o = oracle_factory(tag)
    o.tail_init(DT)
    sp0 = make_plan(tag)
    st = state(sp0, 8000)
    whole = unsharded_device_steps(sp0, st, nsteps)
    group = s.sharding.LocalGroup(sp0.lib, world)
    results, errors = {}, {}
    threads = [threading.Thread(target=_rank_thread, args=(r, group, tag, st, nsteps, results, errors, r, transpose)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
    group.close(); sp0.close()
    for r in range(world):
        for n in PROGS + ("phi", "tend"):
            assert np.array_equal(results[r][n], whole[n]), (r, n)


def _physics_increment(kx, il, ix):
    """A seeded stand-in for get_physical_tendencies (tendencies.f90:203-206 adds to utend, vtend, ttend, trtend in grid space)."""
    rng = np.random.default_rng(77)
    return [rng.uniform(-1.0, 1.0, (kx, il, ix)) * sc for sc in (1e-5, 1e-5, 1e-4, 1e-8)]


def _hook_thread(rank, group, st, results, errors):
    import torch
    import speedy_f90_amd as s
    try:
        sp = make_plan("t30")
        sp.use_own_stream()
        comm = s.sharding.LevelComm(sp, group=group, rank=rank)
        kx = sp.kx
        D = {n: torch.from_numpy(np.ascontiguousarray(st[n])).cuda() for n in st}
        phi = torch.zeros((kx, sp.nx, sp.mx), dtype=torch.complex128, device="cuda")
        inc = [torch.from_numpy(a).cuda() for a in _physics_increment(kx, sp.il, sp.ix)]
        torch.cuda.synchronize()
        comm.sharded_step_grid_(D["vor"], D["div"], D["t"], D["tr"], D["ps"], 2)
        sp.synchronize()
        U, V, PL, lo, hi = comm.sharded_step_operands()          # this rank's levels only: [utend | ..], [vtend | ..], [KE | ttend | trtend | ps]
        nl = hi - lo
        U[:nl] += inc[0][lo:hi]; V[:nl] += inc[1][lo:hi]; PL[nl:2 * nl] += inc[2][lo:hi]; PL[2 * nl:3 * nl] += inc[3][lo:hi]
        torch.cuda.synchronize()
        comm.sharded_step_spectral_(D["vor"], D["div"], D["t"], D["tr"], D["ps"], D["phis"], D["tcorh"], D["qcorh"], SDRAG, 2, DT, ROB, WIL, phi)
        sp.synchronize()
        results[rank] = {n: D[n].cpu().numpy() for n in PROGS}
        comm.close(); sp.close()
    except Exception as e:
        errors[rank] = e


def test_sharded_step_physics_hook(monkeypatch):
    """The two halves of the sharded step with a host "physics" in between (the hook of tendencies.f90:203-206): every rank adds
    its grid-space increments to ITS levels' operands (spdy_sharded_step_operands); the result must equal, bit for bit, the
    unsharded three-call step with the same increments added to the full operands."""
    import torch
    import speedy_f90_amd as s
    monkeypatch.setenv("SPDY_COMM_TIMEOUT_S", "60")
    sp = make_plan("t30")
    kx, nx, mx, il, ix = sp.kx, sp.nx, sp.mx, sp.il, sp.ix
    st = state(sp, 8000)
    D = {n: torch.from_numpy(np.ascontiguousarray(st[n])).cuda() for n in st}
    P = 3 * kx
    c128 = lambda *shape: torch.zeros(shape, dtype=torch.complex128, device="cuda")
    f64 = lambda *shape: torch.zeros(shape, dtype=torch.float64, device="cuda")
    ug, vg, plain_g, px, py = f64(kx, il, ix), f64(kx, il, ix), f64(4 * kx, il, ix), f64(1, il, ix), f64(1, il, ix)
    U, V, PL = f64(P, il, ix), f64(P, il, ix), f64(P + 1, il, ix)
    pvor, pdiv, pspec, phi = c128(P, nx, mx), c128(P, nx, mx), c128(P + 1, nx, mx), c128(kx, nx, mx)
    inc = [torch.from_numpy(a).cuda() for a in _physics_increment(kx, il, ix)]
    sp.inverse_batch_segs_dev(D["vor"][1], D["div"][1], ug, vg, [D[n][1] for n in ("vor", "div", "t", "tr")], plain_g, D["ps"][1:2], px, py,
                              kcos_pairs=2, kcos=1)
    sp.grid_tendencies_dev(ug, vg, plain_g[2 * kx:3 * kx], plain_g[:kx], plain_g[kx:2 * kx], plain_g[3 * kx:], px, py, U, V, PL)
    U[:kx] += inc[0]; V[:kx] += inc[1]; PL[kx:2 * kx] += inc[2]; PL[2 * kx:3 * kx] += inc[3]
    sp.direct_batch_spectral_step_dev(U, V, PL, pvor, pdiv, pspec, D["vor"], D["div"], D["t"], D["tr"], D["ps"], D["phis"], D["tcorh"], D["qcorh"],
                                      SDRAG, 2, DT, ROB, WIL, phi, kcos=2)
    torch.cuda.synchronize()
    whole = {n: D[n].cpu().numpy() for n in PROGS}
    group = s.sharding.LocalGroup(sp.lib, 2)
    results, errors = {}, {}
    threads = [threading.Thread(target=_hook_thread, args=(r, group, st, results, errors)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    group.close(); sp.close()
    for r in range(2):
        for n in PROGS:
            assert np.array_equal(results[r][n], whole[n]), (r, n, synth.relerr(results[r][n], whole[n]))
            assert not np.array_equal(whole[n], st[n])


def test_in_process_group_errors():
    """A rank that never shows up breaks the group after the timeout (SPDY_ERR_COMM) instead of hanging its peers; collectives of
    an in-process communicator are refused inside a graph capture; a group cannot be destroyed under its communicators."""
    import torch
    import speedy_f90_amd as s
    from speedy_f90_amd._lib import SpdyError
    os.environ["SPDY_COMM_TIMEOUT_S"] = "2"
    try:
        sp = make_plan("t30")
        sp.use_own_stream()
        group = s.sharding.LocalGroup(sp.lib, 2)
        comm = s.sharding.LevelComm(sp, group=group, rank=0)
        with pytest.raises(SpdyError):
            group.close()
        full = torch.zeros((8, sp.nx, sp.mx), dtype=torch.complex128, device="cuda")
        torch.cuda.synchronize()
        with pytest.raises(SpdyError, match="ranks arrived"):
            comm.allgather_levels_(full)
        with pytest.raises(SpdyError):                 # the group stays broken
            comm.allgather_levels_(full)
        comm.close(); group.close()
        group = s.sharding.LocalGroup(sp.lib, 1)
        comm = s.sharding.LevelComm(sp, group=group, rank=0)
        with pytest.raises(SpdyError, match="capture"):
            with sp.graph_capture():
                comm.allgather_levels_(full)
        comm.close(); group.close(); sp.close()
    finally:
        os.environ.pop("SPDY_COMM_TIMEOUT_S", None)


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.fixture(scope="module")
def nccl_world1():
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("tag", ["t30", "t63k16"])
def test_sharded_step_rccl_graph_world1(tag, nccl_world1, oracle_factory, monkeypatch):
    """The sharded step over RCCL as ONE captured graph per rank, at world size 1 with SPDY_COMM_FORCE: 1 = the in-place
    ncclAllGather route, 2 = the per-rank ncclBroadcast route of ragged blocks -- both exchanges really issued as graph nodes
    between the transform launches and the column kernels.  Two replays against the oracle and the unsharded device step."""
    import torch
    import speedy_f90_amd as s
    kx = VARIANTS[tag][3]
    o = oracle_factory(tag)
    o.tail_init(DT)
    sp0 = make_plan(tag)
    st = state(sp0, 8000)
    whole = unsharded_device_steps(sp0, st, 2)
    sp0.close()
    ref = st
    for _ in range(2):
        ref, out = oracle_dynamics_step(o, ref, 2, DT, ROB)
    for force in ("1", "2"):
        monkeypatch.setenv("SPDY_COMM_FORCE", force)
        sp = make_plan(tag)
        sp.use_own_stream()
        comm = s.sharding.LevelComm(sp)
        comm.sharded_step_workspace()
        D = {n: torch.from_numpy(np.ascontiguousarray(st[n])).cuda() for n in st}
        phi = torch.zeros((kx, sp.nx, sp.mx), dtype=torch.complex128, device="cuda")
        tend = torch.zeros((4 * kx + 1, sp.nx, sp.mx), dtype=torch.complex128, device="cuda")
        torch.cuda.synchronize()
        with sp.graph_capture() as g:
            comm.sharded_step_(D["vor"], D["div"], D["t"], D["tr"], D["ps"], D["phis"], D["tcorh"], D["qcorh"], SDRAG, 2, 2, DT, ROB, WIL, phi, tend)
        for _ in range(2):
            g.launch()
        sp.synchronize()
        for n in PROGS:
            got = D[n].cpu().numpy()
            assert max(synth.relerr(got, ref[n]), wave_relerr(got, ref[n])) <= TOL, (force, n)
            assert np.array_equal(got, whole[n]), (force, n)
        assert np.array_equal(tend.cpu().numpy(), whole["tend"]) and np.array_equal(phi.cpu().numpy(), whole["phi"])
        g.close(); comm.close(); sp.close()


@pytest.mark.parametrize("tag", ["t30", "t63k16"])
def test_sharded_step_transposed_rccl_world1_in_graph(tag, oracle_factory, monkeypatch):
    """The transposed form over RCCL as ONE captured graph, at world size 1 with SPDY_COMM_FORCE=1: the rank's own piece of each
    of the four exchanges really travels through grouped ncclSend / ncclRecv (and the state gather through ncclBroadcast), packed
    and unpacked through the staging buffer by 2-D copies -- every node of the multi-rank route on a 1-GPU box.  Two replays
    (the second starts with exchange 4 in the caller's arrays), then the gather, against the oracle and the unsharded device step."""
    import torch
    import speedy_f90_amd as s
    kx = VARIANTS[tag][3]
    o = oracle_factory(tag)
    o.tail_init(DT)
    sp0 = make_plan(tag)
    st = state(sp0, 8000)
    whole = unsharded_device_steps(sp0, st, 2)
    sp0.close()
    ref = st
    for _ in range(2):
        ref, out = oracle_dynamics_step(o, ref, 2, DT, ROB)
    monkeypatch.setenv("SPDY_COMM_FORCE", "1")
    monkeypatch.setenv("SPDY_SHARD_TRANSPOSE", "1")
    sp = make_plan(tag)
    sp.use_own_stream()
    comm = s.sharding.LevelComm(sp)
    comm.sharded_step_workspace()
    d = comm.describe()
    assert d["route"] == "rccl" and d["form"] == "transpose" and "rccl" in d["librccl"], d
    D = {n: torch.from_numpy(np.ascontiguousarray(st[n])).cuda() for n in st}
    phi = torch.zeros((kx, sp.nx, sp.mx), dtype=torch.complex128, device="cuda")
    tend = torch.zeros((4 * kx + 1, sp.nx, sp.mx), dtype=torch.complex128, device="cuda")
    torch.cuda.synchronize()
    with sp.graph_capture() as g:
        comm.sharded_step_(D["vor"], D["div"], D["t"], D["tr"], D["ps"], D["phis"], D["tcorh"], D["qcorh"], SDRAG, 2, 2, DT, ROB, WIL, phi, tend)
    nodes = g.num_nodes()
    for _ in range(2):
        g.launch()
    sp.synchronize()
    comm.state_gather_(D["vor"], D["div"], D["t"], D["tr"], D["ps"])
    comm.gather_ranges_(phi, tend)
    sp.synchronize()
    exact = tag.startswith("t30")
    for n in PROGS:
        got = D[n].cpu().numpy()
        assert max(synth.relerr(got, ref[n]), wave_relerr(got, ref[n])) <= TOL, n
        assert np.array_equal(got, whole[n]) if exact else synth.relerr(got, whole[n]) <= 1e-13, n
    if exact:
        assert np.array_equal(tend.cpu().numpy(), whole["tend"]) and np.array_equal(phi.cpu().numpy(), whole["phi"])
    print("\n[transposed step over RCCL, world 1 forced, %s] %d graph nodes; librccl: %s" % (tag, nodes, d["librccl"]))
    g.close(); comm.close(); sp.close()


def _rccl_rank(rank, world, port, tag, q, transpose=False):
    import torch
    import torch.distributed as dist
    import speedy_f90_amd as s
    from oracle.pyoracle import Oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        kx = VARIANTS[tag][3]
        o = Oracle(*VARIANTS[tag])
        if tag in synth.SIGMA_SETS:
            o.set_sigma(synth.SIGMA_SETS[tag])
        o.tail_init(DT)
        sp = make_plan(tag, device=rank)
        sp.use_own_stream()
        comm = s.sharding.LevelComm(sp)
        if transpose:
            comm.set_option("transpose", 1)
        comm.sharded_step_workspace()
        st = state(sp, 8000)
        D = {n: torch.from_numpy(np.ascontiguousarray(st[n])).cuda() for n in st}
        phi = torch.zeros((kx, sp.nx, sp.mx), dtype=torch.complex128, device="cuda")
        G, T = comm.sharded_step_stacks()
        G.fill_(float("nan")); T.fill_(float("nan"))
        torch.cuda.synchronize()
        with sp.graph_capture() as g:
            comm.sharded_step_(D["vor"], D["div"], D["t"], D["tr"], D["ps"], D["phis"], D["tcorh"], D["qcorh"], SDRAG, 2, 2, DT, ROB, WIL, phi)
        ref = st
        for _ in range(2):
            g.launch()
            ref, out = oracle_dynamics_step(o, ref, 2, DT, ROB)
        sp.synchronize()
        if transpose:
            comm.state_gather_(D["vor"], D["div"], D["t"], D["tr"], D["ps"])
            sp.synchronize()
        err = max(max(synth.relerr(D[n].cpu().numpy(), ref[n]), wave_relerr(D[n].cpu().numpy(), ref[n])) for n in PROGS)
        g.close(); comm.close(); sp.close()
        q.put((rank, err))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("world,tag", [(2, "t30"), (3, "t30"), (2, "t63k16"), (4, "t30"), (8, "t30"), (4, "t63k16"), (8, "t63k16")])
def test_sharded_step_rccl_ranks(world, tag, transpose):
    """BASELINE config 3 on real RCCL ranks (one process per GPU): every rank captures the complete sharded step -- its two
    all-gathers included -- into a graph, replays it twice from NaN-poisoned exchange stacks and must hold the oracle's
    prognostics.  world 3 with 8 levels: ragged blocks -> grouped ncclBroadcast; world 8 at T30 L8 is config 3's own rank
    count (one level per rank), world 8 at T63 L16 two levels per rank.  transpose: the transposed form (four grouped
    ncclSend / ncclRecv exchanges, then the state gather).  Skipped on boxes with fewer GPUs."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs, %d visible" % (world, torch.cuda.device_count()))
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_rank, args=(r, world, port, tag, q, transpose)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(600)
        assert pr.exitcode == 0
    got = dict(q.get(timeout=10) for _ in range(world))
    assert sorted(got) == list(range(world)) and max(got.values()) <= TOL, got
