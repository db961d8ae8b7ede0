"""The Fortran host side: the drop-in modules `spectral`, `horizontal_diffusion`, `implicit`, `geopotential`
(speedy.f90_amd/fortran/) that keep the reference's module names, public names and signatures over the C ABI.
A flang-built driver calls them exactly as the model would; results are compared with the oracle at 1e-12."""
import os
import subprocess

import numpy as np
import pytest

import synth
from conftest import ROOT, TOL
from synth import tail_inputs

FDIR = os.path.join(ROOT, "speedy.f90_amd", "fortran")


def driver(tag):
    return os.path.join(FDIR, "build", tag, "dropin_driver")


def test_fortran_sources_keep_reference_api():
    """Same module name and public names as the reference's spectral module (spectral.f90:1-11)."""
    src = open(os.path.join(FDIR, "spectral.f90")).read().lower()
    assert "module spectral" in src
    for name in ("el2", "initialize_spectral", "laplacian", "inverse_laplacian", "spec_to_grid", "grid_to_spec",
                 "grad", "vds", "uvspec", "vdspec", "trunct"):
        assert name in src.split("contains")[0], name
    assert "function spec_to_grid(vorm, kcos) result(vorg)" in src
    assert "function grid_to_spec(vorg) result(vorm)" in src
    assert "bind(c" in open(os.path.join(FDIR, "spdy_c.f90")).read().lower()
    # the tail modules: reference module names and full public sets (horizontal_diffusion.f90:10-11, implicit.f90:10-11,
    # geopotential.f90:11)
    for mod, names in (("horizontal_diffusion", ("initialize_horizontal_diffusion", "do_horizontal_diffusion", "dmp", "dmpd", "dmps",
                                                 "dmp1", "dmp1d", "dmp1s", "tcorv", "qcorv", "tcorh", "qcorh")),
                       ("implicit", ("initialize_implicit", "implicit_terms", "tref", "tref2", "tref3")),
                       ("geopotential", ("initialize_geopotential", "get_geopotential"))):
        src = open(os.path.join(FDIR, mod + ".f90")).read().lower()
        assert "module " + mod in src
        head = src.split("contains")[0]
        pub = " ".join(l for l in head.split("\n") if l.strip().startswith("public"))
        for name in names:
            assert name in pub, (mod, name)
    # the caller side: time_stepping keeps first_step and step(j1, j2, dt) (time_stepping.f90:8, :35)
    src = open(os.path.join(FDIR, "time_stepping.f90")).read().lower()
    assert "module time_stepping" in src and "public first_step, step" in src
    assert "subroutine step(j1, j2, dt)" in src and "subroutine first_step" in src
    # spdy_c.f90 is one-to-one with include/spdy.h: every entry point has a bind(C) interface
    import re
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "spdy.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(spdy_[a-z0-9_]+)\s*\(", hdr))
    bound = set(re.findall(r'name="(spdy_[a-z0-9_]+)"', open(os.path.join(FDIR, "spdy_c.f90")).read()))
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))


def test_reference_callers_resolve_against_dropins():
    """Build container only: compile the drop-ins against the reference's own types/params and check its callers'
    `use ..., only:` lists (and two whole caller files) against them (fortran/check_reference_callers.sh)."""
    if not os.path.isdir("/root/reference/source") or not os.path.exists("/opt/rocm/lib/llvm/bin/flang"):
        pytest.skip("needs /root/reference and flang (build container)")
    r = subprocess.run([os.path.join(FDIR, "check_reference_callers.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "compile unchanged against the drop-in" in r.stdout


@pytest.mark.parametrize("tag", ["t30", "t63"])
def test_driver_fails_loudly_without_gpu(tag):
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    if not os.path.exists(driver(tag)):
        pytest.skip("Fortran driver not built (no flang)")
    r = subprocess.run([driver(tag), "/dev/null", "/dev/null"], capture_output=True, text=True)
    assert r.returncode != 0                          # no device -> error stop, never a silent CPU path
    r = subprocess.run([os.path.join(FDIR, "build", tag, "dropin_step"), "time", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "spdy" in (r.stdout + r.stderr).lower()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["t30", "t63"])
def test_dropin_module_vs_oracle(tag, tmp_path, oracle_factory):
    if not os.path.exists(driver(tag)):
        pytest.skip("Fortran driver not built (no flang on this box and no prebuilt binary)")
    o = oracle_factory(tag)
    nx, mx, il, ix, kx = o.nx, o.mx, o.il, o.ix, o.kx
    S = synth.spectra(2, o.trunc, first=40, full_rows=True)
    G = synth.grids(2, ix, il, first=40)
    sk, tk, ps = tail_inputs(kx, nx, mx)
    o.tail_init(4800.0)
    dmp, dmp1 = o.table("dmp").reshape(nx, mx), o.table("dmp1").reshape(nx, mx)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        for a in (S, G, sk, tk, ps, dmp, dmp1):
            f.write(np.ascontiguousarray(a).tobytes())
    r = subprocess.run([driver(tag), str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(fout, np.float64)
    pos = [0]

    def take(shape, cplx):
        n = int(np.prod(shape)) * (2 if cplx else 1)
        a = raw[pos[0]:pos[0] + n]
        pos[0] += n
        return a.view(np.complex128).reshape(shape) if cplx else a.reshape(shape)

    def ok(x, ref):
        assert synth.relerr(x, ref) <= TOL

    sp, gr = (nx, mx), (il, ix)
    assert np.array_equal(take(sp, False), o.table("el2").reshape(sp))
    ok(take(gr, False), o.spec_to_grid(S[0], 1))
    ok(take(gr, False), o.spec_to_grid(S[1], 2))
    ok(take(sp, True), o.grid_to_spec(G[0]))
    ok(take(sp, True), o.laplacian(S[0]))
    ok(take(sp, True), o.inverse_laplacian(S[0]))
    ok(take(sp, True), o.trunct(S[0]))
    rdx, rdy = o.grad(S[0]); ok(take(sp, True), rdx); ok(take(sp, True), rdy)
    ru, rv = o.uvspec(S[0], S[1]); ok(take(sp, True), ru); ok(take(sp, True), rv)
    a, b = o.vds(S[0], S[1]); ok(take(sp, True), a); ok(take(sp, True), b)
    a, b = o.vdspec(G[0], G[1], 2); ok(take(sp, True), a); ok(take(sp, True), b)
    kc = [1 + (k % 2) for k in range(1, kx + 1)]
    glev = take((kx,) + gr, False)
    for k in range(kx):
        ok(glev[k], o.spec_to_grid(sk[k], kc[k]))
    slev = take((kx,) + sp, True)
    for k in range(kx):
        ok(slev[k], o.grid_to_spec(glev[k]))
    # module tables filled by initialize_horizontal_diffusion / initialize_implicit (public in the reference)
    for name in ("dmp", "dmpd", "dmps", "dmp1", "dmp1d", "dmp1s"):
        assert np.array_equal(take(sp, False), o.table(name).reshape(sp)), name
    for name in ("tcorv", "qcorv", "tref", "tref2", "tref3"):
        assert np.array_equal(take((kx,), False), o.table(name)), name
    ok(take((kx,) + sp, True), o.hdiff(tk, sk, dmp, dmp1))
    ok(take(sp, True), o.hdiff(ps, sk[0], o.table("dmps").reshape(sp), o.table("dmp1s").reshape(sp)))
    ok(take((kx,) + sp, True), o.geopotential(tk, ps))
    rd, rt, rp = o.implicit_terms(sk, tk, ps)
    ok(take((kx,) + sp, True), rd); ok(take((kx,) + sp, True), rt); ok(take(sp, True), rp)
    # level-stack sequences in one call each; inputs: the *updated* sk, tk written back by implicit_terms above
    ug, vg = take((kx,) + gr, False), take((kx,) + gr, False)
    for k in range(kx):
        ru, rv = o.uvspec(rd[k], rt[k])
        ok(ug[k], o.spec_to_grid(ru, 2)); ok(vg[k], o.spec_to_grid(rv, 2))
    gx, gy = take((kx,) + gr, False), take((kx,) + gr, False)
    for k in range(kx):
        rdx, rdy = o.grad(rd[k])
        ok(gx[k], o.spec_to_grid(rdx, 2)); ok(gy[k], o.spec_to_grid(rdy, 2))
    vorl, divl = take((kx,) + sp, True), take((kx,) + sp, True)
    for k in range(kx):
        a, b = o.vdspec(gx[k], gy[k], 2)
        ok(vorl[k], a); ok(divl[k], b)
    assert pos[0] == raw.size


def standin_physics(o, st, utend, vtend, ttend, trtend):
    """support/host_physics.f90 restated: the linear terms the stand-in `physics` module adds (time level 1 of the prognostics,
    phi = get_geopotential(t(:,:,:,1), phis) as tendencies.f90:203)."""
    phi = o.geopotential(st["t"][0], st["phis"])
    pslg = o.spec_to_grid(st["ps"][0], 1)
    for k in range(o.kx):
        utend[k] = utend[k] + 1.0e-12 * o.spec_to_grid(phi[k], 1)
        vtend[k] = 0.999 * vtend[k] + 1.0e-3 * o.spec_to_grid(st["vor"][0, k], 1)
        ttend[k] = ttend[k] - 1.0e-6 * (o.spec_to_grid(st["t"][0, k], 1) - 250.0)
        g = o.spec_to_grid(st["tr"][0, k], 1) + o.spec_to_grid(st["div"][0, k], 1)
        trtend[k] = trtend[k] + 1.0e-7 * pslg - 1.0e-6 * g


@pytest.mark.gpu
@pytest.mark.parametrize("phys", [False, True])
@pytest.mark.parametrize("tag", ["t30", "t63"])
def test_time_stepping_dropin_vs_oracle(tag, phys, tmp_path, oracle_factory):
    """The `time_stepping` drop-in (fortran/time_stepping.f90): a flang-built main loop calls first_step and step(2, 2, 2*delt)
    as the model does (time_stepping.f90:11-24, :35-118 without the column physics); the prognostics stay in HBM, the leapfrog
    step is one captured graph.  After the start-up sequence and after each leapfrog step: both time levels of the five
    prognostics, the geopotential and the tendencies the step applied, against the oracle's call-by-call sequence at 1e-12
    of each array's maximum, in the plain norm and with the global mean removed.

    phys: the -DSPDY_WITH_PHYSICS build -- step() calls the model's physics%get_physical_tendencies on the host between the grid
    tendencies and the direct transforms (tendencies.f90:203-206); support/host_physics.f90 stands in for the column physics."""
    from dynstep import ROB, WIL, state, oracle_dynamics_step, wave_relerr
    exe = os.path.join(FDIR, "build", tag, "dropin_step_phys" if phys else "dropin_step")
    hook = standin_physics if phys else None
    if not os.path.exists(exe):
        pytest.skip("Fortran driver not built (no flang on this box and no prebuilt binary)")
    o = oracle_factory(tag)
    nx, mx, kx = o.nx, o.mx, o.kx
    st = state(o, 8000)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        for n in ("vor", "div", "t", "tr", "ps", "phis", "tcorh", "qcorh"):
            f.write(np.ascontiguousarray(st[n]).tobytes())
    nleap = 2
    r = subprocess.run([exe, str(fin), str(fout), str(nleap)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(fout, np.float64).view(np.complex128)
    pos = [0]

    def take(*shape):
        n = int(np.prod(shape))
        a = raw[pos[0]:pos[0] + n]
        pos[0] += n
        return a.reshape(shape)

    delt = float(np.float32(86400.0) / np.float32(36))               # params.f90:31
    # first_step (time_stepping.f90:11-24): forward half step from level 1, leapfrog step without filter, then 2*delt
    o.tail_init(0.5 * delt); ref, _ = oracle_dynamics_step(o, st, 1, 0.5 * delt, 0.0, j2=1, physics=hook)
    o.tail_init(delt); ref, out = oracle_dynamics_step(o, ref, 1, delt, 0.0, j2=2, physics=hook)
    o.tail_init(2 * delt)
    worst = {}
    snap_want = o.output(ref["vor"][0], ref["div"][0], ref["t"][0], ref["tr"][0], out["phi"], ref["ps"][0])   # snapshot after first_step
    for rec in range(1 + nleap):
        if rec:
            ref, out = oracle_dynamics_step(o, ref, 2, 2 * delt, ROB, j2=2, physics=hook)
        got = {n: take(2, kx, nx, mx) for n in ("vor", "div", "t", "tr")}
        got["ps"], got["phi"] = take(2, nx, mx), take(kx, nx, mx)
        for n in ("vordt", "divdt", "tdt", "trdt"):
            got[n] = take(kx, nx, mx)
        got["psdt"] = take(nx, mx)
        for n, a in got.items():
            want = ref[n] if n in ref and n != "phi" else out[n]
            e = max(synth.relerr(a, want), wave_relerr(a, want))
            worst[n] = max(worst.get(n, 0.0), e)
            assert e <= TOL, (tag, rec, n, e)
    assert pos[0] == raw.size
    if not phys:
        # an UNMODIFIED host: its main loop reads the host arrays right after step() (speedy.f90:41-50) and never asks for a
        # download.  With $SPDY_HOST_REFRESH=1 step() keeps them current itself (time_stepping%host_refresh_interval) -- same
        # bytes as with explicit prognostics_from_device calls; without it they are stale (the initial state), which is what
        # the option is for
        fout2 = tmp_path / "out_unmodified.bin"
        env = dict(os.environ, DROPIN_UNMODIFIED_HOST="1", SPDY_HOST_REFRESH="1")
        r2 = subprocess.run([exe, str(fin), str(fout2), str(nleap)], capture_output=True, text=True, timeout=300, env=env)
        assert r2.returncode == 0, r2.stdout + r2.stderr
        raw2 = np.fromfile(fout2, np.float64).view(np.complex128)
        nprog = (4 * 2 * kx + 2 + kx) * nx * mx                      # vor, div, t, tr, ps, phi of one record
        rec_len = raw.size // (1 + nleap)
        for rec in range(1 + nleap):
            assert np.array_equal(raw2[rec * rec_len:][:nprog], raw[rec * rec_len:][:nprog]), rec
        env.pop("SPDY_HOST_REFRESH")
        r3 = subprocess.run([exe, str(fin), str(tmp_path / "out_stale.bin"), str(nleap)], capture_output=True, text=True, timeout=300, env=env)
        stale = np.fromfile(tmp_path / "out_stale.bin", np.float64).view(np.complex128)
        assert r3.returncode == 0 and not np.array_equal(stale[nleap * rec_len:][:nprog], raw[nleap * rec_len:][:nprog])
    # the gridded snapshot after the start-up sequence, from the device-resident prognostics (output_fields_from_device = the computing
    # lines of input_output.f90:183-205; the oracle's restatement of them is pinned bit for bit: test_output_fields_pinned).
    # the device state differs from the oracle's by ~1e-14, so a value next to a float32 rounding boundary may land on the
    # other side: at most one value in 2000 may differ at all, and by no more than half a float32 ulp of the array's maximum
    snap = np.fromfile(str(fout) + ".snapshot", np.float32)
    il, ix = o.il, o.ix
    assert all(np.isfinite(b).all() for b in snap_want)
    assert snap.size == (5 * kx + 1) * il * ix
    flips = 0
    for i, (n, b) in enumerate(zip(("u", "v", "t", "q", "phi", "ps"), snap_want)):
        a = snap[i * kx * il * ix:][:b.size].reshape(b.shape)
        ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
        # (a value near zero moves by many of ITS ulps when the state moves by 1e-14 of the maximum: the bound is half a
        # float32 ulp of the array's maximum, and values that differ at all are counted)
        assert np.abs(a.astype(np.float64) - b.astype(np.float64)).max() <= 6e-8 * np.abs(b).max(), (tag, n, int(ulp.max()))
        flips += int((ulp != 0).sum())
    assert flips <= snap.size // 2000, flips
    worst["snapshot_flips"] = float(flips)
    print("\n[time_stepping drop-in %s%s] worst relative errors: " % (tag, " + host physics" if phys else "") + " ".join("%s %.1e" % kv for kv in worst.items()))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["t30", "t63"])
def test_sharded_step_from_one_fortran_process(tag, tmp_path, oracle_factory):
    """One flang-built process, one OpenMP thread + plan per rank (support/dropin_step_sharded.f90): each leapfrog step is one
    spdy_sharded_step_dev call per rank through the ISO_C_BINDING interfaces (spdy_c.f90), the ranks joined by an in-process
    group.  The driver itself fails unless every rank ends with the same bytes; here: 1, 2, 3 and 8 ranks give the same
    bytes as each other (the transforms are level-independent, whoever runs them) and hold the north star's 1e-12 against
    the oracle's call-by-call step(2, 2, 2*delt) (time_stepping.f90:35-118)."""
    from dynstep import ROB, state, oracle_dynamics_step, wave_relerr
    exe = os.path.join(FDIR, "build", tag, "dropin_step_sharded")
    if not os.path.exists(exe):
        pytest.skip("Fortran driver not built (no flang on this box and no prebuilt binary)")
    o = oracle_factory(tag)
    nx, mx, kx = o.nx, o.mx, o.kx
    st = state(o, 8100)
    fin = tmp_path / "in.bin"
    with open(fin, "wb") as f:
        for n in ("vor", "div", "t", "tr", "ps", "phis", "tcorh", "qcorh"):
            f.write(np.ascontiguousarray(st[n]).tobytes())
    nleap = 2
    env = dict(os.environ, SPDY_COMM_TIMEOUT_S="60")
    raws = {}
    for world in (1, 2, 3, 8):
        fout = tmp_path / ("out%d.bin" % world)
        r = subprocess.run([exe, str(fin), str(fout), str(nleap), str(world)], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.count("transforms levels") == world, r.stdout
        raws[world] = np.fromfile(fout, np.float64).view(np.complex128)
        assert np.array_equal(raws[world].view(np.int64), raws[1].view(np.int64)), world
        # ... and the TRANSPOSED form of the same step (levels <-> point / coefficient ranges; include/spdy.h) from the same Fortran
        # host: the environment selects it when the communicators are created, the driver gathers the state before reading it back.
        # Same bytes as the all-gather form at T30; at T63 to rounding (there the all-gather form applies vds inside the spectral
        # step, the transposed form before its exchange).
        ft = tmp_path / ("out%dt.bin" % world)
        r = subprocess.run([exe, str(fin), str(ft), str(nleap), str(world)], capture_output=True, text=True, timeout=300,
                           env=dict(env, SPDY_SHARD_TRANSPOSE="1"))
        assert r.returncode == 0, r.stdout + r.stderr
        tr_ = np.fromfile(ft, np.float64).view(np.complex128)
        if tag == "t30":
            assert np.array_equal(tr_.view(np.int64), raws[1].view(np.int64)), ("transposed", world)
        else:
            assert tr_.shape == raws[1].shape and synth.relerr(tr_, raws[1]) <= 1e-13, ("transposed", world, synth.relerr(tr_, raws[1]))
    delt = float(np.float32(86400.0) / np.float32(36))               # params.f90:31
    o.tail_init(2 * delt)
    ref = st
    for _ in range(nleap):
        ref, out = oracle_dynamics_step(o, ref, 2, 2 * delt, ROB, j2=2)
    raw, pos, worst = raws[1], 0, {}
    shapes = [(n, (2, kx, nx, mx)) for n in ("vor", "div", "t", "tr")] + [("ps", (2, nx, mx)), ("phi", (kx, nx, mx))] + \
             [(n, (kx, nx, mx)) for n in ("vordt", "divdt", "tdt", "trdt")] + [("psdt", (nx, mx))]
    for n, shape in shapes:
        a = raw[pos:pos + int(np.prod(shape))].reshape(shape)
        pos += a.size
        want = ref[n] if n in ref and n != "phi" else out[n]
        worst[n] = max(synth.relerr(a, want), wave_relerr(a, want))
        assert worst[n] <= TOL, (tag, n, worst[n])
    assert pos == raw.size
    print("\n[sharded step from one Fortran process, %s, 1/2/3/8 ranks bit-equal] worst relative errors: " % tag
          + " ".join("%s %.1e" % kv for kv in worst.items()))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["t30", "t63"])
def test_deferred_step_launches_same_state(tag):
    """time_stepping%steps_per_launch ($SPDY_STEPS_PER_LAUNCH): plain launches (0, the default), one captured graph per step (1),
    or K leapfrog steps collected and sent as one graph of K captured steps.  The same kernels in the same order: 50 steps from
    the same state must end in the same bits (the driver prints a checksum of the final state) for K = 0, 1, 4 (two steps
    left over for the flush) and 8."""
    exe = os.path.join(FDIR, "build", tag, "dropin_step")
    if not os.path.exists(exe):
        pytest.skip("Fortran driver not built (no flang on this box and no prebuilt binary)")
    sums = {}
    for k in ("0", "1", "4", "8"):
        r = subprocess.run([exe, "time", "50"], capture_output=True, text=True, timeout=300, env=dict(os.environ, SPDY_STEPS_PER_LAUNCH=k))
        assert r.returncode == 0, r.stdout + r.stderr
        f = r.stdout.split()
        assert len(f) >= 6, r.stdout
        sums[k] = f[5]
    assert sums["0"] == sums["1"] == sums["4"] == sums["8"], sums
