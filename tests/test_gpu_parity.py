"""GPU: parity of the HIP path (through the C-ABI) against the reference.

Bar (north_star / SURVEY.md s8c):  max|x - x_ref| / max|x_ref| <= 1e-12  per output array, FP64.
The checker is the golden vectors (reference outputs) and the C oracle (pinned to them by
tests/test_oracle_golden.py); nothing here reads /root/reference.
"""
import numpy as np
import pytest

import synth
from conftest import TOL
from synth import tail_inputs

pytestmark = pytest.mark.gpu
TAGS = ("t30", "t63")


@pytest.fixture(scope="module")
def plans():
    import speedy_f90_amd as s
    cache = {}

    def get(tag, max_batch=256, fused=-1):
        key = (tag, max_batch)
        if key not in cache:
            cache[key] = s.Spectral(tag, kx=8, max_batch=max_batch, device=0)
        cache[key].set_fused(fused)      # 1 = fused single-pass kernels (T30), 0 = four-kernel path
        return cache[key]
    yield get
    for p in cache.values():
        p.close()


def ok(x, ref, tol=TOL):
    assert x.shape == ref.shape
    err = synth.relerr(x, ref)
    assert err <= tol, err


@pytest.mark.parametrize("tag,fused", [("t30", 0), ("t30", 1), ("t63", 0), ("t63", 1)])
def test_stages_vs_golden(tag, fused, plans, golden):
    g, sp = golden(tag), plans(tag, fused=fused)
    nb = g["leginv"].shape[0]
    S, G = g["S"][:nb], g["G"][:nb]
    ok(sp.legendre_inv(S), g["leginv"])
    ok(sp.fourier_inv(g["leginv"], 1), g["finv1"])
    ok(sp.fourier_inv(g["leginv"], 2), g["finv2"])
    ok(sp.fourier_dir(G), g["fdir"])
    ok(sp.legendre_dir(g["fdir"]), g["legdir"])
    ok(sp.spec_to_grid(S, 1), g["s2g1"])
    ok(sp.spec_to_grid(S, 2), g["s2g2"])
    ok(sp.grid_to_spec(G), g["g2s"])
    # single-field drop-in signatures
    ok(sp.spec_to_grid(S[0], 2), g["s2g2"][0])
    ok(sp.grid_to_spec(G[0]), g["g2s"][0])
    s1 = sp.grid_to_spec(np.ones(sp.grid_shape))
    ok(s1, g["ones_g2s"])
    # structural facts of the reference: row nx and everything outside the triangle is exactly 0
    out = sp.grid_to_spec(G)
    assert np.all(out[:, -1, :] == 0)
    l = np.add.outer(np.arange(sp.nx), np.arange(sp.mx))
    assert np.all(out[:, l > sp.trunc + 1] == 0)
    assert np.all(sp.fourier_dir(G)[:, :, 1] == 0)      # Im(m'=0) written as 0 (fourier.f90:76)


@pytest.mark.parametrize("tag,fused", [("t30", 0), ("t30", 1), ("t63", 0), ("t63", 1)])
def test_operators_vs_golden(tag, fused, plans, golden):
    g, sp = golden(tag), plans(tag, fused=fused)
    S, G = g["S"], g["G"]
    ok(sp.laplacian(S[0]), g["lap"])
    ok(sp.inverse_laplacian(S[0]), g["invlap"])
    ok(sp.trunct(S[0]), g["trunct"])
    dx, dy = sp.grad(S[0]); ok(dx, g["grad_dx"]); ok(dy, g["grad_dy"])
    a, b = sp.vds(S[0], S[1]); ok(a, g["vds_vor"]); ok(b, g["vds_div"])
    a, b = sp.uvspec(S[0], S[1]); ok(a, g["uv_u"]); ok(b, g["uv_v"])
    a, b = sp.vdspec(G[0], G[1], 2); ok(a, g["vdspec2_vor"]); ok(b, g["vdspec2_div"])
    a, b = sp.vdspec(G[0], G[1], 1); ok(a, g["vdspec1_vor"]); ok(b, g["vdspec1_div"])


@pytest.mark.parametrize("tag", TAGS)
def test_tail_vs_golden(tag, plans, golden):
    g, sp = golden(tag), plans(tag)
    div, t, ps = tail_inputs(sp.kx, sp.nx, sp.mx)
    for dt in g["dts"]:
        key = "dt%d_" % int(dt)
        sp.initialize_implicit(float(dt))
        if dt in g["imp_dts"]:
            a, b, c = sp.implicit_terms(div, t, ps)
            ok(a, g[key + "imp_div_out"]); ok(b, g[key + "imp_t_out"]); ok(c, g[key + "imp_ps_out"])
        if key + "hdiff2d" in g.files:
            ok(sp.do_horizontal_diffusion(ps, 2 * ps, g[key + "dmps"], g[key + "dmp1s"]), g[key + "hdiff2d"])
        if key + "hdiff3d" in g.files:
            ok(sp.do_horizontal_diffusion(t, div, g[key + "dmp"], g[key + "dmp1"]), g[key + "hdiff3d"])


@pytest.mark.parametrize("tag,nb,fused", [("t30", 1, 0), ("t30", 7, 0), ("t30", 48, 0), ("t30", 73, 0), ("t30", 91, 0),
                                          ("t30", 129, 0), ("t30", 1, 1), ("t30", 2, 1), ("t30", 5, 1), ("t30", 48, 1),
                                          ("t30", 73, 1), ("t30", 91, 1), ("t30", 129, 1), ("t30", 255, 1),
                                          ("t63", 1, 0), ("t63", 9, 0), ("t63", 96, 0),
                                          ("t63", 1, 1), ("t63", 2, 1), ("t63", 9, 1), ("t63", 96, 1), ("t63", 255, 1)])
def test_batches_vs_oracle(tag, nb, fused, plans, oracle_factory):
    """Model-shaped batches (48 / 73 / 91) and ragged ones (partial 4-field tiles of the
    fused kernels), mixed kcos, through both kernel paths."""
    sp, o = plans(tag, fused=fused), oracle_factory(tag)
    S = synth.spectra(nb, sp.trunc, first=1000, full_rows=True)
    G = synth.grids(nb, sp.ix, sp.il, first=1000)
    kcos = np.array([2 if (b % 6) in (4, 5) else 1 for b in range(nb)], np.int32)   # u,v slots of every 6
    got = sp.spec_to_grid(S, kcos)
    ref = np.stack([o.spec_to_grid(S[b], int(kcos[b])) for b in range(nb)])
    for b in range(nb):
        ok(got[b], ref[b])
    got = sp.grid_to_spec(G)
    ref = np.stack([o.grid_to_spec(G[b]) for b in range(nb)])
    for b in range(nb):
        ok(got[b], ref[b])


def test_empty_batch(plans):
    sp = plans("t30")
    assert sp.grid_to_spec(np.zeros((0,) + sp.grid_shape)).shape == (0,) + sp.spec_shape
    assert sp.spec_to_grid(np.zeros((0,) + sp.spec_shape, np.complex128)).shape == (0,) + sp.grid_shape


def test_inactive_coefficients_are_ignored(plans, oracle_factory):
    """The inverse transform must not read l > trunc+1 (legendre.f90:93: m <= nsh2(n))."""
    sp, o = plans("t30", fused=1), oracle_factory("t30")
    S = synth.spectra(2, 30, first=5, full_rows=True)
    junk = S.copy()
    l = np.add.outer(np.arange(sp.nx), np.arange(sp.mx))
    junk[:, l > sp.trunc + 1] = 1e300 + 1e300j
    ref = np.stack([o.spec_to_grid(S[b], 1) for b in range(2)])
    for fused in (1, 0):
        sp.set_fused(fused)
        ok(sp.spec_to_grid(junk, 1), ref)
    # Im(m'=0) is dropped by fourier_inv (fourier.f90:34-36)
    junk = S.copy(); junk[:, :, 0] += 3.0j
    for fused in (1, 0):
        sp.set_fused(fused)
        ok(sp.spec_to_grid(junk, 1), ref)


def test_max_batch_enforced(plans):
    import speedy_f90_amd as s
    sp = plans("t30", 256)
    with pytest.raises(s.SpdyError):
        sp.grid_to_spec(np.zeros((257,) + sp.grid_shape))


@pytest.mark.parametrize("tag,nb,fused", [("t30", 6144, 1), ("t30", 6143, 1), ("t30", 6144, 0), ("t63", 1536, 0), ("t63", 1536, 1), ("t63", 1535, 1)])
def test_full_size_device_resident(tag, nb, fused, oracle_factory):
    """BASELINE sizes (B=6144 at T30, 1536 at T63; ~226 MB of grid data), device-resident path on
    torch's stream.  Checked against the oracle on a strided sample of fields plus two
    size-independent properties over the whole batch: linearity and batch independence."""
    import torch
    import speedy_f90_amd as s
    o = oracle_factory(tag)
    sp = s.Spectral(tag, kx=8, max_batch=nb, device=0)
    sp.set_fused(fused)
    sp.use_torch_stream()
    uniq = 64
    G = synth.grids(uniq, sp.ix, sp.il, first=7000)
    reps = (nb + uniq - 1) // uniq
    dG = torch.from_numpy(G).cuda().repeat(reps, 1, 1)[:nb].contiguous()
    scale = torch.arange(1, nb + 1, dtype=torch.float64, device="cuda").view(nb, 1, 1) / nb
    dG = dG * scale                                   # every field distinct
    dS = torch.zeros((nb, sp.nx, sp.mx), dtype=torch.complex128, device="cuda")
    dG2 = torch.zeros_like(dG)
    sp.grid_to_spec_dev(dG, dS)
    sp.spec_to_grid_dev(dS, dG2, kcos=1)
    torch.cuda.synchronize()
    hS, hG2, hG = dS.cpu().numpy(), dG2.cpu().numpy(), dG.cpu().numpy()
    for b in list(range(0, nb, 97)) + [nb - 1]:
        ok(hS[b], o.grid_to_spec(hG[b]))
        ok(hG2[b], o.spec_to_grid(hS[b], 1))
    # batch independence + linearity: field b is (b+1)/nb times its template field
    base = hS[:uniq] / (np.arange(1, uniq + 1).reshape(-1, 1, 1) / nb)
    rel = np.abs(hS / (np.arange(1, nb + 1).reshape(-1, 1, 1) / nb) - np.tile(base, (reps, 1, 1))[:nb])
    assert rel.max() <= 1e-12 * np.abs(base).max()
    # T(a*x + y) = a*T(x) + T(y)
    a = 0.37
    dMix = a * dG[:uniq] + dG[uniq:2 * uniq]
    dSm = torch.zeros((uniq, sp.nx, sp.mx), dtype=torch.complex128, device="cuda")
    sp.grid_to_spec_dev(dMix.contiguous(), dSm)
    torch.cuda.synchronize()
    lin = (a * dS[:uniq] + dS[uniq:2 * uniq] - dSm).abs().max().item()
    assert lin <= 1e-12 * dS[:2 * uniq].abs().max().item()
    sp.close()


def test_device_ops_and_profile(oracle_factory):
    """Device-pointer operator entry points + the HIP-event profiler used by bench.py."""
    import torch
    import speedy_f90_amd as s
    o = oracle_factory("t30")
    sp = s.Spectral("t30", kx=8, max_batch=64, device=0)
    sp.use_torch_stream()
    S = synth.spectra(16, 30, first=300, full_rows=True)
    G = synth.grids(16, sp.ix, sp.il, first=300)
    dS = torch.from_numpy(S).cuda()
    u, v = torch.zeros_like(dS[:8]), torch.zeros_like(dS[:8])
    sp.uvspec_dev(dS[:8].contiguous(), dS[8:].contiguous(), u, v)
    dG = torch.from_numpy(G).cuda()
    vor, div = torch.zeros_like(dS[:8]), torch.zeros_like(dS[:8])
    sp.vdspec_dev(dG[:8].contiguous(), dG[8:].contiguous(), vor, div, 2)
    torch.cuda.synchronize()
    for b in range(8):
        ru, rv = o.uvspec(S[b], S[8 + b])
        ok(u[b].cpu().numpy(), ru); ok(v[b].cpu().numpy(), rv)
        rvor, rdiv = o.vdspec(G[b], G[8 + b], 2)
        ok(vor[b].cpu().numpy(), rvor); ok(div[b].cpu().numpy(), rdiv)
    # implicit + hdiff on device
    sp.initialize_implicit(4800.0); o.tail_init(4800.0)
    d, t, p = tail_inputs(8, sp.nx, sp.mx)
    dd, dt_, dp = torch.from_numpy(d).cuda(), torch.from_numpy(t).cuda(), torch.from_numpy(p).cuda()
    out = torch.zeros_like(dt_)
    sp.hdiff_dev(dt_, dd, "dmp", "dmp1", out)
    sp.implicit_terms_dev(dd, dt_, dp)
    torch.cuda.synchronize()
    ok(out.cpu().numpy(), o.hdiff(t, d, o.table("dmp"), o.table("dmp1")))
    ra, rb, rc = o.implicit_terms(d, t, p)
    ok(dd.cpu().numpy(), ra); ok(dt_.cpu().numpy(), rb); ok(dp.cpu().numpy(), rc)
    # profiler
    sp.set_profiling(True)
    g2 = torch.zeros_like(dG)
    s2 = torch.zeros_like(dS)
    sp.set_fused(0)
    for _ in range(3):
        sp.grid_to_spec_dev(dG, s2); sp.spec_to_grid_dev(s2, g2)
    prof = sp.get_profile()
    assert all(prof[k][1] == 3 and prof[k][0] > 0 for k in sp.KERNEL_KINDS[:4])     # forced four-kernel path
    sp.set_fused(-1)                                                                  # auto = fused at T30
    for _ in range(2):
        sp.grid_to_spec_dev(dG, s2); sp.spec_to_grid_dev(s2, g2)
    prof = sp.get_profile()
    assert prof["s2g_fused"][1] == 2 and prof["g2s_fused"][1] == 2 and prof["fourier_inv"][1] == 0
    sp.set_profiling(False)
    sp.close()


@pytest.mark.parametrize("tag", ["t30k5", "t30k7", "t63k16"])
def test_tail_other_level_counts(tag, golden, oracle_factory):
    """The reference's other sigma-level sets (geometry.f90:42-48: kx = 5, 7) and the 16-level T63 configuration
    (BASELINE config 5; half levels supplied with spdy_plan_set_sigma), against golden outputs of flang builds of the
    reference at those level counts (oracle/build_ref.sh, tests/golden/ref_extra.npz)."""
    import speedy_f90_amd as s
    from conftest import VARIANTS
    from golden.make_golden import L16_SUB, geop_inputs
    g, o = golden("extra"), oracle_factory(tag)
    trunc, ix, iy, kx = VARIANTS[tag]
    sp = s.Spectral((trunc, ix, iy), kx=kx, max_batch=2 * kx, device=0)
    cut = (lambda a: a[L16_SUB]) if tag == "t63k16" else (lambda a: a)
    if tag == "t63k16":
        with pytest.raises(s.SpdyError):                  # no sigma levels yet: refused, not guessed
            sp.initialize_implicit(4800.0)
        sp.set_sigma(synth.SIGMA_L16)
    for name in ("hsg", "dhs", "fsg", "dhsr", "fsgr", "tcorv", "qcorv"):
        assert np.array_equal(sp.table(name), g[tag + "_" + name]), name
    d, t, p = tail_inputs(kx, sp.nx, sp.mx)
    for dt in (1200.0, 4800.0):
        key = "%s_dt%d_" % (tag, int(dt))
        if key + "imp_div_out" not in g.files:
            continue
        sp.initialize_implicit(dt); o.tail_init(dt)
        gd, gt, gp = sp.implicit_terms(d, t, p)
        ok(cut(gd), g[key + "imp_div_out"]); ok(cut(gt), g[key + "imp_t_out"]); ok(gp, g[key + "imp_ps_out"])
        rd, rt, rp = o.implicit_terms(d, t, p)            # the whole arrays against the (pinned) oracle
        ok(gd, rd); ok(gt, rt); ok(gp, rp)
        dmp, dmp1 = o.table("dmpd").reshape(sp.nx, sp.mx), o.table("dmp1d").reshape(sp.nx, sp.mx)
        ok(cut(sp.do_horizontal_diffusion(t, d, dmp, dmp1)), g[key + "hdiff3d"])
    T, phis = geop_inputs(kx, sp.nx, sp.mx)
    ok(cut(sp.get_geopotential(T, phis)), g[tag + "_geop"])
    sp.close()


@pytest.mark.parametrize("tag", TAGS)
def test_geopotential_vs_golden(tag, plans, golden):
    from golden.make_golden import geop_inputs
    sp = plans(tag)
    T, phis = geop_inputs(sp.kx, sp.nx, sp.mx)
    ok(sp.get_geopotential(T, phis), golden("extra")[tag + "_geop"])


def test_kcos_other_values_mean_cosgr(plans, golden):
    """fourier.f90:47-51: kcos == 1 is plain, ANY other value multiplies by cosgr -- also in the fused entry points."""
    g, sp = golden("t30"), plans("t30", fused=1)
    S = g["S"][:2]
    for kc in (0, 2, 3, -1):
        ok(sp.spec_to_grid(S, kc), g["s2g2"])
        ug, vg = sp.uvspec_to_grid(S[0], S[1], kc)
        ug2, vg2 = sp.uvspec_to_grid(S[0], S[1], 2)
        assert np.array_equal(ug, ug2) and np.array_equal(vg, vg2)


@pytest.mark.parametrize("tag", ["t30", "t63"])
def test_host_staging_routes_agree(tag, monkeypatch):
    """Host-pointer calls: the pinned host-mapped staging route of small calls (the kernels read and write the staging buffers
    across the link themselves, spdy_plan::hstage) and the device-staging route (hipMemcpyAsync both ways, SPDY_HOST_STAGE_KB=0)
    give the same bits -- transforms with one flag and with per-field kcos, the two-array operators, in-place trunct -- and a
    batch larger than the threshold falls back to the device route inside the same plan."""
    import speedy_f90_amd as s
    res = {}
    for kb in ("0", "512"):
        monkeypatch.setenv("SPDY_HOST_STAGE_KB", kb)          # read when the plan allocates its staging (first host-pointer call)
        sp = s.Spectral(tag, kx=8, max_batch=64, device=0)
        S = synth.spectra(64, sp.trunc, first=4100, full_rows=True)
        G = synth.grids(64, sp.ix, sp.il, first=4100)
        out = []
        for nb in (1, 3, 64):                                  # 64 grids: 2.4 MB (T30) / 9.4 MB (T63) per buffer -> device route
            kc = np.array([1 + (b % 3 == 1) for b in range(nb)], np.int32)
            out += [sp.spec_to_grid(S[:nb], 2), sp.spec_to_grid(S[:nb], kc), sp.grid_to_spec(G[:nb])]
            out += list(sp.uvspec(S[:nb], S[nb - 1::-1])) + list(sp.vdspec(G[:nb], G[nb - 1::-1], 2)) + [sp.trunct(S[:nb])]
        out += [sp.spec_to_grid(S[5], 1), sp.grid_to_spec(G[5])]
        res[kb] = out
        sp.close()
    assert len(res["0"]) == len(res["512"])
    for a, b in zip(res["0"], res["512"]):
        assert a.shape == b.shape and np.array_equal(a, b)
