"""Operator + transform sequences done in one pass (include/spdy.h: spdy_uvspec_to_grid_dev, spdy_grad_to_grid_dev,
and vdspec, which is one kernel at T30): against the oracle's own sequence of reference calls
(uvspec / grad then spec_to_grid, spectral.f90:98-227) and against the library's unfused composition."""
import numpy as np
import pytest

import synth
from conftest import TOL

pytestmark = pytest.mark.gpu


def ok(a, b, tol=TOL):
    assert a.shape == b.shape
    assert synth.relerr(a, b) <= tol, synth.relerr(a, b)


@pytest.mark.parametrize("res,nb", [("t30", 1), ("t30", 24), ("t30", 257), ("t63", 5), ("t63f", 5), ("t63f", 1), ("t63", 41)])
@pytest.mark.parametrize("kcos", [2, 1])
def test_uvspec_and_grad_to_grid(res, nb, kcos, oracle_factory):
    import torch
    import speedy_f90_amd as s
    o = oracle_factory(res[:3])
    sp = s.Spectral(res[:3], kx=8, max_batch=max(nb, 8), device=0)
    if res.endswith("f"):
        sp.set_fused(1)                                        # T63: operator kernel + ONE two-segment fused launch
    sp.use_torch_stream()
    S = synth.spectra(2 * nb, sp.trunc, first=700, full_rows=True)     # row nx populated: uvspec must not care
    vor, div = torch.from_numpy(S[:nb]).cuda(), torch.from_numpy(S[nb:]).cuda()
    shape = (nb, sp.il, sp.ix)
    ug, vg, gx, gy = (torch.full(shape, np.nan, dtype=torch.float64, device="cuda") for _ in range(4))
    sp.uvspec_to_grid_dev(vor, div, ug, vg, kcos)
    sp.grad_to_grid_dev(vor, gx, gy, kcos)
    # the unfused composition through the same library
    u, v, dx, dy = (torch.zeros_like(vor) for _ in range(4))
    sp.uvspec_dev(vor, div, u, v)
    check = __import__("speedy_f90_amd").check
    check(sp.lib.spdy_grad_dev(sp.h, nb, sp._dp(vor), sp._dp(dx), sp._dp(dy)))
    ref = [torch.zeros(shape, dtype=torch.float64, device="cuda") for _ in range(4)]
    for src, dst in zip((u, v, dx, dy), ref):
        sp.spec_to_grid_dev(src, dst, kcos=kcos)
    torch.cuda.synchronize()
    for got, want in zip((ug, vg, gx, gy), ref):
        ok(got.cpu().numpy(), want.cpu().numpy(), 1e-13)
    # the reference's own call sequence on a sample of the batch
    for b in sorted({0, nb // 2, nb - 1}):
        ru, rv = o.uvspec(S[b], S[nb + b])
        ok(ug[b].cpu().numpy(), o.spec_to_grid(ru, kcos)); ok(vg[b].cpu().numpy(), o.spec_to_grid(rv, kcos))
        rdx, rdy = o.grad(S[b])
        ok(gx[b].cpu().numpy(), o.spec_to_grid(rdx, kcos)); ok(gy[b].cpu().numpy(), o.spec_to_grid(rdy, kcos))
    sp.close()


@pytest.mark.parametrize("nb", [1, 2, 5, 16, 20])
def test_t63_operators_derived_on_load(nb, oracle_factory):
    """Row f1 at T63 (round 6): model-sized inverse launches evaluate uvspec / grad where the fused kernel loads its operands
    (csrc/spdy_fused_t63.inc: t63_inv_load_b_op -- the n +- 1 neighbours by lane exchange inside the wave, halo groups at the
    edge of a wave's slots) instead of running an operator kernel in front.  Checked (a) against the reference's own call
    sequence (uvspec / grad, then spec_to_grid: spectral.f90:98-110, 124-144, 173-196) through the oracle, at 1e-12, with the
    dead part of the rhomboid (l > trunc + 1) holding finite junk -- the reference's stencils read those entries at n + 1,
    whatever they hold, and so must the fold; (b) against the operator-kernel route of the same library (option
    t63_derive = 0) at 1e-13; (c) as segments of the five-segment launch of a model step next to plain segments."""
    import torch
    import speedy_f90_amd as s
    o = oracle_factory("t63")
    sp = s.Spectral("t63", kx=8, max_batch=max(3 * nb + 1, 8), device=0)
    sp.use_torch_stream()
    S = synth.spectra(5 * nb + 2, sp.trunc, first=8100, full_rows=True)
    rng = np.random.default_rng(808 + nb)
    l = np.arange(sp.mx)[None, :] + np.arange(sp.nx)[:, None]
    junk = rng.uniform(-1, 1, S.shape) + 1j * rng.uniform(-1, 1, S.shape)
    S = np.where(l[None] > sp.trunc + 1, junk, S)                         # the transforms ignore it; the operators' n + 1 reads do not
    dS = torch.from_numpy(S).cuda()
    vor, div, spl, psi = dS[:nb], dS[nb:2 * nb], dS[2 * nb:5 * nb + 1], dS[5 * nb + 1:]
    gs = (sp.il, sp.ix)
    f64 = lambda n: torch.full((n,) + gs, float("nan"), dtype=torch.float64, device="cuda")

    def run():
        out = dict(ug=f64(nb), vg=f64(nb), gx=f64(nb), gy=f64(nb), mug=f64(nb), mvg=f64(nb), mpl=f64(3 * nb + 1), mgx=f64(1), mgy=f64(1))
        sp.uvspec_to_grid_dev(vor, div, out["ug"], out["vg"], 2)
        sp.grad_to_grid_dev(vor, out["gx"], out["gy"], 1)
        sp.inverse_batch_grad_dev(vor, div, out["mug"], out["mvg"], spl, out["mpl"], psi, out["mgx"], out["mgy"], kcos_pairs=2, kcos=1, kcos_grad=2)
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in out.items()}
    a = run()
    sp.set_option("t63_derive", 0)
    b = run()
    sp.set_option("t63_derive", 1)
    for k in a:
        assert not np.isnan(a[k]).any(), k
        ok(a[k], b[k], 1e-13)
    for i in sorted({0, nb // 2, nb - 1}):
        ru, rv = o.uvspec(S[i], S[nb + i])
        ok(a["ug"][i], o.spec_to_grid(ru, 2)); ok(a["vg"][i], o.spec_to_grid(rv, 2))
        ok(a["mug"][i], o.spec_to_grid(ru, 2)); ok(a["mvg"][i], o.spec_to_grid(rv, 2))
        rdx, rdy = o.grad(S[i])
        ok(a["gx"][i], o.spec_to_grid(rdx, 1)); ok(a["gy"][i], o.spec_to_grid(rdy, 1))
    rdx, rdy = o.grad(S[5 * nb + 1])
    ok(a["mgx"][0], o.spec_to_grid(rdx, 2)); ok(a["mgy"][0], o.spec_to_grid(rdy, 2))
    ok(a["mpl"][3 * nb], o.spec_to_grid(S[5 * nb], 1))
    sp.close()


@pytest.mark.parametrize("res,nb", [("t30", 1), ("t30", 3), ("t30", 48), ("t30", 300), ("t63", 1), ("t63", 7)])
@pytest.mark.parametrize("kcos", [2, 1])
def test_vdspec_one_pass(res, nb, kcos, oracle_factory):
    """fused = 1: one kernel at T30; at T63 ONE two-segment fused launch + vds.  fused = 0: five kernels."""
    import torch
    import speedy_f90_amd as s
    o = oracle_factory(res)
    sp = s.Spectral(res, kx=8, max_batch=max(nb, 8), device=0)
    sp.use_torch_stream()
    G = synth.grids(2 * nb, sp.ix, sp.il, first=900)
    ug, vg = torch.from_numpy(G[:nb]).cuda(), torch.from_numpy(G[nb:]).cuda()
    outs = {}
    for fused in (1, 0):                                   # one kernel / five kernels
        sp.set_fused(fused)
        vor = torch.full((nb, sp.nx, sp.mx), np.nan, dtype=torch.complex128, device="cuda")
        div = torch.full_like(vor, np.nan)
        sp.vdspec_dev(ug, vg, vor, div, kcos)
        torch.cuda.synchronize()
        outs[fused] = (vor.cpu().numpy(), div.cpu().numpy())
    ok(outs[1][0], outs[0][0], 1e-13); ok(outs[1][1], outs[0][1], 1e-13)
    for b in sorted({0, nb // 2, nb - 1}):
        rvor, rdiv = o.vdspec(G[b], G[nb + b], kcos)
        ok(outs[1][0][b], rvor); ok(outs[1][1][b], rdiv)
    sp.close()


@pytest.mark.parametrize("res,nb", [("t30", 1), ("t30", 8), ("t30", 300), ("t63", 3)])
def test_host_pointer_level_stacks(res, nb, oracle_factory):
    """The host-pointer forms the Fortran level-stack extensions call; a large stack also exercises that the
    one-pass vdspec never reads a grid that another workgroup's output has overwritten."""
    import speedy_f90_amd as s
    o = oracle_factory(res)
    sp = s.Spectral(res, kx=8, max_batch=max(nb, 8), device=0)
    S = synth.spectra(2 * nb, sp.trunc, first=1300, full_rows=True)
    G = synth.grids(2 * nb, sp.ix, sp.il, first=1300)
    ug, vg = sp.uvspec_to_grid(S[:nb], S[nb:], 2)
    gx, gy = sp.grad_to_grid(S[:nb], 2)
    vor, div = sp.vdspec(G[:nb], G[nb:], 2)
    for b in sorted({0, nb // 3, nb - 1}):
        ru, rv = o.uvspec(S[b], S[nb + b])
        ok(ug[b], o.spec_to_grid(ru, 2)); ok(vg[b], o.spec_to_grid(rv, 2))
        rdx, rdy = o.grad(S[b])
        ok(gx[b], o.spec_to_grid(rdx, 2)); ok(gy[b], o.spec_to_grid(rdy, 2))
        a, c = o.vdspec(G[b], G[nb + b], 2)
        ok(vor[b], a); ok(div[b], c)
    # whole-stack consistency with the one-field-at-a-time calls of the same library
    one_vor, one_div = sp.vdspec(G[nb - 1], G[2 * nb - 1], 2)
    assert np.array_equal(vor[nb - 1], one_vor) and np.array_equal(div[nb - 1], one_div)
    sp.close()


def test_hdiff_multi_equals_separate_calls(oracle_factory):
    """The seven diffusion calls of a time step in one launch (spdy_hdiff_multi_dev) = the same calls one by one."""
    import torch
    import speedy_f90_amd as s
    from golden.make_golden import tail_inputs
    o = oracle_factory("t30")
    sp = s.Spectral("t30", kx=8, max_batch=16, device=0)
    sp.initialize_implicit(4800.0); o.tail_init(4800.0)
    names = [("dmp", "dmp1"), ("dmpd", "dmp1d"), ("dmp", "dmp1"), ("dmp", "dmp1"), ("dmps", "dmp1s"), ("dmps", "dmp1s"), ("dmpd", "dmp1d")]
    nlevs = [8, 8, 8, 8, 8, 8, 1]
    ops, want = [], []
    for i, ((a, b), nl) in enumerate(zip(names, nlevs)):
        f = torch.from_numpy(synth.spectra(nl, 30, first=2000 + 10 * i, full_rows=True)).cuda()
        t = torch.from_numpy(synth.spectra(nl, 30, first=3000 + 10 * i, full_rows=True)).cuda()
        single = torch.zeros_like(f)
        sp.hdiff_dev(f, t, a, b, single)
        out = torch.full_like(f, float("nan"))
        ops.append((f, t, a, b, out)); want.append(single)
    sp.hdiff_multi_dev(ops)
    sp.synchronize()
    for (f, t, a, b, out), single in zip(ops, want):
        assert torch.equal(out, single)
    f, t, a, b, out = ops[1]
    ok(out.cpu().numpy(), o.hdiff(f.cpu().numpy(), t.cpu().numpy(), o.table(a).reshape(sp.nx, sp.mx), o.table(b).reshape(sp.nx, sp.mx)))
    with pytest.raises(s.SpdyError):
        sp.hdiff_multi_dev(ops + ops)                       # 14 > SPDY_HDIFF_MAX_OPS
    sp.close()


@pytest.mark.parametrize("res,npairs,nplain", [("t30", 24, 25), ("t30", 1, 1), ("t30", 3, 8), ("t30", 300, 301), ("t63", 3, 5),
                                               ("t63f", 3, 5), ("t63f", 1, 1), ("t63f", 48, 49)])
def test_direct_batch_one_launch(res, npairs, nplain):
    """spdy_direct_batch_dev = vdspec of the pairs + grid_to_spec of the plain fields, bit for bit.  t63f = fused kernels
    pinned (spdy_plan_set_fused(1)): the batch is ONE three-segment launch (+ vds), as it is in auto mode."""
    import torch
    import speedy_f90_amd as s
    sp = s.Spectral(res[:3], kx=8, max_batch=max(npairs, nplain, 8), device=0)
    if res.endswith("f"):
        sp.set_fused(1)
    G = torch.from_numpy(synth.grids(2 * npairs + nplain, sp.ix, sp.il, first=4000)).cuda()
    ug, vg, gp = G[:npairs], G[npairs:2 * npairs], G[2 * npairs:]
    cs = (sp.nx, sp.mx)
    want = [torch.zeros((n,) + cs, dtype=torch.complex128, device="cuda") for n in (npairs, npairs, nplain)]
    sp.vdspec_dev(ug, vg, want[0], want[1], 2)
    sp.grid_to_spec_dev(gp, want[2])
    got = [torch.full((n,) + cs, float("nan"), dtype=torch.complex128, device="cuda") for n in (npairs, npairs, nplain)]
    sp.direct_batch_dev(ug, vg, got[0], got[1], gp, got[2], 2)
    sp.synchronize()
    for a, b in zip(got, want):
        if res == "t63":     # the separate calls form other field pairs than the one launch (pairs are formed inside a segment) -- equal to rounding is the contract
            ok(a.cpu().numpy(), b.cpu().numpy(), 1e-13)
        else:
            assert torch.equal(a, b)
    sp.close()


@pytest.mark.parametrize("res,npairs,nplain", [("t30", 16, 59), ("t30", 1, 1), ("t30", 5, 2), ("t30", 300, 299), ("t63", 3, 5),
                                               ("t63f", 3, 5), ("t63f", 1, 1), ("t63f", 16, 64)])
def test_inverse_batch_one_launch(res, npairs, nplain):
    """spdy_inverse_batch_dev = uvspec_to_grid of the pairs + spec_to_grid of the plain fields (mixed kcos), bit for bit.
    t63f = fused kernels pinned: uvspec + ONE three-segment launch."""
    import torch
    import speedy_f90_amd as s
    sp = s.Spectral(res[:3], kx=8, max_batch=max(npairs, nplain, 8), device=0)
    if res.endswith("f"):
        sp.set_fused(1)
    S = torch.from_numpy(synth.spectra(2 * npairs + nplain, sp.trunc, first=5000, full_rows=True)).cuda()
    vor, div, spl = S[:npairs], S[npairs:2 * npairs], S[2 * npairs:]
    kc = torch.tensor([1 + (i % 3 == 0) for i in range(nplain)], dtype=torch.int32, device="cuda")
    gs = (sp.il, sp.ix)
    want = [torch.zeros((n,) + gs, dtype=torch.float64, device="cuda") for n in (npairs, npairs, nplain)]
    sp.uvspec_to_grid_dev(vor, div, want[0], want[1], 2)
    sp.spec_to_grid_dev(spl, want[2], d_kcos=kc)
    got = [torch.full((n,) + gs, float("nan"), dtype=torch.float64, device="cuda") for n in (npairs, npairs, nplain)]
    sp.inverse_batch_dev(vor, div, got[0], got[1], spl, got[2], kcos_pairs=2, d_kcos=kc)
    sp.synchronize()
    for a, b in zip(got, want):
        if res == "t63":     # the separate calls form other field pairs than the one launch (pairs are formed inside a segment) -- equal to rounding is the contract
            ok(a.cpu().numpy(), b.cpu().numpy(), 1e-13)
        else:
            assert torch.equal(a, b)
    sp.close()


@pytest.mark.parametrize("res,npairs,nplain,ngrad", [("t30", 8, 32, 1), ("t30", 1, 1, 1), ("t30", 5, 3, 4), ("t30", 300, 299, 7),
                                                     ("t63", 3, 5, 1), ("t63f", 16, 64, 1), ("t63f", 3, 4, 3)])
def test_inverse_batch_with_gradient(res, npairs, nplain, ngrad):
    """spdy_inverse_batch_grad_dev = spdy_inverse_batch_dev + spdy_grad_to_grid_dev (tendencies.f90:89-107, 121-123): ONE launch
    at T30 (gradient tiles ride along as uvspec tiles with a zero vorticity and the grad tables), one five-segment fused launch at T63."""
    import torch
    import speedy_f90_amd as s
    sp = s.Spectral(res[:3], kx=8, max_batch=max(npairs + ngrad, nplain, 8), device=0)
    if res.endswith("f"):
        sp.set_fused(1)
    S = torch.from_numpy(synth.spectra(2 * npairs + nplain + ngrad, sp.trunc, first=6100, full_rows=True)).cuda()
    vor, div, spl, psi = S[:npairs], S[npairs:2 * npairs], S[2 * npairs:2 * npairs + nplain], S[2 * npairs + nplain:]
    gs = (sp.il, sp.ix)
    sizes = (npairs, npairs, nplain, ngrad, ngrad)
    want = [torch.zeros((n,) + gs, dtype=torch.float64, device="cuda") for n in sizes]
    sp.inverse_batch_dev(vor, div, want[0], want[1], spl, want[2], kcos_pairs=2, kcos=1)
    kg = 1 if ngrad == 4 else 2                               # the gradient's own kcos, different from the pairs' once
    sp.grad_to_grid_dev(psi, want[3], want[4], kg)
    got = [torch.full((n,) + gs, float("nan"), dtype=torch.float64, device="cuda") for n in sizes]
    sp.inverse_batch_grad_dev(vor, div, got[0], got[1], spl, got[2], psi, got[3], got[4], kcos_pairs=2, kcos=1, kcos_grad=kg)
    sp.synchronize()
    for a, b in zip(got, want):
        if res == "t63":     # the separate calls form other field pairs than the one launch -- equal to rounding is the contract
            ok(a.cpu().numpy(), b.cpu().numpy(), 1e-13)
        else:
            assert torch.equal(a, b)
    sp.close()


@pytest.mark.parametrize("res,npairs,segsizes,ngrad", [("t30", 8, (8, 8, 8, 8), 1), ("t30", 5, (5, 5, 5, 5), 1), ("t30", 7, (7, 1, 3, 2), 0),
                                                       ("t30", 3, (1, 0, 4), 2), ("t30", 300, (299, 301), 7),
                                                       ("t63f", 16, (16, 16, 16, 16), 1), ("t63f", 5, (5, 3, 1, 2), 0), ("t63", 3, (2, 3), 1)])
def test_inverse_batch_segments(res, npairs, segsizes, ngrad):
    """spdy_inverse_batch_segs_dev: the plain spectra read in place from up to four separate arrays (what tendencies.f90:89-101
    does with vor, div, t, tr) must give the bits of the single-array call on the concatenated stack -- odd segment sizes put
    the two fields of a T30 tile (and of a T63 pair at a segment's end) in different arrays."""
    import torch
    import speedy_f90_amd as s
    nplain = sum(segsizes)
    sp = s.Spectral(res[:3], kx=8, max_batch=max(npairs + ngrad, nplain, 8), device=0)
    if res.endswith("f"):
        sp.set_fused(1)
    S = torch.from_numpy(synth.spectra(2 * npairs + nplain + max(ngrad, 1), sp.trunc, first=6400, full_rows=True)).cuda()
    vor, div, spl, psi = S[:npairs], S[npairs:2 * npairs], S[2 * npairs:2 * npairs + nplain], S[2 * npairs + nplain:][:ngrad]
    # separate allocations, so that nothing is contiguous by accident
    parts, first = [], 0
    for n in segsizes:
        parts.append(spl[first:first + n].clone())
        first += n
    gs = (sp.il, sp.ix)
    sizes = (npairs, npairs, nplain, ngrad, ngrad)
    want = [torch.zeros((n,) + gs, dtype=torch.float64, device="cuda") for n in sizes]
    got = [torch.full((n,) + gs, float("nan"), dtype=torch.float64, device="cuda") for n in sizes]
    if ngrad:
        sp.inverse_batch_grad_dev(vor, div, want[0], want[1], spl, want[2], psi, want[3], want[4], kcos_pairs=2, kcos=1)
        sp.inverse_batch_segs_dev(vor, div, got[0], got[1], parts, got[2], psi, got[3], got[4], kcos_pairs=2, kcos=1)
    else:
        sp.inverse_batch_dev(vor, div, want[0], want[1], spl, want[2], kcos_pairs=2, kcos=1)
        sp.inverse_batch_segs_dev(vor, div, got[0], got[1], parts, got[2], kcos_pairs=2, kcos=1)
    sp.synchronize()
    for a, b in zip(got, want):
        if a.numel() == 0:
            continue
        if res.startswith("t63"):
            # a T63 pair is formed inside a segment: odd segments pair fields differently from the concatenated stack, and a
            # field's bits do not depend on its partner (tests/test_gpu_determinism.py) -- equal to rounding is the contract here
            ok(a.cpu().numpy(), b.cpu().numpy(), 1e-13)
        else:
            assert torch.equal(a, b)
    sp.close()
