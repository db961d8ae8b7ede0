"""Races in the persistent, barrier-synchronised kernels would show up as run-to-run differences or as a
dependence of a field's result on its position in the batch.  Every entry point must be bit-reproducible."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("res,fused,sizes", [("t30", 1, (1, 2, 3, 255, 256, 257, 455, 457, 511, 513, 1031, 2049)),   # 455 | 457: either side of the 16 MB streaming threshold
                                              ("t30", 0, (1, 7, 65, 513)),
                                              ("t63", 1, (1, 9, 65, 200, 300)),   # 300 = whole-pair walk of the inverse kernel, the others by-chunk
                                              ("t63", -1, (1, 9, 79, 80, 300)),  # auto mode: no path switch with the batch size any more (round 3)
                                              ("t63", 0, (1, 9, 65))])
def test_repeatable_and_position_independent(res, fused, sizes):
    import torch
    import speedy_f90_amd as s
    nmax = max(sizes)
    sp = s.Spectral(res, kx=8, max_batch=nmax, device=0)
    sp.set_fused(fused)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(12345)
    G = torch.from_numpy(rng.uniform(-0.5, 0.5, (nmax, sp.il, sp.ix))).to(dev)
    torch.cuda.synchronize()
    spec_ref = grid_ref = None
    for nb in sizes:
        runs = []
        for rep in range(3):
            spec = torch.full((nb, sp.nx, sp.mx), float("nan"), dtype=torch.complex128, device=dev)
            out = torch.full((nb, sp.il, sp.ix), float("nan"), dtype=torch.float64, device=dev)
            vor = torch.full((nb // 2, sp.nx, sp.mx), float("nan"), dtype=torch.complex128, device=dev)
            div = torch.full_like(vor, float("nan"))
            sp.grid_to_spec_dev(G[:nb], spec)
            sp.spec_to_grid_dev(spec, out, kcos=2)
            if nb >= 2:
                sp.vdspec_dev(G[:nb // 2], G[nb // 2:2 * (nb // 2)], vor, div, 2)
            sp.synchronize()
            runs.append((spec, out, vor, div))
        for a, b in zip(runs[0], runs[1]):
            assert torch.equal(a, b)
        for a, b in zip(runs[0], runs[2]):
            assert torch.equal(a, b)
        assert not torch.isnan(torch.view_as_real(runs[0][0])).any() and not torch.isnan(runs[0][1]).any()
        if spec_ref is None:
            spec_ref, grid_ref = runs[0][0][:1].clone(), runs[0][1][:1].clone()
        # field 0 is the same field in every batch: its result may not depend on the batch around it
        assert torch.equal(runs[0][0][:1], spec_ref) and torch.equal(runs[0][1][:1], grid_ref)
        # and the last field of the batch equals the same field transformed alone
        one_s = torch.zeros((1, sp.nx, sp.mx), dtype=torch.complex128, device=dev)
        one_g = torch.zeros((1, sp.il, sp.ix), dtype=torch.float64, device=dev)
        sp.grid_to_spec_dev(G[nb - 1:nb], one_s); sp.spec_to_grid_dev(one_s, one_g, kcos=2); sp.synchronize()
        assert torch.equal(runs[0][0][nb - 1:nb], one_s) and torch.equal(runs[0][1][nb - 1:nb], one_g)
    sp.close()


def test_wave_placement_assumption():
    """The fused T63 kernels give their Legendre (matrix-core) and FFT waves separate SIMDs by hardware wave index, assuming the
    dispatcher places the eight waves of a 512-thread workgroup round-robin -- waves w and w + 4 on one SIMD (the inverse kernel
    keeps its Legendre and FFT waves on separate SIMDs, the direct kernel pairs one of each per SIMD: both measured faster that
    way, csrc/spdy_fused_t63.inc).  spdy_wave_placement reads the SIMD id of every wave of one such workgroup per CU."""
    import speedy_f90_amd as s
    sp = s.Spectral("t63", kx=8, max_batch=4, device=0)
    for _ in range(3):
        simd, bad = sp.wave_placement()
        assert bad == 0, (simd, bad)
        assert simd[:4] == simd[4:] and sorted(simd[:4]) == [0, 1, 2, 3], simd
    sp.close()


@pytest.mark.parametrize("nb", [1, 2, 3, 16, 91, 170])
def test_t30_small_batch_forms_agree(nb, monkeypatch):
    """Small T30 inverse launches (at most a third as many tiles as CUs: every model-shaped launch, every one-field call) run by
    (tile, third of the latitudes) work items -- csrc/spdy_fused_t30.inc, PART -- instead of whole tiles.  Every (latitude,
    column) is the same chain of matrix instructions either way: the two forms must agree BIT FOR BIT, in every mode of the
    kernel: plain fields with mixed kcos, uvspec pairs, gradients, and the mixed batch of a model step (pairs + gradient +
    plain spectra from several arrays)."""
    import torch
    import speedy_f90_amd as s
    sp = s.Spectral("t30", kx=8, max_batch=max(nb, 8), device=0)
    dev = torch.device("cuda", 0)
    S = torch.from_numpy(synth.spectra(max(nb, 4), sp.trunc, first=321, full_rows=True)).to(dev)
    kc = torch.tensor([2 if b % 3 == 1 else 1 for b in range(nb)], dtype=torch.int32, device=dev)
    npair = max(1, min(nb // 2, 24))
    f64 = lambda n: torch.full((n, sp.il, sp.ix), float("nan"), dtype=torch.float64, device=dev)

    def run():
        out = {"plain": f64(nb), "ug": f64(npair), "vg": f64(npair), "gx": f64(npair), "gy": f64(npair),
               "mug": f64(npair), "mvg": f64(npair), "mpl": f64(nb), "mgx": f64(1), "mgy": f64(1)}
        sp.spec_to_grid_dev(S[:nb], out["plain"], d_kcos=kc)
        sp.uvspec_to_grid_dev(S[:npair], S[npair:2 * npair] if 2 * npair <= S.shape[0] else S[:npair], out["ug"], out["vg"], 2)
        sp.grad_to_grid_dev(S[:npair], out["gx"], out["gy"], 2)
        n0 = nb // 2
        segs = [S[:n0], S[n0:nb]] if n0 else [S[:nb]]
        sp.inverse_batch_segs_dev(S[:npair], S[1:npair + 1], out["mug"], out["mvg"], segs, out["mpl"], S[2:3], out["mgx"], out["mgy"],
                                  kcos_pairs=2, kcos=1)
        sp.synchronize()
        return out
    a = run()
    sp.set_option("t30_part", 0)          # (= a plan created under SPDY_T30_NOPART: the environment is read once per plan)
    b = run()
    sp.set_option("t30_part", 1)
    c = run()
    for k in a:
        assert not torch.isnan(a[k]).any(), k
        assert torch.equal(a[k], b[k]), (nb, k)
        assert torch.equal(a[k], c[k]), (nb, k)
    sp.close()


@pytest.mark.parametrize("nb", [1, 2, 3, 25, 73, 83, 84, 85, 128])
def test_t30_small_direct_forms_agree(nb, monkeypatch):
    """Small T30 direct launches (at most a sixth as many tiles as CUs: the T30 L8 step's launch, every one-field call) run
    with THREE workgroups per tile -- all three do the tile's row FFTs, each contracts and stores a third of the zonal
    wavenumbers (csrc/spdy_fused_t30.inc, NSPLIT).  Every coefficient is the same chain of matrix instructions either way: the
    split form, the whole-tile form (option t30_split = 0) and a large batch's persistent walk must agree BIT FOR BIT in every mode
    of the kernel: plain fields, fields with a latitude factor, the vdspec pairs, a model step's mixed direct batch."""
    import torch
    import speedy_f90_amd as s
    sp = s.Spectral("t30", kx=8, max_batch=2048, device=0)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1234)
    G = torch.from_numpy(rng.uniform(-0.5, 0.5, (2048, sp.il, sp.ix))).to(dev)
    npair = max(1, min(nb // 3, 24))
    c128 = lambda n: torch.full((n, sp.nx, sp.mx), float("nan"), dtype=torch.complex128, device=dev)

    def run():
        out = {"plain": c128(nb), "vor": c128(npair), "div": c128(npair), "mvor": c128(npair), "mdiv": c128(npair), "mpl": c128(nb)}
        sp.grid_to_spec_dev(G[:nb], out["plain"])
        sp.vdspec_dev(G[:npair], G[npair:2 * npair], out["vor"], out["div"], 2)
        sp.direct_batch_dev(G[:npair], G[npair:2 * npair], out["mvor"], out["mdiv"], G[100:100 + nb], out["mpl"], kcos=2)
        sp.synchronize()
        return out
    a = run()
    sp.set_option("t30_split", 0)
    b = run()
    sp.set_option("t30_split", 1)
    for k in a:
        assert not torch.isnan(torch.view_as_real(a[k])).any(), k
        assert torch.equal(a[k], b[k]), (nb, k)
    # ... and inside a batch large enough for the persistent walk of the throughput form
    big = c128(2048)
    sp.grid_to_spec_dev(G, big)
    sp.synchronize()
    assert torch.equal(big[:nb], a["plain"])
    sp.close()


@pytest.mark.parametrize("nb", [1, 2, 9, 73, 146, 255])
def test_t63_small_direct_forms_agree(nb, monkeypatch):
    """Small T63 direct batches (at most half as many pairs as CUs) run STAGED -- the row FFTs as a launch of their own over
    (pair, chunk, field) items, then the fused kernel's Legendre waves fed by movers (csrc/spdy_fused_t63.inc) -- instead of one
    fused launch whose eight steps are each as long as one FFT wave's phase.  Same code for every row, same accumulation order:
    the staged form, the fused split form (option t63_stage = 0) and a large batch's whole-pair walk must agree BIT FOR BIT for plain
    fields (alone and as a segment of a model step's direct batch); the vdspec pairs agree to rounding (vds in registers vs as a kernel)."""
    import torch
    import speedy_f90_amd as s
    sp = s.Spectral("t63", kx=8, max_batch=600, device=0)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(4321)
    G = torch.from_numpy(rng.uniform(-0.5, 0.5, (600, sp.il, sp.ix))).to(dev)
    npair = max(1, min(nb // 3, 48))
    c128 = lambda n: torch.full((n, sp.nx, sp.mx), float("nan"), dtype=torch.complex128, device=dev)

    def run():
        out = {"plain": c128(nb), "vor": c128(npair), "div": c128(npair), "mvor": c128(npair), "mdiv": c128(npair), "mpl": c128(nb)}
        sp.grid_to_spec_dev(G[:nb], out["plain"])
        sp.vdspec_dev(G[:npair], G[npair:2 * npair], out["vor"], out["div"], 2)
        sp.direct_batch_dev(G[:npair], G[npair:2 * npair], out["mvor"], out["mdiv"], G[100:100 + nb], out["mpl"], kcos=2)
        sp.synchronize()
        return out
    a = run()
    sp.set_option("t63_stage", 0)
    b = run()
    sp.set_option("t63_stage", 1)
    for k in a:
        assert not torch.isnan(torch.view_as_real(a[k])).any(), k
        if k in ("plain", "mpl"):
            assert torch.equal(a[k], b[k]), (nb, k)
        else:
            # the vdspec pairs: the TRANSFORMS are the same bits in both forms, but since round 6 the staged form applies vds to the
            # pair's spectra in registers (csrc/spdy_fused_t63.inc: t63_dir_writeout_vds) where the fused form runs vds_kernel
            # behind the launch -- the same expressions, contracted differently by the compiler: equal to rounding
            x, y = torch.view_as_real(a[k]), torch.view_as_real(b[k])
            assert float((x - y).abs().max()) <= 1e-13 * float(y.abs().max()), (nb, k)
    # ... and inside a batch large enough for the whole-pair walk of the throughput form
    big = c128(600)
    sp.grid_to_spec_dev(G, big)
    sp.synchronize()
    assert torch.equal(big[:nb], a["plain"])
    sp.close()


@pytest.mark.parametrize("tag,kx", [("t63", 8), ("t30", 8)])
def test_write_through_policy_same_bits(tag, kx, monkeypatch):
    """Model-sized launches with several MB of output store it write-through (sc0 sc1) instead of write-back (csrc: write_through_policy,
    option wt_min_mb): a cache policy, not arithmetic.  The by-chunk inverse launch, the grid tendencies and (T63) the staged direct
    launch must give the same bits with the policy forced on for every launch (1 MB) and switched off (0)."""
    import torch
    import speedy_f90_amd as s
    sp = s.Spectral(tag, kx=kx, max_batch=4 * kx + 4, device=0)
    sp.initialize_implicit(2400.0)
    dev = torch.device("cuda", 0)
    nb = 4 * kx + 2
    S = torch.from_numpy(synth.spectra(nb, sp.trunc, first=11, full_rows=True)).to(dev)
    rng = np.random.default_rng(99)
    px, py = (torch.from_numpy(rng.uniform(-1e-2, 1e-2, (1, sp.il, sp.ix))).to(dev) for _ in range(2))

    def run():
        f64 = lambda n: torch.full((n, sp.il, sp.ix), float("nan"), dtype=torch.float64, device=dev)
        G, U, V, PL = f64(nb), f64(3 * kx), f64(3 * kx), f64(3 * kx + 1)
        back = torch.full((3 * kx, sp.nx, sp.mx), float("nan"), dtype=torch.complex128, device=dev)
        sp.spec_to_grid_dev(S, G, kcos=1)
        g = [G[i * kx:(i + 1) * kx] for i in range(4)]
        sp.grid_tendencies_dev(g[0], g[1], g[2] + 250.0, g[3], g[0] * 1e-6, g[1].abs() * 1e-3, px, py, U, V, PL)
        sp.grid_to_spec_dev(U, back)
        sp.synchronize()
        return G, U, V, PL, back
    sp.set_option("wt_min_mb", 1)
    a = run()
    sp.set_option("wt_min_mb", 0)
    b = run()
    for i, (x, y) in enumerate(zip(a, b)):
        assert not torch.isnan(x.real if x.is_complex() else x).any(), i
        assert torch.equal(x, y), (tag, i)
    sp.close()
