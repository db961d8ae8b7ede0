#!/usr/bin/env python3
"""Robustness sweep (GPU box): fused kernels against the four-kernel path over random batch sizes and every composite entry
point (plain, uvspec->grid, grad->grid, vdspec, mixed inverse (+grad) / direct batches), T30 and T63.  Agreement bar 1e-13
relative to max |x| (the two paths differ by summation order only); exits non-zero on the first failure."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth, speedy_f90_amd as s

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
dev = torch.device("cuda", 0)


def rel(a, b):
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-300))


def run(sp, fused, G, S, npairs, nplain, ngrad):
    sp.set_fused(fused)
    out = {}
    n = G.shape[0]
    spec = torch.zeros((n, sp.nx, sp.mx), dtype=torch.complex128, device=dev)
    sp.grid_to_spec_dev(G, spec); out["g2s"] = spec
    grid = torch.zeros_like(G); sp.spec_to_grid_dev(S[:n], grid, kcos=2); out["s2g"] = grid
    gs = lambda m: torch.zeros((m, sp.il, sp.ix), dtype=torch.float64, device=dev)
    cs = lambda m: torch.zeros((m, sp.nx, sp.mx), dtype=torch.complex128, device=dev)
    vor, div, spl, psi = S[:npairs], S[npairs:2 * npairs], S[2 * npairs:2 * npairs + nplain], S[2 * npairs + nplain:2 * npairs + nplain + ngrad]
    a, b = gs(npairs), gs(npairs); sp.uvspec_to_grid_dev(vor, div, a, b, 2); out["uv_u"], out["uv_v"] = a, b
    a, b = gs(ngrad), gs(ngrad); sp.grad_to_grid_dev(psi, a, b, 2); out["gr_x"], out["gr_y"] = a, b
    a, b = cs(npairs), cs(npairs); sp.vdspec_dev(G[:npairs], G[npairs:2 * npairs], a, b, 2); out["vd_v"], out["vd_d"] = a, b
    a, b, c, d, e = gs(npairs), gs(npairs), gs(nplain), gs(ngrad), gs(ngrad)
    sp.inverse_batch_grad_dev(vor, div, a, b, spl, c, psi, d, e, kcos_pairs=2, kcos=1)
    out["ib_u"], out["ib_v"], out["ib_p"], out["ib_gx"], out["ib_gy"] = a, b, c, d, e
    a, b, c = cs(npairs), cs(npairs), cs(nplain)
    sp.direct_batch_dev(G[:npairs], G[npairs:2 * npairs], a, b, G[2 * npairs:2 * npairs + nplain], c, 2)
    out["db_v"], out["db_d"], out["db_p"] = a, b, c
    sp.synchronize()
    return out


worst = 0.0
for res, nmax, cases in (("t30", 1300, 14), ("t63", 330, 10)):
    sp = s.Spectral(res, kx=8, max_batch=nmax, device=0)
    for case in range(cases):
        npairs, nplain, ngrad = int(rng.integers(1, nmax // 4)), int(rng.integers(1, nmax // 3)), int(rng.integers(1, 4))
        n = 2 * npairs + nplain + ngrad
        G = torch.from_numpy(synth.grids(n, sp.ix, sp.il, first=int(rng.integers(0, 10000)))).to(dev)
        S = torch.from_numpy(synth.spectra(n, sp.trunc, first=int(rng.integers(0, 10000)), full_rows=True)).to(dev)
        a, b = run(sp, 1, G, S, npairs, nplain, ngrad), run(sp, 0, G, S, npairs, nplain, ngrad)
        for k in a:
            r = rel(a[k], b[k])
            worst = max(worst, r)
            if not r <= 1e-13:
                print("FAIL", res, "npairs", npairs, "nplain", nplain, "ngrad", ngrad, k, r)
                sys.exit(1)
        print(res, "npairs %4d nplain %4d ngrad %d ok" % (npairs, nplain, ngrad))
    sp.close()
print("all sizes agree; worst relative difference %.2e" % worst)
