#!/usr/bin/env python3
"""GPU box: soak test of bit-reproducibility.  For each (resolution, batch) runs `n` round trips (grid_to_spec + spec_to_grid)
on the same input and compares EVERY launch's spectra and grids with the first launch's on the device (torch.equal); prints the
number of launches that differed.  Behind the statement in DESIGN s4.3 that the product's by-SIMD role assignment never showed
the timing-dependent values of the T63_ROLE_MIX experiment.    python tools/soak_determinism.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import speedy_f90_amd as s

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
for res, nb in (("t63", 1536), ("t63", 146), ("t63", 72), ("t63", 9), ("t30", 6144), ("t30", 91), ("t30", 3)):   # 146 / 72 / 9: the staged direct form (two / one pair per contraction workgroup); t30 91 / 3: the inverse kernel by latitude thirds
    sp = s.Spectral(res, kx=8, max_batch=nb, device=0)
    sp.use_torch_stream()
    torch.manual_seed(11)
    g = torch.randn(nb, sp.il, sp.ix, dtype=torch.float64, device="cuda")
    sc = torch.zeros(nb, sp.nx, sp.mx, dtype=torch.complex128, device="cuda")
    o = torch.zeros_like(g)
    sp.grid_to_spec_dev(g, sc); sp.spec_to_grid_dev(sc, o, kcos=2)
    sc0, o0 = sc.clone(), o.clone()
    bad_s = torch.zeros((), dtype=torch.int64, device="cuda"); bad_g = torch.zeros((), dtype=torch.int64, device="cuda")
    t0 = time.time()
    for i in range(n):
        sc.zero_(); o.zero_()
        sp.grid_to_spec_dev(g, sc); sp.spec_to_grid_dev(sc, o, kcos=2)
        bad_s += (torch.view_as_real(sc) != torch.view_as_real(sc0)).any()
        bad_g += (o != o0).any()
    torch.cuda.synchronize()
    print("%s nb=%-5d %d round trips: %d launches with different spectra, %d with different grids  (%.1f s)"
          % (res, nb, n, int(bad_s), int(bad_g), time.time() - t0), flush=True)
    sp.close()

# round 6: the operators folded into the T63 model-sized launches -- uvspec / grad derived by all eight waves and staged through LDS
# (by-chunk inverse kernel), vds in the staged contraction's registers: every launch against the first
import synth
sp = s.Spectral("t63", kx=16, max_batch=80, device=0)
sp.use_torch_stream()
S = torch.from_numpy(synth.spectra(6 * 16 + 1, 63, first=3, full_rows=True)).cuda()
f64 = lambda k: torch.zeros(k, sp.il, sp.ix, dtype=torch.float64, device="cuda")
c128 = lambda k: torch.zeros(k, sp.nx, sp.mx, dtype=torch.complex128, device="cuda")
ug, vg, pg, gx, gy, vo, dv = f64(16), f64(16), f64(64), f64(1), f64(1), c128(16), c128(16)


def folded():
    sp.inverse_batch_grad_dev(S[:16], S[16:32], ug, vg, S[32:96], pg, S[96:], gx, gy, kcos_pairs=2, kcos=1, kcos_grad=2)
    sp.vdspec_dev(ug, vg, vo, dv, 2)
    return [t.clone() for t in (ug, vg, pg, gx, gy, vo, dv)]


first = folded()
bad = torch.zeros((), dtype=torch.int64, device="cuda")
t0 = time.time()
for i in range(n):
    for t in (ug, vg, pg, gx, gy):
        t.zero_()
    now = folded()
    for a, b in zip(now, first):
        bad += (torch.view_as_real(a) != torch.view_as_real(b)).any() if a.is_complex() else (a != b).any()
torch.cuda.synchronize()
print("t63 L16 step-shaped inverse batch (uvspec | grad derived on load) + vdspec (vds in the contraction), %d repeats: %d arrays that differed  (%.1f s)"
      % (n, int(bad.item()), time.time() - t0))
sp.close()

