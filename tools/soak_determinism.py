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
