#!/usr/bin/env python3
"""GPU box: T63 transform latency at model-shaped batch sizes, fused field-pair kernels vs the four-kernel path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import speedy_f90_amd as s
sp = s.Spectral("t63", kx=16, max_batch=256, device=0)
sp.use_own_stream()
for nb in (16, 32, 64, 96, 146, 256):
    g = torch.randn(nb, 96, 192, dtype=torch.float64, device="cuda")
    sc = torch.zeros(nb, 65, 64, dtype=torch.complex128, device="cuda")
    row = [nb]
    for fused in (1, 0):
        sp.set_fused(fused)
        for fn in (lambda: sp.grid_to_spec_dev(g, sc), lambda: sp.spec_to_grid_dev(sc, g)):
            for _ in range(5):
                fn()
            sp.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                fn()
            sp.synchronize()
            row.append((time.perf_counter() - t0) / 50 * 1e6)
    print("nb %4d | fused g2s %6.1f s2g %6.1f us | four-kernel g2s %6.1f s2g %6.1f us" % tuple(row))
