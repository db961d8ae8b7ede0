#!/usr/bin/env python3
"""GPU box: rate of the HOST-pointer drop-in path (PCIe copies + sync included), for DESIGN.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, synth, speedy_f90_amd as s
sp = s.Spectral("t30", kx=8, max_batch=6144, device=0)
for nb in (1, 8, 48, 512, 6144):
    G = synth.grids(min(nb, 64), 96, 48); G = np.tile(G, ((nb + 63) // 64, 1, 1))[:nb].copy()
    sp.spec_to_grid(sp.grid_to_spec(G))
    reps = max(3, 2000 // nb)
    t0 = time.perf_counter()
    for _ in range(reps):
        S = sp.grid_to_spec(G); G2 = sp.spec_to_grid(S, 1)
    dt = time.perf_counter() - t0
    print("host-pointer path  nb=%5d : %10.0f round trips/s  (%.1f us per call pair)" % (nb, nb * reps / dt, dt / reps * 1e6))
