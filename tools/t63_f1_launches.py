#!/usr/bin/env python3
"""GPU box: what row f1 looks like at T63 after round 6 -- graph nodes (launches) and replay time of the composite entry points at
model sizes: uvspec / grad -> grid, vdspec, a step's inverse and direct batch (T63 L16 shapes), with the operators evaluated
inside the transform launches (default) and as kernels in front (option t63_derive = 0; vds has no switch of its own: its
unfolded route is the fused form, option t63_stage = 0)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth
import speedy_f90_amd as s

kx = 16
sp = s.Spectral("t63", kx=kx, max_batch=4 * kx + 4, device=0)
sp.use_own_stream()
dev = torch.device("cuda", 0)
S = torch.from_numpy(synth.spectra(6 * kx + 1, 63, first=3, full_rows=True)).to(dev)
f64 = lambda n: torch.zeros((n, sp.il, sp.ix), dtype=torch.float64, device=dev)
c128 = lambda n: torch.zeros((n, sp.nx, sp.mx), dtype=torch.complex128, device=dev)
ug, vg, pg, gx, gy = f64(kx), f64(kx), f64(4 * kx), f64(1), f64(1)
U, V, PL = torch.randn_like(f64(3 * kx)), torch.randn_like(f64(3 * kx)), torch.randn_like(f64(3 * kx + 1))
vo, dv, ps = c128(3 * kx), c128(3 * kx), c128(3 * kx + 1)


def timed(fn, reps=200):
    with sp.graph_capture() as g:
        fn()
    n = g.num_nodes()
    for _ in range(10):
        g.launch()
    sp.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.launch()
    sp.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    g.close()
    return n, us


cases = {
    "uvspec_to_grid, 16 pairs": lambda: sp.uvspec_to_grid_dev(S[:kx], S[kx:2 * kx], ug, vg, 2),
    "grad_to_grid, 1 field": lambda: sp.grad_to_grid_dev(S[:1], gx, gy, 2),
    "step inverse batch (16 uv pairs + 64 plain + grad)": lambda: sp.inverse_batch_grad_dev(S[:kx], S[kx:2 * kx], ug, vg, S[2 * kx:6 * kx], pg, S[6 * kx:], gx, gy,
                                                                                              kcos_pairs=2, kcos=1, kcos_grad=2),
    "vdspec, 16 pairs": lambda: sp.vdspec_dev(U[:kx], V[:kx], vo[:kx], dv[:kx], 2),
    "step direct batch (48 uv pairs + 49 plain)": lambda: sp.direct_batch_dev(U, V, vo, dv, PL, ps, kcos=2),
}
print("%-55s %28s %28s" % ("entry point", "operators in the launch", "operator kernels apart"))
for name, fn in cases.items():
    sp.set_option("t63_derive", 1); sp.set_option("t63_stage", 1)
    a = timed(fn)
    sp.set_option("t63_derive", 0); sp.set_option("t63_stage", 0)
    b = timed(fn)
    sp.set_option("t63_derive", 1); sp.set_option("t63_stage", 1)
    print("%-55s %10d launches %8.2f us %10d launches %8.2f us" % (name, a[0], a[1], b[0], b[1]))
sp.close()
