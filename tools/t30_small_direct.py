#!/usr/bin/env python3
"""GPU box: what the split form of small T30 direct batches (three workgroups per tile, a third of the zonal wavenumbers each)
buys.  Graph-replay time per launch of model-shaped direct batches and of the captured T30 L8 step, default vs SPDY_T30_NOSPLIT=1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth
import speedy_f90_amd as s
import bench

dev = torch.device("cuda", 0)


def measure():
    sp = s.Spectral("t30", kx=8, max_batch=256, device=0)
    sp.use_own_stream()
    out = {}
    f64 = lambda n: torch.randn((n, sp.il, sp.ix), dtype=torch.float64, device=dev)
    c128 = lambda n: torch.zeros((n, sp.nx, sp.mx), dtype=torch.complex128, device=dev)
    ug, vg, pg = f64(24), f64(24), f64(25)
    vor, div, ps = c128(24), c128(24), c128(25)
    torch.cuda.synchronize()
    for nb in (1, 2, 16, 48, 73, 128):
        G, S = f64(nb), c128(nb)
        torch.cuda.synchronize()
        out["plain_%d" % nb] = bench._time_graph_us(sp, lambda: sp.grid_to_spec_dev(G, S), per_graph=10, reps=30, warm=5)
    out["mixed_73"] = bench._time_graph_us(sp, lambda: sp.direct_batch_dev(ug, vg, vor, div, pg, ps), per_graph=10, reps=30, warm=5)
    sp.close()
    out["step_t30_l8"] = bench.dynamics_step_time(s, torch, synth, "t30", 8, dev)["us_per_step"]
    return out


a = measure()
os.environ["SPDY_T30_NOSPLIT"] = "1"
b = measure()
os.environ.pop("SPDY_T30_NOSPLIT")
c = measure()
print("%-14s %10s %10s %10s" % ("launch", "split", "whole tile", "split"))
for k in a:
    print("%-14s %10.2f %10.2f %10.2f" % (k, a[k], b[k], c[k]))
