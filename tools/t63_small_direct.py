#!/usr/bin/env python3
"""GPU box: staged vs fused form of small T63 direct batches (graph-replay time per launch) and the captured T63 steps."""
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
    sp = s.Spectral("t63", kx=16, max_batch=256, device=0)
    sp.use_own_stream()
    out = {}
    for nb in (2, 16, 73, 146, 255):
        G = torch.randn((nb, sp.il, sp.ix), dtype=torch.float64, device=dev)
        S = torch.zeros((nb, sp.nx, sp.mx), dtype=torch.complex128, device=dev)
        torch.cuda.synchronize()
        out["direct_%d" % nb] = bench._time_graph_us(sp, lambda: sp.grid_to_spec_dev(G, S), per_graph=10, reps=20, warm=3)
    sp.close()
    out["step_t63_l16"] = bench.dynamics_step_time(s, torch, synth, "t63", 16, dev)["us_per_step"]
    out["step_t63_l8"] = bench.dynamics_step_time(s, torch, synth, "t63", 8, dev)["us_per_step"]
    return out


a = measure()
os.environ["SPDY_T63_NOSTAGE"] = "1"
b = measure()
os.environ.pop("SPDY_T63_NOSTAGE")
c = measure()
print("%-14s %10s %10s %10s" % ("launch", "staged", "fused", "staged"))
for k in a:
    print("%-14s %10.2f %10.2f %10.2f" % (k, a[k], b[k], c[k]))
