#!/usr/bin/env python3
"""GPU box: what the small-batch form of the T30 inverse kernel (work items = (tile, third of the latitudes)) buys.
Graph-replay time per launch of model-shaped inverse batches and of the captured T30 L8 step, default vs SPDY_T30_NOPART=1."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import synth
import speedy_f90_amd as s
import bench

dev = torch.device("cuda", 0)


def replay_us(sp, fn, per_graph=10, reps=30):
    return bench._time_graph_us(sp, fn, per_graph=per_graph, reps=reps, warm=5)


def measure():
    sp = s.Spectral("t30", kx=8, max_batch=256, device=0)
    sp.use_own_stream()
    out = {}
    c128 = lambda n: torch.from_numpy(synth.spectra(n, 30, first=5, full_rows=True)).to(dev)
    f64 = lambda n: torch.zeros((n, sp.il, sp.ix), dtype=torch.float64, device=dev)
    vor, div, pl = c128(8), c128(8), c128(75)
    ug, vg, pg, px, py = f64(8), f64(8), f64(75), f64(1), f64(1)
    torch.cuda.synchronize()
    for nb in (1, 2, 16, 48, 91, 170):
        S, G = c128(nb), f64(nb)
        torch.cuda.synchronize()
        out["plain_%d" % nb] = replay_us(sp, lambda: sp.spec_to_grid_dev(S, G, kcos=1))
    out["mixed_91"] = replay_us(sp, lambda: sp.inverse_batch_grad_dev(vor, div, ug, vg, pl[:74], pg[:74], pl[74:75], px, py))
    sp.close()
    out["step_t30_l8"] = bench.dynamics_step_time(s, torch, synth, "t30", 8, dev)["us_per_step"]
    return out


a = measure()
os.environ["SPDY_T30_NOPART"] = "1"
b = measure()
os.environ.pop("SPDY_T30_NOPART")
c = measure()
print("%-14s %10s %10s %10s" % ("launch", "by part", "whole tile", "by part"))
for k in a:
    print("%-14s %10.2f %10.2f %10.2f" % (k, a[k], b[k], c[k]))
