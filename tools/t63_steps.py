#!/usr/bin/env python3
"""GPU box: captured T63 step times and graph-replay time of model-sized inverse / direct launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth, bench
import speedy_f90_amd as s
dev = torch.device("cuda", 0)
sp = s.Spectral("t63", kx=16, max_batch=256, device=0); sp.use_own_stream()
for nb in (2, 16, 92, 182, 255):
    S = torch.from_numpy(synth.spectra(nb, 63, first=3, full_rows=True)).to(dev)
    G = torch.zeros((nb, sp.il, sp.ix), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    print("inverse_%d %.2f us" % (nb, bench._time_graph_us(sp, lambda: sp.spec_to_grid_dev(S, G, kcos=1), per_graph=10, reps=20, warm=3)))
sp.close()
for kx in (16, 8):
    print("step_t63_l%d %.2f us" % (kx, bench.dynamics_step_time(s, torch, synth, "t63", kx, dev)["us_per_step"]))
