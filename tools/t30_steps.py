import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth, bench
import speedy_f90_amd as s
dev = torch.device("cuda", 0)
for kx in (16, 8):
    print("step_t30_l%d %.2f us" % (kx, bench.dynamics_step_time(s, torch, synth, "t30", kx, dev)["us_per_step"]))
