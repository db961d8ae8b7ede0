"""One captured dynamical-core step (bench.py's extras) on its own, for rocprofv3 --kernel-trace: per-kernel durations inside the graph."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench, synth, speedy_f90_amd as s
res, kx = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("t63", 16)
print(bench.dynamics_step_time(s, torch, synth, res, kx, torch.device("cuda", 0), reps=200))
