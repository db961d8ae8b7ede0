#!/usr/bin/env python3
"""Debug helper (GPU box): s_memtime marks of block 0 of spectral_step_kernel (STEP_MARK in csrc/spdy_step.hip; needs
`make -C speedy.f90_amd trace`).  Usage: SPDY_LIB=speedy.f90_amd/build_dbg/libspdy_trace.so python tools/step_trace.py [t30|t63] [kx]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench, synth, speedy_f90_amd as s
res, kx = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("t30", 8)
print(bench.dynamics_step_time(s, torch, synth, res, kx, torch.device("cuda", 0), reps=20))
from importlib import import_module
lib = import_module("speedy_f90_amd._lib").load()
buf = np.zeros(32, np.int64)
lib.spdy_debug_step_trace.argtypes = [ctypes.c_void_p]
lib.spdy_debug_step_trace(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(2, 16)[:, :8]
t0 = t[t > 0].min()
names = ["entry", "P1 loads issued + LDS puts", "after sync", "after level loops + sync", "after per-level update (2 syncs)", "after implicit", "after hdiff", "exit"]
for w in range(2):
    print("wave", w, [int(v - t0) for v in t[w]])
print(names)
