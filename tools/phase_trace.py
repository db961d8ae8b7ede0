#!/usr/bin/env python3
"""Debug helper (GPU box): per-phase cycle counts of the fused T30 kernels, workgroup 0.
Needs the trace build:  hipcc ... -DSPDY_PHASE_TRACE -> speedy.f90_amd/build_dbg/libspdy_trace.so"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import speedy_f90_amd as s
from importlib import import_module
lib_mod = import_module("speedy_f90_amd._lib")
lib_mod.LIB_PATH = os.path.join(ROOT, "speedy.f90_amd", "build_dbg", "libspdy_trace.so")
s.LIB_PATH = lib_mod.LIB_PATH
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
sp = s.Spectral("t30", max_batch=nb, device=0)
sp.use_torch_stream(); sp.set_fused(1)
g = torch.randn(nb, 48, 96, dtype=torch.float64, device="cuda")
sc = torch.zeros(nb, 32, 31, dtype=torch.complex128, device="cuda")
o = torch.zeros_like(g)
for _ in range(3):
    sp.grid_to_spec_dev(g, sc); sp.spec_to_grid_dev(sc, o)
torch.cuda.synchronize()
buf = np.zeros(2 * 8 * 8 * 4, np.int64)
sp.lib.spdy_debug_phase_trace(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(2, 8, 8, 4)   # kernel, tile iter, mark, wave
names = [["top", "bar0", "staged+bar", "L done", "bar", "fft done", "copyout done", "-"],
         ["top", "bar0", "staged+bar", "fft done", "final done", "bar", "L done", "prefetch issued"]]
for k, kn in enumerate(("s2g_fused", "g2s_fused")):
    print("==", kn, "(cycles since tile top, per wave; tile iterations 1..4)")
    for it in range(1, 5):
        base = t[k, it, 0].min()
        nxt = t[k, it + 1, 0].min() if it + 1 < 8 and t[k, it + 1, 0].min() > 0 else 0
        print(" tile iter %d  total %s" % (it, (nxt - base) if nxt else "?"))
        for m in range(8):
            print("   %-16s" % names[k][m], " ".join("%6d" % (v - base) for v in t[k, it, m]))
for k, kn in enumerate(("s2g_fused", "g2s_fused")):
    tops = [t[k, it, 0].min() for it in range(8) if t[k, it, 0].min() > 0]
    print(kn, "tile-to-tile (cycles):", [int(b - a) for a, b in zip(tops[:-1], tops[1:])])
    ends = [t[k, it, m][t[k, it, m] > 0].max() for it in range(8) for m in (6, 7) if (t[k, it, m] > 0).any()]
    print(kn, "first top -> last mark:", int(max(ends) - tops[0]), "cycles over", len(tops), "tiles")
