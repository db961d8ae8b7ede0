#!/usr/bin/env python3
"""Debug helper (GPU box): s_memtime marks of workgroup 0 in the fused T63 kernels (T63_MARK in csrc/spdy_fused_t63.inc).
Needs the trace build: make -C speedy.f90_amd trace."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import speedy_f90_amd as s
from importlib import import_module
lib_mod = import_module("speedy_f90_amd._lib")
lib_mod.LIB_PATH = os.environ.get("SPDY_TRACE_LIB", os.path.join(ROOT, "speedy.f90_amd", "build_dbg", "libspdy_trace.so"))
s.LIB_PATH = lib_mod.LIB_PATH
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
sp = s.Spectral("t63", max_batch=nb, device=0)
g = torch.randn(nb, 96, 192, dtype=torch.float64, device="cuda")
sc = torch.zeros(nb, 65, 64, dtype=torch.complex128, device="cuda")
o = torch.zeros_like(g)
for _ in range(3):
    sp.grid_to_spec_dev(g, sc); sp.spec_to_grid_dev(sc, o)
torch.cuda.synchronize()
buf = np.zeros(2 * 8 * 24 * 8, np.int64)
sp.lib.spdy_debug_t63_trace(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(2, 8, 24, 8)
for k, kn in enumerate(("g2s_fused_t63", "s2g_fused_t63")):
    if not t[k].any():
        continue
    t0 = t[k][t[k] > 0].min()
    print("==", kn, "ticks since the first mark; rows = steps, per wave: marks 0..6 (-1 = not hit)")
    for wv in range(8):
        if not t[k, wv].any():
            continue
        print(" wave", wv)
        for st in range(24):
            if t[k, wv, st].any():
                print("   step %2d " % st, " ".join("%7d" % (v - t0 if v > 0 else -1) for v in t[k, wv, st, :7]))
sp.set_profiling(True)
for _ in range(5):
    sp.grid_to_spec_dev(g, sc); sp.spec_to_grid_dev(sc, o)
print({k: v[0] / max(v[1], 1) for k, v in sp.get_profile().items() if v[1]})
