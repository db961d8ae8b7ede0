#!/usr/bin/env python3
"""Debug helper (GPU box): s_memtime marks of workgroup 0 in the three-pair T63 direct kernel (T63_MARK in csrc/spdy_t63_tri.inc).
Marks per step: 0 step start, 1 rows landed (vmcnt), 2 rows in registers, 3 DMA + zero fill issued, 4 sub-transforms + twiddles +
lane transposes done, 5 past barrier A, 6 Fourier rows written + barrier B, 7 contraction done.
Needs the trace build: make -C speedy.f90_amd trace.   usage: phase_trace_tri.py [nb]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SPDY_T63_TRI"] = "1"
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
for _ in range(3):
    sp.grid_to_spec_dev(g, sc)
torch.cuda.synchronize()
buf = np.zeros(2 * 8 * 24 * 8, np.int64)
sp.lib.spdy_debug_t63_trace(buf.ctypes.data_as(ctypes.c_void_p))
t1 = buf.reshape(2, 8, 24, 8)[1]
t = buf.reshape(2, 8, 24, 8)[0]
t0 = t[t > 0].min()
print("nb =", nb, ": ticks; per wave and step: start, then the length of each interval between consecutive marks 0..7, then the step")
for wv in range(4):
    print(" wave", wv)
    for st in range(24):
        if t[wv, st].any():
            m = t[wv, st]
            d = [int(m[i + 1] - m[i]) for i in range(7)]
            nxt = t[wv, st + 1, 0] if st + 1 < 24 and t[wv, st + 1, 0] > 0 else 0
            c = t1[wv, st]
            cd = ([int(c[0] - m[6])] + [int(c[i + 1] - c[i]) for i in range(3)] + [int(m[7] - c[3])]) if c[:4].all() else []
            print("   step %2d  %8d | " % (st, m[0] - t0) + " ".join("%6d" % v for v in d) + " | %6d" % ((nxt - m[0]) if nxt else (m[7] - m[0])) + "   contract: " + " ".join("%5d" % v for v in cd))
