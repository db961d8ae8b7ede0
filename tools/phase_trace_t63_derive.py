#!/usr/bin/env python3
"""Debug helper (GPU box): s_memtime marks of workgroup 0 of the by-chunk T63 inverse kernel with its operands DERIVED on load
(uvspec folded into the operand load, csrc/spdy_fused_t63.inc: t63_inv_load_b_op) and with plain operands.
Needs the trace build: make -C speedy.f90_amd trace.   usage: phase_trace_t63_derive.py [nb]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import synth
import speedy_f90_amd as s
from importlib import import_module
lib_mod = import_module("speedy_f90_amd._lib")
lib_mod.LIB_PATH = os.environ.get("SPDY_TRACE_LIB", os.path.join(ROOT, "speedy.f90_amd", "build_dbg", "libspdy_trace.so"))
s.LIB_PATH = lib_mod.LIB_PATH
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sp = s.Spectral("t63", max_batch=max(nb, 8), device=0)
S = torch.from_numpy(synth.spectra(2 * nb, 63, first=1, full_rows=True)).cuda()
ug, vg = (torch.zeros(nb, 96, 192, dtype=torch.float64, device="cuda") for _ in range(2))
for derive in (1, 0):
    sp.set_option("t63_derive", derive)
    for _ in range(3):
        sp.uvspec_to_grid_dev(S[:nb], S[nb:], ug, vg, 2)
    torch.cuda.synchronize()
    buf = np.zeros(2 * 8 * 24 * 8, np.int64)
    sp.lib.spdy_debug_t63_trace(buf.ctypes.data_as(ctypes.c_void_p))
    t = buf.reshape(2, 8, 24, 8)[1]
    t0 = t[t > 0].min()
    print("== s2g_fused_t63 by-chunk, derive =", derive, ": ticks since the first mark; step 23 = [entry, operands loaded, first barrier passed, (derived) second half of the loader starts]")
    for wv in range(8):
        if not t[wv].any():
            continue
        print(" wave", wv)
        for st in list(range(8)) + [23]:
            if t[wv, st].any():
                print("   step %2d " % st, " ".join("%7d" % (v - t0 if v > 0 else -1) for v in t[wv, st, :7]))
