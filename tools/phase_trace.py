#!/usr/bin/env python3
"""Debug helper (GPU box): per-phase cycle counts of the fused T30 kernels, workgroup 0.
Needs the trace build:  make -C speedy.f90_amd trace  (-> build_dbg/libspdy_trace.so)
Marks: see the PHASE_MARK(kernel, mark) calls in csrc/spdy_fused_t30.inc"""
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
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
sp = s.Spectral("t30", max_batch=nb, device=0)
sp.use_torch_stream(); sp.set_fused(1)
g = torch.randn(nb, 48, 96, dtype=torch.float64, device="cuda")
sc = torch.zeros(nb, 32, 31, dtype=torch.complex128, device="cuda")
o = torch.zeros_like(g)
for _ in range(3):
    sp.grid_to_spec_dev(g, sc); sp.spec_to_grid_dev(sc, o)
torch.cuda.synchronize()
buf = np.zeros(2 * 8 * 8 * 8, np.int64)
sp.lib.spdy_debug_phase_trace(buf.ctypes.data_as(ctypes.c_void_p))
t = buf.reshape(2, 8, 8, 8)[..., :8]   # kernel, step, mark, wave (0-3 Legendre, 4-6 FFT)
for k, kn in enumerate(("s2g_fused", "g2s_fused")):
    print("==", kn, "(cycles since the earliest mark of the step, per wave; steps 1..4 (TRACE_TAIL=1: the last eight); -1 = mark not hit)")
    for it in (range(0, 7) if os.environ.get("TRACE_TAIL") else range(1, 5)):   # TRACE_TAIL: a -DSPDY_TRACE_TAIL build (last 8 steps)
        live = t[k, it][t[k, it] > 0]
        if not live.size:
            continue
        base = live.min()
        nl = t[k, it + 1][t[k, it + 1] > 0]
        print(" step %d  total %s" % (it, int(nl.min() - base) if nl.size else "?"))
        for m in range(8):
            if (t[k, it, m] > 0).any():
                print("   mark %d " % m, " ".join("%6d" % (v - base if v > 0 else -1) for v in t[k, it, m]))

# whole-kernel timeline (workgroup 0 and the last one): entry, top of every step, exit -- in s_memtime ticks
tl = np.zeros(2 * 2 * 8 * 32, np.int64)
sp.lib.spdy_debug_timeline(tl.ctypes.data_as(ctypes.c_void_p))
tl = tl.reshape(2, 2, 8, 32)
for k, kn in enumerate(("s2g_fused", "g2s_fused")):
    for b in range(2):
        print("== timeline", kn, "workgroup", "0" if b == 0 else "last")
        for wv, role in ((0, "Legendre wave 0"), (4, "FFT wave 4")):
            row = tl[k, b, wv]
            t0 = row[0]
            tops = [int(v - t0) for v in row[1:26] if v > 0]
            pro = [int(v - t0) if v > 0 else -1 for v in row[26:31]]
            print("   %-16s entry 0 | prologue marks %s | step tops %s | exit %d" % (role, pro, tops, int(row[31] - t0)))
# entry / exit of every workgroup of the last launch of each kernel (100 MHz counter): dispatch ramp and finishing spread
if hasattr(sp.lib, "spdy_debug_wg_span"):
    span = np.zeros(2 * 512 * 2, np.int64)
    sp.lib.spdy_debug_wg_span(span.ctypes.data_as(ctypes.c_void_p))
    span = span.reshape(2, 512, 2)
    for k, kn in enumerate(("s2g_fused", "g2s_fused")):
        v = span[k][span[k][:, 0] > 0]
        if not len(v):
            continue
        t0 = v[:, 0].min()
        ent, ex = (v[:, 0] - t0) / 100.0, (v[:, 1] - t0) / 100.0
        q = lambda a: " ".join("%.2f" % x for x in np.percentile(a, [0, 10, 50, 90, 100]))
        d = ex - ent
        print("   per XCD (workgroup %% 8): median duration " + " ".join("%.1f" % np.median(d[x::8]) for x in range(8))
              + " | slowest workgroups " + " ".join("%d(%.1f)" % (i, d[i]) for i in np.argsort(d)[-8:]))
        print("== workgroup spans %s (%d workgroups; us after the first entry; min p10 p50 p90 max): entry %s | exit %s | duration %s"
              % (kn, len(v), q(ent), q(ex), q(ex - ent)))
# the same launches timed with events, for the tick rate
sp.set_profiling(True)
for _ in range(5):
    sp.grid_to_spec_dev(g, sc); sp.spec_to_grid_dev(sc, o)
print({k: v[0] / max(v[1], 1) for k, v in sp.get_profile().items() if v[1]})
