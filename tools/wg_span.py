#!/usr/bin/env python3
"""GPU box: entry / exit of every workgroup of the fused T30 kernels in the 100 MHz real-time counter, from a build that records
nothing else (make -C speedy.f90_amd exp EXPNAME=span EXPFLAGS=-DSPDY_WG_SPAN): how long the slowest workgroup runs, how far
the workgroups finish apart, per-XCD medians (compare the slowest exit with the launch time bench.py reports: the rest is the kernel boundary).
    SPDY_LIB=$PWD/speedy.f90_amd/build_dbg/libspdy_span.so python tools/wg_span.py [nb]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import speedy_f90_amd as s
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
sp = s.Spectral("t30", max_batch=nb, device=0)
sp.use_torch_stream(); sp.set_fused(1)
g = torch.randn(nb, 48, 96, dtype=torch.float64, device="cuda")
sc = torch.zeros(nb, 32, 31, dtype=torch.complex128, device="cuda")
o = torch.zeros_like(g)
for _ in range(5):
    sp.grid_to_spec_dev(g, sc); sp.spec_to_grid_dev(sc, o)
torch.cuda.synchronize()
span = np.zeros(2 * 512 * 2, np.int64)
sp.lib.spdy_debug_wg_span(span.ctypes.data_as(ctypes.c_void_p))
span = span.reshape(2, 512, 2)
for k, kn in enumerate(("s2g_fused", "g2s_fused")):
    v = span[k][span[k][:, 0] > 0]
    t0 = v[:, 0].min()
    ent, ex = (v[:, 0] - t0) / 100.0, (v[:, 1] - t0) / 100.0
    d = ex - ent
    q = lambda a: " ".join("%.2f" % x for x in np.percentile(a, [0, 10, 50, 90, 100]))
    print("%s: %d workgroups; us after the first entry (min p10 p50 p90 max): entry %s | exit %s" % (kn, len(v), q(ent), q(ex)))
    print("   per XCD (workgroup %% 8) median duration " + " ".join("%.1f" % np.median(d[x::8]) for x in range(8)))
