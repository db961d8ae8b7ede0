#!/usr/bin/env python3
"""GPU box: per-kernel time of the fused T63 (and optionally T30) transforms for the library in $SPDY_LIB, with a CRC of the
results so that experiment builds (make -C speedy.f90_amd exp EXPNAME=x EXPFLAGS=-D...) can be compared bit for bit.
    SPDY_LIB=speedy.f90_amd/build_dbg/libspdy_x.so python tools/t63_variants.py [t63|t30] [nb] [reps]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import speedy_f90_amd as s

res = sys.argv[1] if len(sys.argv) > 1 else "t63"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else (1536 if res == "t63" else 6144)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
sp = s.Spectral(res, max_batch=nb, device=0)
sp.use_own_stream()
torch.manual_seed(7)
g = torch.randn(nb, sp.il, sp.ix, dtype=torch.float64, device="cuda")
sc = torch.zeros(nb, sp.nx, sp.mx, dtype=torch.complex128, device="cuda")
o = torch.zeros_like(g)
torch.cuda.synchronize()
for _ in range(60):
    sp.grid_to_spec_dev(g, sc); sp.spec_to_grid_dev(sc, o, kcos=1)
sp.synchronize()
best = {}
for blk in range(3):
    sp.set_profiling(True)
    for _ in range(reps):
        sp.grid_to_spec_dev(g, sc); sp.spec_to_grid_dev(sc, o, kcos=1)
    for k, v in sp.get_profile().items():
        if v[1]:
            best[k] = min(best.get(k, 1e9), v[0] / v[1] * 1e3)
    sp.set_profiling(False)
crc = lambda t: "%08x" % zlib.crc32(t.cpu().numpy().tobytes())
print("%-28s %s nb=%d  " % (os.path.basename(os.environ.get("SPDY_LIB", "libspdy.so")), res, nb)
      + "  ".join("%s %.1f us" % (k, v) for k, v in sorted(best.items()))
      + "  rt/s %.3f M  crc spec %s grid %s" % (nb / sum(best.values()), crc(sc), crc(o)))
sp.close()
