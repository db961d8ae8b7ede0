#!/usr/bin/env python3
"""GPU box: where do two builds' T63 inverse transforms differ?  SPDY_LIB_A / SPDY_LIB_B = two libspdy builds; prints, for the
plain inverse transform of nb fields run `reps` times with build B, which (field, row, column) elements differ from build A."""
import os, sys, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import speedy_f90_amd as s
    nb, reps, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    sp = s.Spectral("t63", max_batch=nb, device=0)
    sp.use_own_stream()
    torch.manual_seed(7)
    g = torch.randn(nb, sp.il, sp.ix, dtype=torch.float64, device="cuda")
    sc = torch.zeros(nb, sp.nx, sp.mx, dtype=torch.complex128, device="cuda")
    o = torch.zeros_like(g)
    torch.cuda.synchronize()
    sp.grid_to_spec_dev(g, sc); sp.synchronize()
    res = []
    for r in range(reps):
        o.zero_(); torch.cuda.synchronize()
        for _ in range(3):
            sp.grid_to_spec_dev(g, sc)
            sp.spec_to_grid_dev(sc, o, kcos=1)
        sp.synchronize()
        res.append(o.cpu().numpy().copy())
    np.save(out, np.stack(res))
    sys.exit(0)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
reps = 4
outs = {}
for tag in ("A", "B"):
    env = dict(os.environ, SPDY_LIB=os.environ["SPDY_LIB_" + tag])
    f = "/tmp/probe_%s.npy" % tag
    subprocess.check_call([sys.executable, __file__, "child", str(nb), str(reps), f], env=env)
    outs[tag] = np.load(f)
ref = outs["A"][0]
print("A self-consistent:", all(np.array_equal(ref, x) for x in outs["A"]))
for r in range(reps):
    d = outs["B"][r] != ref
    print("B run %d: %d differing values" % (r, int(d.sum())))
    if d.any():
        f, row, col = np.nonzero(d)
        print("   fields", np.unique(f)[:20], "n =", np.unique(f).size)
        print("   rows", np.unique(row)[:48], "n =", np.unique(row).size)
        print("   cols", np.unique(col)[:32], "..", np.unique(col)[-8:], "n =", np.unique(col).size)
        e = np.abs(outs["B"][r] - ref)[d]
        print("   |diff| max %.3e median %.3e  (|ref| max %.3e)" % (e.max(), np.median(e), np.abs(ref).max()))
