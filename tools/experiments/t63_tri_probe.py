"""GPU box: the three-pairs-per-workgroup T63 direct kernel (csrc/spdy_t63_tri.inc) against the pair-per-workgroup kernel --
bits and launch time at B = 1536 (and other sizes given on the command line).
usage: python tools/t63_tri_probe.py [nb ...]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speedy_f90_amd as s

sizes = [int(a) for a in sys.argv[1:]] or [6, 96, 1536]
dev = torch.device("cuda", 0)
sp = s.Spectral("t63", kx=8, max_batch=max(sizes), device=0)
rng = np.random.default_rng(5)
for nb in sizes:
    G = torch.from_numpy(rng.uniform(-0.5, 0.5, (nb, sp.il, sp.ix))).to(dev)
    out = {}
    for mode in ("0", "1"):
        os.environ["SPDY_T63_TRI"] = mode
        S = torch.full((nb, sp.nx, sp.mx), float("nan"), dtype=torch.complex128, device=dev)
        sp.grid_to_spec_dev(G, S)
        sp.synchronize()
        t = []
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                sp.grid_to_spec_dev(G, S)
            sp.synchronize()
            t.append((time.perf_counter() - t0) / 20 * 1e6)
        out[mode] = (S, min(t))
    a, b = out["0"][0], out["1"][0]
    d = (torch.view_as_real(a) - torch.view_as_real(b)).abs()
    nbad = int((d > 0).sum())
    nan = int(torch.isnan(torch.view_as_real(b)).sum())
    print(f"nb={nb}: pair kernel {out['0'][1]:.1f} us, tri kernel {out['1'][1]:.1f} us, differing values {nbad}, NaN {nan}, max|d| {float(torch.nan_to_num(d).max()):.3e}", flush=True)
    if nbad:
        bad = (d > 0).any(dim=-1)
        idx = bad.nonzero()[:8].tolist()
        print("  first differing (field, n, m):", idx)
        fields = sorted(set(int(i) for i in bad.nonzero()[:, 0].tolist()))
        print("  fields with differences:", fields[:20], "..." if len(fields) > 20 else "")
