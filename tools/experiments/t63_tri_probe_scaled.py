"""GPU box: scaled (vdspec) direct transforms, three-pair kernel vs pair kernel: where do they differ?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speedy_f90_amd as s
dev = torch.device("cuda", 0)
sp = s.Spectral("t63", kx=8, max_batch=64, device=0)
rng = np.random.default_rng(977)
G = torch.from_numpy(rng.uniform(-0.5, 0.5, (64, sp.il, sp.ix))).to(dev)
for npair in (1, 2, 3, 6):
    out = {}
    for mode in ("0", "1"):
        os.environ["SPDY_T63_TRI"] = mode
        v = torch.full((npair, sp.nx, sp.mx), float("nan"), dtype=torch.complex128, device=dev)
        d = torch.full((npair, sp.nx, sp.mx), float("nan"), dtype=torch.complex128, device=dev)
        sp.vdspec_dev(G[:npair], G[npair:2 * npair], v, d, 2)
        sp.synchronize()
        out[mode] = (v.cpu().numpy(), d.cpu().numpy())
    for name, i in (("vor", 0), ("div", 1)):
        a, b = out["0"][i], out["1"][i]
        bad = np.argwhere(a != b)
        print(f"npair={npair} {name}: {len(bad)} differing coefficients", bad[:6].tolist(), "max|d|", float(np.abs(a - b).max()))
