import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, speedy_f90_amd as s
for res, nb in (("t30", 6143), ("t63", 777)):
    sp = s.Spectral(res, kx=8, max_batch=nb, device=0)
    dev = torch.device("cuda", 0)
    g = torch.rand((nb, sp.il, sp.ix), dtype=torch.float64, device=dev) - 0.5
    spec = torch.zeros((nb, sp.nx, sp.mx), dtype=torch.complex128, device=dev)
    out = torch.zeros_like(g)
    vor = torch.zeros((nb // 2, sp.nx, sp.mx), dtype=torch.complex128, device=dev); div = torch.zeros_like(vor)
    torch.cuda.synchronize()
    ref = None; bad = 0; t0 = time.time(); n = 0
    while time.time() - t0 < 40:
        for _ in range(20):
            sp.grid_to_spec_dev(g, spec); sp.spec_to_grid_dev(spec, out, kcos=2)
            sp.vdspec_dev(g[:nb // 2], out[:nb // 2], vor, div, 2)
            n += 1
        sp.synchronize()
        cur = (spec.clone(), out.clone(), vor.clone(), div.clone())
        if ref is None: ref = cur
        else: bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(ref, cur))
    print(res, "iterations", n, "mismatching snapshots", bad)
    sp.close()
