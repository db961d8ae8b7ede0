"""GPU box experiment: the T30 round trip over 6144 fields as ONE chain (plan A, whole batch) against TWO independent chains of
3072 fields each on two plans / two streams (the hardware may backfill one chain's draining kernel with the other chain's next).
usage: python tools/two_stream_halves.py [t30|t63] [nchains]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speedy_f90_amd as s
res = sys.argv[1] if len(sys.argv) > 1 else "t30"
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nb = 6144 if res == "t30" else 1536
dev = torch.device("cuda", 0)
plans = [s.Spectral(res, kx=8, max_batch=nb, device=0) for _ in range(nch)]
for p in plans:
    p.use_own_stream()
sp = plans[0]
g = torch.randn((nb, sp.il, sp.ix), dtype=torch.float64, device=dev)
o = torch.zeros_like(g)
sc = torch.zeros((nb, sp.nx, sp.mx), dtype=torch.complex128, device=dev)
torch.cuda.synchronize()
K = 40
def whole():
    for _ in range(K):
        sp.grid_to_spec_dev(g, sc); sp.spec_to_grid_dev(sc, o, kcos=1)
def chains():
    h = nb // nch
    for _ in range(K):
        for c, p in enumerate(plans):
            p.grid_to_spec_dev(g[c * h:(c + 1) * h], sc[c * h:(c + 1) * h])
        for c, p in enumerate(plans):
            p.spec_to_grid_dev(sc[c * h:(c + 1) * h], o[c * h:(c + 1) * h], kcos=1)
for name, fn in (("one chain", whole), ("%d chains" % nch, chains), ("one chain", whole), ("%d chains" % nch, chains)):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K * 1e6)
    print("%-10s %.1f us per step = %.2f M round trips/s" % (name, best, nb / best), flush=True)
