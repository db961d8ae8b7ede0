import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench, speedy_f90_amd as s
dev = torch.device('cuda', 0)
res = sys.argv[1] if len(sys.argv) > 1 else 't63'
nb = 1536 if res == 't63' else 6144
sp = s.Spectral(res, kx=8, max_batch=nb, device=0); sp.use_own_stream()
for name, g in (("randn", torch.randn((nb, sp.il, sp.ix), dtype=torch.float64, device=dev)),
                ("uniform", torch.rand((nb, sp.il, sp.ix), dtype=torch.float64, device=dev) - 0.5),
                ("zeros", torch.zeros((nb, sp.il, sp.ix), dtype=torch.float64, device=dev))):
    sc = torch.zeros((nb, sp.nx, sp.mx), dtype=torch.complex128, device=dev)
    o = torch.zeros_like(g)
    torch.cuda.synchronize()
    def rt():
        sp.grid_to_spec_dev(g, sc); sp.spec_to_grid_dev(sc, o, kcos=1)
    for reps in (30, 200):
        us = bench._time_us(torch, sp, rt, reps=reps, warm=10)
        print(res, name, reps, round(us, 1), "us", round(nb / us, 3), "M/s")
