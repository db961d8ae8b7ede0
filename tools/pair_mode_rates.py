import sys, json, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench, speedy_f90_amd as s
dev = torch.device('cuda', 0)
sp = s.Spectral('t30', kx=8, max_batch=6144, device=0); sp.use_own_stream()
n = 3072
f64 = lambda n: torch.randn((n, sp.il, sp.ix), dtype=torch.float64, device=dev)
c128 = lambda n: torch.zeros((n, sp.nx, sp.mx), dtype=torch.complex128, device=dev)
ug, vg, vor, div = f64(n), f64(n), c128(n), c128(n)
g6, s6 = f64(6144), c128(6144)
torch.cuda.synchronize()
print('vdspec', bench._time_us(torch, sp, lambda: sp.vdspec_dev(ug, vg, vor, div, 2), reps=50, warm=10))
print('g2s plain', bench._time_us(torch, sp, lambda: sp.grid_to_spec_dev(g6, s6), reps=50, warm=10))
print('uvspec_to_grid', bench._time_us(torch, sp, lambda: sp.uvspec_to_grid_dev(vor, div, ug, vg, 2), reps=50, warm=10))
print('s2g plain', bench._time_us(torch, sp, lambda: sp.spec_to_grid_dev(s6, g6, kcos=1), reps=50, warm=10))
