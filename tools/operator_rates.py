#!/usr/bin/env python3
"""GPU box: HBM rate of the spectral-space operator kernels on a large device-resident batch (T30, nb fields)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import speedy_f90_amd as s

res = sys.argv[1] if len(sys.argv) > 1 else "t30"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 6144
sp = s.Spectral(res, kx=8, max_batch=nb, device=0)
sp.use_torch_stream()
dev = torch.device("cuda", 0)
c128 = torch.complex128
A, B, C, D = (torch.randn((nb, sp.nx, sp.mx), dtype=c128, device=dev) for _ in range(4))
arr = sp.nx * sp.mx * 16


def rate(name, fn, narrays):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("%-20s %8.1f us  %7.1f GB/s (%d arrays of %d B per field)" % (name, us, narrays * arr * nb / us / 1e3, narrays, arr))


lib, h, dp = sp.lib, sp.h, sp._dp
rate("laplacian", lambda: s.check(lib.spdy_laplacian_dev(h, nb, dp(A), dp(B))), 2)
rate("inverse_laplacian", lambda: s.check(lib.spdy_inverse_laplacian_dev(h, nb, dp(A), dp(B))), 2)
rate("trunct", lambda: s.check(lib.spdy_trunct_dev(h, nb, dp(B))), 2)
rate("grad", lambda: s.check(lib.spdy_grad_dev(h, nb, dp(A), dp(B), dp(C))), 3)
rate("vds", lambda: sp.uvspec_dev(A, B, C, D) if False else s.check(lib.spdy_vds_dev(h, nb, dp(A), dp(B), dp(C), dp(D))), 4)
rate("uvspec", lambda: sp.uvspec_dev(A, B, C, D), 4)
if res == "t30":
    G1, G2 = (torch.randn((nb // 2, sp.il, sp.ix), dtype=torch.float64, device=dev) for _ in range(2))
    half = nb // 2
    def vdspec():
        sp.vdspec_dev(G1, G2, A[:half], B[:half], 2)
    for _ in range(3):
        vdspec()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        vdspec()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    byts = half * 2 * (sp.il * sp.ix * 8 + arr)
    print("%-20s %8.1f us  %7.1f GB/s (%d (u,v) pairs: 2 grids in, vor + div out)" % ("vdspec (one pass)", us, byts / us / 1e3, half))
    U1, U2 = (torch.zeros((half, sp.il, sp.ix), dtype=torch.float64, device=dev) for _ in range(2))
    def timed(name, fn, byts):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("%-34s %8.1f us  %7.1f GB/s" % (name, us, byts / us / 1e3))
    io = half * 2 * (sp.il * sp.ix * 8 + arr)
    timed("uvspec_to_grid (one pass)", lambda: sp.uvspec_to_grid_dev(A[:half], B[:half], U1, U2, 2), io)
    def two_step():
        sp.uvspec_dev(A[:half], B[:half], C[:half], D[:half])
        sp.spec_to_grid_dev(C[:half], U1, kcos=2); sp.spec_to_grid_dev(D[:half], U2, kcos=2)
    timed("uvspec + 2 x spec_to_grid", two_step, io)
    timed("grad_to_grid (one pass)", lambda: sp.grad_to_grid_dev(A[:half], U1, U2, 2), half * (2 * sp.il * sp.ix * 8 + arr))
