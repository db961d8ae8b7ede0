import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, speedy_f90_amd as s
nb = 6144
def run(mode):
    half = nb // 2
    g = torch.randn(nb, 48, 96, dtype=torch.float64, device="cuda")
    sc = torch.zeros(nb, 32, 31, dtype=torch.complex128, device="cuda"); o = torch.zeros_like(g)
    if mode == "one":
        sp = s.Spectral("t30", max_batch=nb, device=0); sp.use_torch_stream()
        def step():
            sp.grid_to_spec_dev(g, sc); sp.spec_to_grid_dev(sc, o)
        plans = [sp]
    else:
        os.environ["SPDY_WG_PER_CU"] = "1" if mode == "two_1wg" else "2"
        st = [torch.cuda.Stream(), torch.cuda.Stream()]
        plans = [s.Spectral("t30", max_batch=half, device=0) for _ in range(2)]
        for pl, stream in zip(plans, st):
            pl.set_fused(1)
            with torch.cuda.stream(stream): pl.use_torch_stream()
        def step():
            # stream 0: g2s(h0) s2g(h0) ... ; stream 1 offset by one kernel so opposite directions overlap
            with torch.cuda.stream(st[0]):
                plans[0].grid_to_spec_dev(g[:half], sc[:half]); plans[0].spec_to_grid_dev(sc[:half], o[:half])
            with torch.cuda.stream(st[1]):
                plans[1].grid_to_spec_dev(g[half:], sc[half:]); plans[1].spec_to_grid_dev(sc[half:], o[half:])
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, "%.2f M rt/s" % (nb * 50 / dt / 1e6), "%.1f us/step" % (dt / 50 * 1e6))
for rep in range(2):
    for m in ("one", "two_2wg", "two_1wg"): run(m)
