"""GPU box: the T30 round trip with separate buffers and written back in place (grid -> spec -> the same grid), for the product
library and for experiment builds (SPDY_LIB), interleaved.   usage: python tools/inplace_ab.py [t30|t63] lib1.so lib2.so ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import torch
import speedy_f90_amd as s
res = sys.argv[1]
nb = 6144 if res == "t30" else 1536
dev = torch.device("cuda", 0)
sp = s.Spectral(res, kx=8, max_batch=nb, device=0)
sp.use_own_stream()
g = torch.randn((nb, sp.il, sp.ix), dtype=torch.float64, device=dev)
o = torch.zeros_like(g)
sc = torch.zeros((nb, sp.nx, sp.mx), dtype=torch.complex128, device=dev)
def timed(fn, reps=60, warm=15):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        with sp.graph_capture() as gr:
            for _ in range(reps): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); gr.launch(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
        gr.close()
    return best
def sep():
    sp.grid_to_spec_dev(g, sc); sp.spec_to_grid_dev(sc, o, kcos=1)
def inp():
    sp.grid_to_spec_dev(g, sc); sp.spec_to_grid_dev(sc, g, kcos=1)
a = timed(sep); b = timed(inp)
print(json.dumps({"separate_M_rt_s": nb / a, "in_place_M_rt_s": nb / b, "separate_us": a, "in_place_us": b}))
''' % ROOT
res = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("t30", "t63") else "t30"
libs = [a for a in sys.argv[1:] if a.endswith(".so")] or [os.path.join(ROOT, "speedy.f90_amd", "libspdy.so")]
for rnd in range(2):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", CODE, res], capture_output=True, text=True, env=dict(os.environ, SPDY_LIB=os.path.abspath(lib)))
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print("%-44s separate %.2f M rt/s (%.1f us)   in place %.2f M rt/s (%.1f us)" % (os.path.basename(lib), d["separate_M_rt_s"], d["separate_us"], d["in_place_M_rt_s"], d["in_place_us"]), flush=True)
        except Exception:
            print(lib, "FAILED", r.stdout[-300:], r.stderr[-300:])
