#!/usr/bin/env python3
"""GPU box: the launch sequence of one T30 L8 model step's spectral side, eager versus one HIP-graph launch.

Sequence (batch sizes from SURVEY.md s3.4 / tendencies.f90:89-107, :212-234, time_stepping.f90:56-121):
  uvspec on 2*kx fields; spec_to_grid on 91 fields; vdspec on 3*kx (u,v) pairs (48 direct transforms + vds);
  grid_to_spec on 25 more fields (73 direct transforms per step); 7 horizontal diffusions over kx levels;
  implicit_terms.  The grid-space physics between the two transform batches is not part of this path.
Prints microseconds per step for both ways of launching and the reference's CPU time for the same transforms."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import speedy_f90_amd as s
import synth

kx = 8
RES = sys.argv[1] if len(sys.argv) > 1 else "t30"
sp = s.Spectral(RES, kx=kx, max_batch=128, device=0)
sp.initialize_implicit(4800.0)
dev = torch.device("cuda", 0)
c128, f64 = torch.complex128, torch.float64
S = torch.from_numpy(synth.spectra(91, sp.trunc, first=0)).to(dev)
G = torch.zeros((91, sp.il, sp.ix), dtype=f64, device=dev)
ug, vg = torch.randn((3 * kx, sp.il, sp.ix), dtype=f64, device=dev), torch.randn((3 * kx, sp.il, sp.ix), dtype=f64, device=dev)
vor, div = torch.zeros((3 * kx, sp.nx, sp.mx), dtype=c128, device=dev), torch.zeros((3 * kx, sp.nx, sp.mx), dtype=c128, device=dev)
S2 = torch.zeros((25, sp.nx, sp.mx), dtype=c128, device=dev)
u, v = torch.zeros((2 * kx, sp.nx, sp.mx), dtype=c128, device=dev), torch.zeros((2 * kx, sp.nx, sp.mx), dtype=c128, device=dev)
fld = [torch.from_numpy(synth.spectra(kx, sp.trunc, first=100 + 8 * i)).to(dev) for i in range(7)]
fdt = [torch.from_numpy(synth.spectra(kx, sp.trunc, first=200 + 8 * i)).to(dev) for i in range(7)]
out = [torch.zeros_like(f) for f in fld]
psdt = torch.from_numpy(synth.spectra(1, sp.trunc, first=300)[0]).to(dev)


ugr, vgr = torch.zeros((2 * kx, sp.il, sp.ix), dtype=f64, device=dev), torch.zeros((2 * kx, sp.il, sp.ix), dtype=f64, device=dev)
names = [("dmp", "dmp1"), ("dmpd", "dmp1d"), ("dmp", "dmp1"), ("dmp", "dmp1"), ("dmps", "dmp1s"), ("dmps", "dmp1s"), ("dmpd", "dmp1d")]


MULTI = True


def tail():
    sp.grid_to_spec_dev(G[:25], S2)
    if MULTI:
        sp.hdiff_multi_dev([(fld[i], fdt[i], a, b, out[i]) for i, (a, b) in enumerate(names)])
    else:
        for i, (a, b) in enumerate(names):
            sp.hdiff_dev(fld[i], fdt[i], a, b, out[i])
    sp.implicit_terms_dev(out[1], out[0], psdt)


def step_calls():          # the reference's call sequence, batched: uvspec, then all 91 inverse transforms, vdspec, ...
    sp.uvspec_dev(S[:2 * kx], S[2 * kx:4 * kx], u, v)
    sp.spec_to_grid_dev(S, G, kcos=1)
    sp.vdspec_dev(ug, vg, vor, div, 2)
    tail()


def step_batches():        # one launch per batch: inverse (uvspec pairs + plain), direct (vdspec pairs + plain), hdiff, implicit
    sp.inverse_batch_dev(S[:2 * kx], S[2 * kx:4 * kx], ugr, vgr, S[:59], G[:59], kcos_pairs=2, kcos=1)
    sp.direct_batch_dev(ug, vg, vor, div, G[:25], S2, 2)
    sp.hdiff_multi_dev([(fld[i], fdt[i], a, b, out[i]) for i, (a, b) in enumerate(names)])
    sp.implicit_terms_dev(out[1], out[0], psdt)


def step_fused():          # uvspec folded into its 32 inverse transforms; the other 59 as one batch
    sp.uvspec_to_grid_dev(S[:2 * kx], S[2 * kx:4 * kx], ugr, vgr, 2)
    sp.spec_to_grid_dev(S[:59], G[:59], kcos=1)
    sp.vdspec_dev(ug, vg, vor, div, 2)
    tail()


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    sp.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sp.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("%s L8 spectral-side step: 91 inverse + 73 direct transforms, uvspec, vds, 7 hdiff, implicit_terms" % RES.upper())
def step_calls_7():
    global MULTI
    MULTI = False
    step_calls()
    MULTI = True


for label, fn in (("7 hdiff launches, batched transforms", step_calls_7), ("one hdiff launch, batched transforms", step_calls),
                  ("one hdiff launch, uvspec folded into s2g", step_fused),
                  ("four launches: inverse batch, direct batch, hdiff, implicit", step_batches)):
    eager = timeit(fn)
    with sp.graph_capture() as g:
        fn()
    graph = timeit(g.launch)
    print("  %-42s eager %7.1f us   one HIP graph %7.1f us per step" % (label, eager, graph))
    g.close()
try:
    from oracle.pyoracle import Reference
    ref = Reference(RES)
    import numpy as np
    Gh = synth.grids(82, ref.ix, ref.il, first=0)       # (91 + 73) / 2 round trips
    t0 = time.perf_counter(); ref.roundtrip_loop(Gh, 5); dt = (time.perf_counter() - t0) / 5
    print("  reference, one host core, the same 164 transforms alone: %8.1f us" % (dt * 1e6))
except Exception as e:  # the checker library is optional here
    print("  (reference timing unavailable: %s)" % e)
