#!/usr/bin/env python3
"""GPU box: measured per-array errors of the captured dynamical-core step (tests/test_gpu_step.py::test_dynamical_core_step_graph)
against the oracle's call-by-call sequence, after each of two chained steps: max|x - ref| / max|ref| over the whole array, and the
same with the global mean (coefficient (m', n) = (0, 1)) removed from both -- the norm that the 353 K mean of t cannot carry.
    python tools/step_error_budget.py [t30 t63k16 ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import synth
import conftest
import test_gpu_step as T

tags = sys.argv[1:] or ["t30", "t63k16"]
out = {}
for tag in tags:
    kx = conftest.VARIANTS[tag][3]
    sp = T.make_plan(tag, 4 * kx + 4)
    from oracle.pyoracle import Oracle, build
    build()
    o = Oracle(*conftest.VARIANTS[tag])
    if tag in synth.SIGMA_SETS:
        o.set_sigma(synth.SIGMA_SETS[tag])
    res = T.run_dynamical_core_steps(sp, o, tag, one_launch_tail=False, nsteps=2)
    out[tag] = res
    for k, v in res.items():
        print(tag, k, " ".join("%s=%.2e/%.2e" % (n, e[0], e[1]) for n, e in v.items()))
    sp.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "step_error_budget.json"), "w"), indent=1)
