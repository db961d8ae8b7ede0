#!/usr/bin/env python3
"""Prints the numbers of a bench.py JSON line that the round's notes quote (headline, kernels, model-shaped launches, steps)."""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("value %.3f M rt/s  ms/step %.4f  kernels(ms) %s  frac %.3f  mfma_util %s" % (
    d["value"] / 1e6, d["ms_per_step"], {k: round(v, 4) for k, v in r["all_kernels_ms"].items()}, r["frac"], r.get("mfma_util")))
e = d.get("extras", {})
for k in ("vdspec_one_pass", "uvspec_to_grid", "inverse_batch_6144", "direct_batch_6144", "inverse_91", "inverse_48", "direct_73",
          "dynamics_step_t30_l8", "dynamics_step_t63_l16", "round_trip_b24576", "round_trip_in_place"):
    if k in e:
        print(" ", k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in e[k].items() if isinstance(b, (int, float))})
for k in ("t63_round_trip", "t30_round_trip"):
    if k in e:
        x = e[k]
        print(" ", k, round(x["round_trips_per_s"] / 1e6, 3), "M eager;", round(x.get("round_trips_per_s_replayed", 0) / 1e6, 3), "M replayed;",
              {a: round(b, 2) for a, b in x["kernel_us"].items()})
for k in ("fortran_step_loop", "host_pointer_dropin"):
    if k in e:
        print(" ", k, json.dumps(e[k])[:700])
if "multi_gpu" in d:
    print("  multi_gpu", json.dumps(d["multi_gpu"])[:900])
if "cpu_baseline" in d:
    print("  cpu", d["cpu_baseline"]["value"], d.get("cpu_baseline_socket", {}).get("value"), d.get("gpu_over_cpu_socket"))
