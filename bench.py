#!/usr/bin/env python3
"""Headline benchmark: grid<->spectral round trips per second at T30 (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--res t30|t63] [--batch B]

One "step" = one pass of the hot path over one device-resident batch of B synthetic 2-D
fields: grid_to_spec followed by spec_to_grid(.,kcos=1) (spectral.f90:98-122), FP64.
Workload at N=1: BASELINE.json configs[1] -- T30 L8 fields, B = 6144 per GPU (SURVEY.md s8d:
~226 MB of grid data, far beyond L2/Infinity Cache so HBM is really exercised).
N > 1 (torchrun, one rank per GPU): the batch index (field x level) is sharded, every rank
transforms its own B fields, no data-path collective -> weak scaling; value = all ranks' round
trips / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline     : dominant kernel's algorithmic bytes per launch / its HIP-event launch time
  cpu_baseline : the reference's own CPU path (oracle/_ref, flang build) -- or the C port if
                 that is absent -- timed on ONE host core on a bounded sample of the same fields
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def algorithmic_bytes(sp):
    """Bytes each kernel kind must move per field (no tables: amortised over the batch)."""
    spec = sp.mx * sp.nx * 16
    grid = sp.ix * sp.il * 8
    four = 2 * sp.mx * sp.il * 8
    return {"legendre_inv": spec + four, "fourier_inv": four + grid,
            "fourier_dir": grid + four, "legendre_dir": four + spec,
            "s2g_fused": spec + grid, "g2s_fused": grid + spec,
            "round_trip": 2 * (spec + grid)}


def cpu_baseline(res, sample_fields=256, target_s=10.0):
    """Reference CPU path on one host core, bounded to ~target_s seconds."""
    import synth
    from oracle.pyoracle import Oracle, Reference, RESOLUTIONS
    if Reference.available(res):
        impl, kind = Reference(res), "reference"
    else:
        impl, kind = Oracle(*RESOLUTIONS[res]), "port"
    G = synth.grids(sample_fields, impl.ix, impl.il, first=0)
    t0 = time.perf_counter()
    impl.roundtrip_loop(G, 1)
    one = time.perf_counter() - t0
    nrep = max(1, int(target_s / max(one, 1e-6)))
    t0 = time.perf_counter()
    impl.roundtrip_loop(G, nrep)
    dt = time.perf_counter() - t0
    return {"value": sample_fields * nrep / dt, "unit": "round trips/s", "cores": 1, "kind": kind,
            "sample": "%d passes over %d synthetic %s fields (grid_to_spec + spec_to_grid, one field at a "
                      "time), %.1f s on one core of %d" % (nrep, sample_fields, res.upper(), dt, os.cpu_count() or 1)}


def cpu_baseline_all_cores(res, fields=64, target_s=4.0):
    """The same reference loop on every host core at once: one PROCESS per core (oracle/cpu_worker.py), each
    transforming its own fields one at a time for ~target_s seconds.  The reference itself is single-threaded;
    this is the generous 'whole host' number next to the single-core one."""
    import subprocess
    nproc = os.cpu_count() or 1
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), res, str(fields), str(target_s)]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for _ in range(nproc)]
    total, rate, kind, ok = 0, 0.0, "reference", 0
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=60 + 10 * target_s)
            n, dt, kind = out.split()
            total += int(n)
            rate += int(n) / float(dt)
            ok += 1
        except Exception:
            pr.kill()
    wall = time.perf_counter() - t0
    return {"value": rate, "unit": "round trips/s", "cores": ok, "kind": kind,
            "sample": "%d processes (one per logical core), each %.0f s over its own %d synthetic %s fields; sum of the "
                      "per-process rates; %.1f s wall incl. start-up" % (ok, target_s, fields, res.upper(), wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--res", default="t30", choices=["t30", "t63"])
    ap.add_argument("--batch", type=int, default=0, help="fields per GPU (default 6144 at T30, 1536 at T63)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fused", type=int, default=-1, help="1 fused single-pass kernels, 0 four-kernel path, -1 auto")
    args = ap.parse_args()

    import torch
    import speedy_f90_amd as s
    import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    nb = args.batch or (6144 if args.res == "t30" else 1536)
    sp = s.Spectral(args.res, kx=8, max_batch=nb, device=local)
    # the plan keeps its own (non-default) stream; torch.cuda.synchronize() below covers every stream of the device
    sp.use_own_stream()
    sp.set_fused(args.fused)

    # synthetic white-noise grids: 64 seeded templates tiled and rescaled per field so
    # that every field of the batch is distinct; each rank owns its own shard of the batch index
    uniq = 64
    tmpl = torch.from_numpy(synth.grids(uniq, sp.ix, sp.il, first=rank * uniq)).to(dev)
    reps = (nb + uniq - 1) // uniq
    grid = tmpl.repeat(reps, 1, 1)[:nb].contiguous()
    grid *= (1.0 + torch.arange(nb, dtype=torch.float64, device=dev).view(nb, 1, 1) / nb)
    spec = torch.zeros((nb, sp.nx, sp.mx), dtype=torch.complex128, device=dev)
    out = torch.zeros_like(grid)

    def step():
        sp.grid_to_spec_dev(grid, spec)
        sp.spec_to_grid_dev(spec, out, kcos=1)

    torch.cuda.synchronize()             # the inputs were produced on torch's stream; the plan runs on its own
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = s.sharding.max_over_ranks(max(wall, e0.elapsed_time(e1) / 1e3), dev)

    # per-kernel launch durations: HIP events on the launch stream, outside the timed region
    sp.set_profiling(True)
    prof_steps = max(5, min(20, args.steps))
    for _ in range(prof_steps):
        step()
    prof = sp.get_profile()
    sp.set_profiling(False)

    # N > 1 only, outside the timed region and best-effort: the one exchange a level-sharded model step needs --
    # all-gather of the implicit solve's inputs over RCCL -- timed on its own so the scaling of the
    # transform metric can be read "with and without the gather"
    gather_ms = None
    if dist and args.res == "t30":
        try:
            lo, hi = s.sharding.shard_range(sp.kx, rank, world)
            loc = torch.zeros((hi - lo, sp.nx, sp.mx), dtype=torch.complex128, device=dev)
            for _ in range(5):
                s.sharding.allgather_levels(loc, sp.kx); s.sharding.allgather_levels(loc, sp.kx)
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            for _ in range(50):
                s.sharding.allgather_levels(loc, sp.kx); s.sharding.allgather_levels(loc, sp.kx)   # divdt and tdt
            torch.cuda.synchronize()
            gather_ms = s.sharding.max_over_ranks((time.perf_counter() - t0) / 50 * 1e3, dev)
        except Exception as e:  # never let the optional measurement break the bench line
            gather_ms = None
            if rank == 0:
                print("implicit all-gather timing skipped: %s" % e, file=sys.stderr)

    if rank == 0:
        ab = algorithmic_bytes(sp)
        value = world * nb * args.steps / elapsed
        kinds = {k: (ms / max(cnt, 1)) for k, (ms, cnt) in prof.items() if cnt}
        dom = max(kinds, key=kinds.get)
        dom_ms = kinds[dom]
        achieved = ab[dom] * nb / (dom_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                per_field = json.load(open(tpath)).get(args.res, {}).get(dom)
                traffic = per_field * nb if per_field else None
            except Exception:
                traffic = None
        res = {
            "metric": "spectral transforms/sec (grid<->spec round-trip) at %s L8" % args.res.upper(),
            "value": value, "unit": "round trips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s (%dx%d grid, trunc %d) device-resident batch of %d 2-D fields per GPU "
                                   "(field x level index sharded over ranks, no collective); one step = "
                                   "grid_to_spec + spec_to_grid(kcos=1) over the batch"
                                   % (args.res.upper(), sp.ix, sp.il, sp.trunc, nb),
                       "fields_per_gpu": nb, "resolution": args.res, "parallelism": "batch-shard x%d" % world},
            "transforms_per_s": 2 * value,
            "path_hbm_frac": value / world * ab["round_trip"] / (HBM_PEAK_GBS * 1e9),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launch_ms": dom_ms, "bytes_per_launch": ab[dom] * nb,
                         "all_kernels_ms": kinds},
        }
        if world > 1:
            res["implicit_allgather_ms"] = gather_ms   # 2 x all-gather of [kx, nx, mx] complex level slabs, per model step
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.res)
            res["gpu_over_cpu_core"] = value / res["cpu_baseline"]["value"]
            res["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.res)
            res["gpu_over_cpu_all_cores"] = value / res["cpu_baseline_all_cores"]["value"]
        print(json.dumps(res))
    sp.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
