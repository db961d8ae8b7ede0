"""TEST INFRASTRUCTURE ONLY -- one CPU worker of bench.py's all-cores baseline.

    python oracle/cpu_worker.py <res> <fields> <seconds> [<variant> [<cpu>]]

<variant>: "" (flang -O2 build, the parity oracle) or "fast" (-O3 -ffast-math -march=x86-64-v3, mirroring upstream's
-Ofast, gfortran.makefile:18); <cpu>: pin this worker to one logical CPU.

Runs the reference's grid_to_spec + spec_to_grid loop (oracle/_ref, or the C restatement when that is
absent) over <fields> synthetic fields, one field at a time, for about <seconds> of wall time and prints
"<round trips> <elapsed seconds> <kind>".  Separate processes, not threads: the Fortran runtime serialises
concurrent callers on its array-temporary allocator.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))


def main():
    res, nf, secs = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    variant = sys.argv[4] if len(sys.argv) > 4 else ""
    if len(sys.argv) > 5:
        os.sched_setaffinity(0, {int(sys.argv[5])})
    import synth
    from oracle.pyoracle import Oracle, Reference, RESOLUTIONS
    if Reference.available(res + variant):
        impl, kind = Reference(res + variant), "reference"
    elif Reference.available(res):
        impl, kind = Reference(res), "reference"
    else:
        impl, kind = Oracle(*RESOLUTIONS[res]), "port"
    G = synth.grids(nf, impl.ix, impl.il, first=0)
    impl.roundtrip_loop(G, 1)                      # warm-up (tables, caches)
    t0 = time.perf_counter()
    impl.roundtrip_loop(G, 1)
    one = max(time.perf_counter() - t0, 1e-6)
    chunk = max(1, int(0.5 / one))                 # ~0.5 s per call
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        impl.roundtrip_loop(G, chunk)
        done += chunk * nf
    print(done, time.perf_counter() - t0, kind)


if __name__ == "__main__":
    main()
