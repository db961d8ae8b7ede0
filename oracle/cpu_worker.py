"""TEST INFRASTRUCTURE ONLY -- one CPU worker of bench.py's all-cores baseline.

    python oracle/cpu_worker.py <res> <fields> <seconds>

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
    import synth
    from oracle.pyoracle import Oracle, Reference, RESOLUTIONS
    if Reference.available(res):
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
