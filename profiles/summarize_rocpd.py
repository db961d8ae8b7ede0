#!/usr/bin/env python3
"""Turn rocprofv3's rocpd sqlite output (gpurun_out/prof/*/..._results.db) into the small text
summaries committed under profiles/.  Usage: summarize_rocpd.py <results.db> [...]"""
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:70]


for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    print("# " + db)
    try:
        rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
        if rows:
            print("%-72s %6s %12s %10s %6s" % ("kernel (rocprofv3 --kernel-trace --stats)", "calls", "total_us", "avg_us", "pct"))
            for n, calls, tot, avg, pct in rows:
                print("%-72s %6d %12.1f %10.2f %6.2f" % (short(n), calls, tot, avg, pct))
    except sqlite3.Error:
        pass
    try:
        rows = list(c.execute("select kernel_name,counter_name,avg(value),min(value),max(value),count(*) "
                              "from counters_collection group by kernel_name,counter_name"))
        if rows:
            print("%-72s %-12s %14s %14s %14s %5s" % ("kernel (rocprofv3 --pmc)", "counter", "avg", "min", "max", "n"))
            for n, cn, a, lo, hi, k in rows:
                print("%-72s %-12s %14.1f %14.1f %14.1f %5d" % (short(n), cn, a, lo, hi, k))
    except sqlite3.Error:
        pass
    print()
