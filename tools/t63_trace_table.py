#!/usr/bin/env python3
"""Per-step busy time of every wave from a tools/phase_trace_t63.py dump: python tools/t63_trace_table.py <file> [g2s|s2g]"""
import sys
txt = open(sys.argv[1]).read()
kern = sys.argv[2] if len(sys.argv) > 2 else "s2g"
part = txt.split("== s2g")[1] if kern == "s2g" else txt.split("== s2g")[0]
rows, wave = {}, None
for l in part.splitlines():
    if l.startswith(" wave"):
        wave = int(l.split()[1]); continue
    if l.startswith("   step"):
        rows[(wave, int(l.split()[1]))] = [int(x) for x in l.split()[2:]]
for st in range(24):
    if (0, st) not in rows: continue
    e = lambda w: "%6d" % (rows[(w, st)][4] - rows[(w, st)][0]) if (w, st) in rows else "   -  "
    nxt = rows[(0, st + 1)][0] - rows[(0, st)][0] if (0, st + 1) in rows else 0
    print("st%2d start %7d step %6d | L w0 %s w1 %s w4 %s w5 %s | F w2 %s w3 %s w6 %s w7 %s" % ((st, rows[(0, st)][0], nxt) + tuple(e(w) for w in (0, 1, 4, 5, 2, 3, 6, 7))))
