#!/usr/bin/env python3
"""Where does a kernel spill?  Compiles csrc/<file>.hip to gfx950 assembly and lists, per basic block of the kernels whose
mangled name contains <pattern>, the scratch stores/loads next to the block's MFMA / LDS / global-load / FP64 counts.
    python tools/spill_map.py spdy_kernels g2s_fused_t63_kernelILi0 [extra hipcc flags...]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
out = "/tmp/spill_map_%s.s" % src
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", out,
                       os.path.join(root, "speedy.f90_amd", "csrc", src + ".hip")] + extra, stderr=subprocess.DEVNULL)
t = open(out).read()
for m in re.finditer(r"^(_Z\w*%s\w*):" % re.escape(pat), t, re.M):
    a = m.start(); b = t.index(".Lfunc_end", a)
    print("==", m.group(1))
    blk, stats, order = "entry", {}, []
    for ln in t[a:b].split("\n"):
        mm = re.match(r"^(\.LBB\d+_\d+):", ln)
        if mm or blk not in stats:
            if mm: blk = mm.group(1)
            stats[blk] = dict(n=0, sst=0, sld=0, mfma=0, ds=0, gld=0, gst=0, f64=0); order.append(blk)
            if mm: continue
        s = stats[blk]; s["n"] += 1
        s["sst"] += "scratch_store" in ln; s["sld"] += "scratch_load" in ln; s["mfma"] += "v_mfma" in ln
        s["ds"] += bool(re.search(r"\bds_", ln)); s["gld"] += "global_load" in ln; s["gst"] += "global_store" in ln
        s["f64"] += bool(re.search(r"v_(fma|add|mul)_f64", ln))
    small = dict(n=0, sst=0, sld=0)
    for k in order:
        s = stats[k]
        if s["n"] >= 60: print("  %-12s" % k, s)
        else:
            small["n"] += 1; small["sst"] += s["sst"]; small["sld"] += s["sld"]
    print("  %d small blocks: scratch stores %d, loads %d" % (small["n"], small["sst"], small["sld"]))
