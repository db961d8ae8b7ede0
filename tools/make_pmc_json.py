#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the rocprofv3 PMC passes of tools/profile_round.sh (so that the lookup bench.py labels with
`traffic_source` is regenerated from counters, never edited by hand).

    make_pmc_json.py <out.json> <source label> <old.json or -> <results.db> [...]

Per throughput kernel of the fused path (the launches of the bench command at its batch size: B = 6144 at T30, 1536 at T63):
  bytes per field        2 x FETCH_SIZE [KB] (MI355X_MICROARCH.md: FETCH_SIZE counts 64-byte halves of the 128-byte lines
                         HBM delivers on gfx950) + WRITE_SIZE [KB], x 1024, / fields per launch
  mfma_busy_cycles       SQ_VALU_MFMA_BUSY_CYCLES per field (16 per v_mfma_f64_4x4x4_4b_f64; summed over all SIMDs) -- bench.py
                         divides by (launch duration x shader clock x 4 SIMDs x CUs) for `roofline.mfma_util`
  lds_conflict_frac      SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
The four-kernel path's entries (round-1 profiles) are carried over from the old file."""
import json
import sqlite3
import sys

KERNELS = {   # rocprofv3 kernel name prefix -> (resolution, kind, fields per launch of the bench command)
    "spdy::s2g_fused_t30_kernel<0, true, false>": ("t30", "s2g_fused", 6144),
    "spdy::g2s_fused_t30_kernel<0, true, 1>": ("t30", "g2s_fused", 6144),     # (rocprofv3 prints the defaulted NSPLIT too: the
    # two-parameter spelling matched nothing and left round 4's entry in place through round 5)
    "spdy::s2g_fused_t63_kernel<true, false, false>": ("t63", "s2g_fused", 1536),
    "spdy::g2s_fused_t63_kernel<0, true, false>": ("t63", "g2s_fused", 1536),
}


def main():
    out_path, label, old_path = sys.argv[1:4]
    avg = {}
    for db in sys.argv[4:]:
        c = sqlite3.connect(db)
        try:
            # the MEDIAN over a kernel's dispatches: the bench command also launches the same kernels at other batch sizes in its
            # side measurements (in-place round trip, B = 24,576), a handful of dispatches against hundreds at the metric's size
            vals = {}
            for name, counter, v in c.execute("select kernel_name,counter_name,value from counters_collection"):
                name = name.replace("void ", "").split("(")[0]
                for k in KERNELS:
                    if name.startswith(k):
                        vals.setdefault((k, counter), []).append(v)
            for (k, counter), v in vals.items():
                v.sort()
                avg.setdefault(k, {})[counter] = v[len(v) // 2]
        except sqlite3.Error:
            pass
    res = {"_comment": __doc__.split("\n\n")[2].strip(), "source": label, "sources": {}, "t30": {}, "t63": {}, "counters": {}}
    if old_path != "-":      # entries this run does not measure (the other resolution, the four-kernel path) are carried over
        old = json.load(open(old_path))
        for r in ("t30", "t63"):
            res[r] = dict(old.get(r, {}))
        res["counters"] = dict(old.get("counters", {}))
        res["sources"] = dict(old.get("sources", {}))
    for k, (r, kind, nb) in KERNELS.items():
        a = avg.get(k, {})
        if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
            res[r][kind] = (2.0 * a["FETCH_SIZE"] + a["WRITE_SIZE"]) * 1024.0 / nb
            res["sources"]["%s/%s" % (r, kind)] = label
        extra = {}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            extra["mfma_busy_cycles_per_field"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / nb
            extra["mfma_ops_per_field"] = a.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) / nb
        if a.get("SQ_LDS_IDX_ACTIVE"):
            extra["lds_conflict_frac"] = a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_LDS_IDX_ACTIVE"]
        if "GRBM_GUI_ACTIVE" in a:
            extra["gui_active_cycles_per_launch"] = a["GRBM_GUI_ACTIVE"]
        if extra:
            res["counters"]["%s/%s" % (r, kind)] = extra
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
