#!/usr/bin/env python3
"""Summarises the rocprofv3 passes written by tools/pmc_ntt.sh / tools/pmc_step.sh: per kernel and per dispatch,
counter averages, HBM bytes with the gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts
64 B per 128-B request for wide streaming reads, so it is doubled; WRITE_SIZE is exact), VALU instructions per
coefficient and VALU-busy share.  Usage: pmc_summary.py <dir> <coefficients per dispatch> [kernel substring]"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"(\w+(?:<[^()]*>)?)\(", name.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else name[:80]


def main():
    out, coeffs = sys.argv[1], float(sys.argv[2])
    want = sys.argv[3] if len(sys.argv) > 3 else ""
    res = collections.defaultdict(dict)
    for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            for a, b in v.items():
                res[k][a] = b / len(n[k])
            res[k]["dispatches"] = len(n[k])
    for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            if k in res:
                res[k]["avg_ns"] = float(r["AverageNs"])
                res[k]["calls"] = int(r["Calls"])
    keep = {}
    for k, v in res.items():
        if want not in k or k.startswith("__amd") or "elementwise" in k or "distribution" in k:
            continue
        if "FETCH_SIZE" in v:
            v["hbm_read_bytes"] = v["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in v:
            v["hbm_write_bytes"] = v["WRITE_SIZE"] * 1024
        if "SQ_INSTS_VALU" in v:
            v["valu_per_coefficient"] = v["SQ_INSTS_VALU"] * 64 / coeffs
        if "SQ_ACTIVE_INST_VALU" in v and "GRBM_GUI_ACTIVE" in v:
            v["valu_busy"] = v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (v["GRBM_GUI_ACTIVE"] / 8)
        if "avg_ns" in v and "GRBM_GUI_ACTIVE" in v:
            v["clock_GHz_profiled"] = v["GRBM_GUI_ACTIVE"] / 8 / v["avg_ns"]
        if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_LDS_IDX_ACTIVE"):
            v["lds_conflict_share"] = v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]
        if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
            v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        keep[k] = v
    json.dump({"coefficients_per_dispatch": coeffs, "algorithmic_bytes_per_dispatch": 16 * coeffs, "kernels": keep},
              open(out.rstrip("/") + ".json", "w"), indent=1)
    for k, v in keep.items():
        print(k)
        for a in sorted(v):
            b = v[a]
            print(f"    {a:28s} {b:.4g}" if b < 1000 else f"    {a:28s} {b:,.0f}")


if __name__ == "__main__":
    main()
