#!/usr/bin/env python3
"""Per-kernel summary of tools/prof_step_pmc.sh's pass: launches, VALU instructions per launch (wave-instructions),
VALU-busy share.  Usage: python3 tools/pmc_step_summary.py gpurun_out/pmc_step_new [more dirs ...]"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+(?:<[^()]*>)?)\(", name)
    return m.group(1) if m else name[:60]


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob(d + "/sq1/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES":
                cnt[k] += 1
    return agg, cnt


runs = [load(d) for d in sys.argv[1:]]
keys = [k for k in runs[0][0] if not k.startswith("__amd") and "elementwise" not in k and "distribution" not in k and "reduce_kernel" not in k]
keys.sort(key=lambda k: -runs[0][0][k].get("GRBM_GUI_ACTIVE", 0))
for k in keys:
    row = f"{k:44s}"
    for agg, cnt in runs:
        v, n = agg.get(k, {}), max(cnt.get(k, 0), 1)
        g = v.get("GRBM_GUI_ACTIVE", 1)
        row += f" | {cnt.get(k, 0):3d} x {v.get('SQ_INSTS_VALU', 0) / n:11.4e} VALU, busy {v.get('SQ_ACTIVE_INST_VALU', 0) * 4 / 1024 / (g / 8):.3f}"
    print(row)
