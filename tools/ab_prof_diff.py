#!/usr/bin/env python3
"""Side-by-side per-kernel totals of tools/ab_prof.sh's two runs."""
import csv
import glob
import re


def load(d):
    f = glob.glob(f"gpurun_out/ab_prof/{d}/**/*kernel_stats.csv", recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        name = re.sub(r"\(NttArgs.*|\(RowsMacArgs\)|\(TensorArgs\)", "", name)[:70]
        out[name] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
    return out


a, b = load("base"), load("new")
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[1])):
    if "at::" in k or "rocclr" in k:
        continue
    ca, ta = a.get(k, (0, 0.0))
    cb, tb = b.get(k, (0, 0.0))
    print(f"{k:70s} {ca:4d} {ta:9.3f} ms | {cb:4d} {tb:9.3f} ms  {100 * (tb / ta - 1) if ta else 0:+.1f}%")
