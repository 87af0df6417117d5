#!/usr/bin/env python3
"""Copies what tools/r03/final.sh left under gpurun_out/ into profiles/ (run here, after the GPU call): the bench line (with
the PMC traffic of the same run's counter passes), kernel stats, per-config lines and stats, the curated PMC record that
bench.py reads (profiles/ntt_pmc_traffic.json keeps its older rounds' sections) and the raw counter dumps."""
import json
import os
import shutil

S = "gpurun_out/r03final"
cur = json.load(open("profiles/ntt_pmc_traffic.json"))
raw = json.load(open("gpurun_out/pmc_r3final.json"))
k = raw["kernels"]
fc = [n for n in k if n.startswith("ntt_fwd_cols")][0]
fr = [n for n in k if n.startswith("ntt_fwd_rows8")][0]
ir = [n for n in k if n.startswith("ntt_inv_rows8")][0]
ic = [n for n in k if n.startswith("ntt_inv_cols")][0]


def slim(n):
    return {x: k[n][x] for x in ("avg_ns", "hbm_read_bytes", "hbm_write_bytes", "valu_per_coefficient", "valu_busy")}


cur["forward_kernels"] = f"{fc} + {fr}"
cur["inverse_kernels"] = f"{ir} + {ic}"
cur["fetch_bytes"] = k[fc]["hbm_read_bytes"] + k[fr]["hbm_read_bytes"]
cur["write_bytes"] = k[fc]["hbm_write_bytes"] + k[fr]["hbm_write_bytes"]
cur["hbm_bytes_per_forward_transform"] = cur["fetch_bytes"] + cur["write_bytes"]
cur["hbm_bytes_per_inverse_transform"] = sum(k[n]["hbm_read_bytes"] + k[n]["hbm_write_bytes"] for n in (ir, ic))
cur["per_kernel"] = {n: slim(n) for n in (fc, fr, ir, ic)}
cur["raw_counters"] = "profiles/r03_ntt_pmc_raw.json"
json.dump(cur, open("profiles/ntt_pmc_traffic.json", "w"), indent=1)
shutil.copy("gpurun_out/pmc_r3final.json", "profiles/r03_ntt_pmc_raw.json")
line = [x for x in open(f"{S}/bench.json") if x.startswith("{")][-1]
d = json.loads(line)
d["roofline"]["traffic"] = cur["hbm_bytes_per_forward_transform"]
d["roofline"]["traffic_source"] = cur["collected"]
open("profiles/r03_bench.json", "w").write(json.dumps(d) + "\n")
shutil.copy(f"{S}/bench/bench_kernel_stats.csv", "profiles/r03_bench_kernel_stats.csv")
shutil.copy(f"{S}/configs/configs.jsonl", "profiles/r03_configs.jsonl")
for c in (1, 3, 4, 5):
    shutil.copy(f"{S}/configs/prof_c{c}/cfg{c}_kernel_stats.csv", f"profiles/r03_cfg{c}_kernel_stats.csv")
shutil.copy("gpurun_out/pmc_r3n8192.json", "profiles/r03_ntt_n8192_pmc.json")
for name in ("latency.json",):
    if os.path.exists(f"{S}/{name}"):
        shutil.copy(f"{S}/{name}", "profiles/r03_latency_n32768.json")
if os.path.exists(f"{S}/bench_untraced.json"):
    open("profiles/r03_bench_untraced.json", "w").write([x for x in open(f"{S}/bench_untraced.json") if x.startswith("{")][-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "traffic x", cur["hbm_bytes_per_forward_transform"] / cur["algorithmic_bytes"])
