#!/bin/bash
# PMC pass over the metric step (bench.py without the NTT and CPU legs); output under gpurun_out/pmc_step[_<tag>]/
# Usage: tools/prof_step_pmc.sh [tag]
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_step${1:+_$1}
rm -rf $out
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/sq1 -- python3 bench.py --skip-ntt --skip-cpu --steps 2 --warmup 1 > $out/sq1.log 2>&1
python3 - <<PY
import csv, glob, collections
files = glob.glob("$out/sq1/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-48:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:10]:
    g = v.get("GRBM_GUI_ACTIVE", 1)
    # 256 CUs x 4 SIMDs; SQ_ACTIVE_INST_VALU counts cycles x 4 per the guide's convention used in prof_ntt.sh
    n = max(cnt[k], 1)
    print(f"{k:50s} launches {cnt[k]:3d} valu_insts_per_launch {v.get('SQ_INSTS_VALU',0)/n:.4e} valu_busy {v.get('SQ_ACTIVE_INST_VALU',0)*4/1024/(g/8):.3f} gui_active {g:.3e} valu_insts {v.get('SQ_INSTS_VALU',0):.3e} active_valu {v.get('SQ_ACTIVE_INST_VALU',0):.3e} busy_cycles {v.get('SQ_BUSY_CYCLES',0):.3e} wave_cycles {v.get('SQ_WAVE_CYCLES',0):.3e} wait_any {v.get('SQ_WAIT_ANY',0):.3e} wait_inst_any {v.get('SQ_WAIT_INST_ANY',0):.3e}")
PY
