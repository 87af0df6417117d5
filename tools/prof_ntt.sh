#!/bin/bash
# PMC passes over tools/prof_ntt.py; outputs under gpurun_out/pmc_<tag>/
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-ntt}
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 tools/prof_ntt.py 256 2 > $out/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS
run tcc1 FETCH_SIZE TCC_HIT_sum
run tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
python3 - <<PY
import csv, glob, collections
for name in ("sq1","sq2","tcc1","tcc2"):
    files = glob.glob("$out/%s/**/*counter_collection.csv" % name, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-60:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if "ntt" in k:
            print(name, k, {a: round(b) for a, b in v.items()})
PY
