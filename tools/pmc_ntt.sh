#!/bin/bash
# PMC passes over tools/prof_ntt.py (forward + inverse batched NTT); outputs under gpurun_out/pmc_<tag>/ and a JSON
# summary gpurun_out/pmc_<tag>.json.  Usage: tools/pmc_ntt.sh <tag> [polys] [reps]   (environment switches pass through)
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-ntt}
polys=${2:-1024}
reps=${3:-2}
out=$PWD/gpurun_out/pmc_$tag
rm -rf $out
mkdir -p $out
run() { name=$1; shift; timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 tools/prof_ntt.py $polys $reps > $out/$name.log 2>&1; }
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/prof_ntt.py $polys $reps > $out/trace.log 2>&1
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS
run tcc1 FETCH_SIZE TCC_HIT_sum
run tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
python3 tools/pmc_summary.py $out $((polys * 32 * 65536)) ntt
