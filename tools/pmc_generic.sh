#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 hit passes over an arbitrary program; outputs under gpurun_out/pmc_<tag>/
# Usage: tools/pmc_generic.sh <tag> <program> [args...]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift
out=$PWD/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $out/tcc1 -- "$@" > $out/tcc1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/tcc2 -- "$@" > $out/tcc2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- "$@" > $out/trace.log 2>&1 || exit 1
