#!/bin/bash
# Same-box A/B of the metric step's per-kernel VALU instruction counts (one --pmc pass each): tools/ab/libcat_base.so,
# then the tree's library.  Output: gpurun_out/pmc_step_base.txt, gpurun_out/pmc_step_new.txt
cd "$(dirname "$0")/.."
cp cat_amd/lib/libcat_amd.so /tmp/libcat_new.so
cp tools/ab/libcat_base.so cat_amd/lib/libcat_amd.so
tools/prof_step_pmc.sh base > gpurun_out/pmc_step_base.txt 2>&1; rc=$?
cp /tmp/libcat_new.so cat_amd/lib/libcat_amd.so
[ $rc -eq 0 ] || exit $rc
tools/prof_step_pmc.sh new > gpurun_out/pmc_step_new.txt 2>&1
