#!/bin/bash
# Same-box per-kernel A/B: kernel trace of bench.py with tools/ab/libcat_base.so, then with the tree's library.
# Results: gpurun_out/ab_prof/{base,new}/*kernel_stats.csv; tools/ab_prof_diff.py prints them side by side.
cd "$(dirname "$0")/.."
root=$PWD
out=$root/gpurun_out/ab_prof
rm -rf $out; mkdir -p $out
cp cat_amd/lib/libcat_amd.so /tmp/libcat_new.so
cp tools/ab/libcat_base.so cat_amd/lib/libcat_amd.so
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/base -o b -- python3 $root/bench.py --skip-cpu --steps 20 --warmup 3 > $out/base.log 2>&1 || { cp /tmp/libcat_new.so $root/cat_amd/lib/libcat_amd.so; exit 1; }
cp /tmp/libcat_new.so $root/cat_amd/lib/libcat_amd.so
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/new -o b -- python3 $root/bench.py --skip-cpu --steps 20 --warmup 3 > $out/new.log 2>&1 || exit 1
