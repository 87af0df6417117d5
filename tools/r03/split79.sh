# timing experiment: the column pass of N = 65536 as 128 rows x 512 (64-column tiles, 512-byte segments) against today's 256 x 256
set -e
out=gpurun_out/r03h
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/s79_base -o p -- python3 tools/prof_ntt.py 1024 3 > $out/s79_base.log 2>&1
export CAT_SPLIT79=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/s79_new -o p -- python3 tools/prof_ntt.py 1024 3 > $out/s79_new.log 2>&1 || true
grep -h "ntt_fwd_cols" $out/s79_base/p_kernel_stats.csv $out/s79_new/p_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
