# round 3, experiment B: Infinity-Cache probe + poly-major column order on the metric op
set -e
mkdir -p gpurun_out/r03b
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for c in 0 2 4 8 16; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/r03b/mall_$c -o mall -- python3 tools/r03/mall_probe.py $c > gpurun_out/r03b/mall_$c.log 2>&1
done
for v in "CAT_X=0" "CAT_COLS_POLY_MAJOR=1"; do
  echo "== $v" >> gpurun_out/r03b/bench_order.txt
  env $v timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --skip-cpu --skip-ntt >> gpurun_out/r03b/bench_order.txt 2>&1
done
