# round 3, experiment A: order of the workgroups over the buffer (micro-benchmark + the library's kernels)
set -e
mkdir -p gpurun_out
timeout -k 10 200 ./tools/ubench_pass_shapes > gpurun_out/r03_pass_shapes2.txt 2>&1
rm -f gpurun_out/r03_ntt_order.txt
for v in "CAT_X=0" "CAT_COLS_POLY_MAJOR=1" "CAT_ROWS8_BLOCKS=1024" "CAT_ROWS8_BLOCKS=2048" "CAT_ROWS8_BLOCKS=1024 CAT_COLS_POLY_MAJOR=1"; do
  echo "== $v" >> gpurun_out/r03_ntt_order.txt
  env $v timeout -k 10 150 python3 tools/prof_ntt.py 1024 4 >> gpurun_out/r03_ntt_order.txt 2>&1
done
