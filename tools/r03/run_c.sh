# round 3, experiment C: GPU suite after the column order / hygiene changes, then ND = 1 in the forward streaming row pass
set -e
mkdir -p gpurun_out/r03c
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03c/pytest.txt 2>&1 || { tail -30 gpurun_out/r03c/pytest.txt; exit 1; }
tail -3 gpurun_out/r03c/pytest.txt
for v in "CAT_X=0" "CAT_ROWS8_ND1=1"; do
  echo "== $v" >> gpurun_out/r03c/ntt_nd.txt
  env $v timeout -k 10 150 python3 tools/prof_ntt.py 1024 4 >> gpurun_out/r03c/ntt_nd.txt 2>&1
done
cat gpurun_out/r03c/ntt_nd.txt
