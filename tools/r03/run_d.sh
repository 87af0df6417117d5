set -e
mkdir -p gpurun_out/r03d
timeout -k 10 600 python -u -m pytest tests/test_gpu_workspace.py tests/test_gpu_comm.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r03d/pytest.txt 2>&1 || { tail -40 gpurun_out/r03d/pytest.txt; exit 1; }
tail -3 gpurun_out/r03d/pytest.txt
timeout -k 10 400 ./tools/latency_facade 10 > gpurun_out/r03d/latency.json 2> gpurun_out/r03d/latency.err
cat gpurun_out/r03d/latency.json
