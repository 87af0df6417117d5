set -e
mkdir -p gpurun_out/r03e
timeout -k 10 400 ./tools/latency_facade 10 > gpurun_out/r03e/latency.json 2> gpurun_out/r03e/latency.err
cat gpurun_out/r03e/latency.json
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03e/prof -o lat -- ./tools/latency_facade 10 > gpurun_out/r03e/prof.log 2>&1
ls gpurun_out/r03e/prof | head
