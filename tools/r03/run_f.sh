# round 3: throughput + roofline of every BASELINE config other than bench.py's line, then one rocprofv3 kernel-stats pass per config
set -e
out=${1:-gpurun_out/r03f}
mkdir -p $out
timeout -k 10 600 python3 tests/bench_configs.py --steps 10 > $out/configs.jsonl 2> $out/configs.err
cat $out/configs.jsonl | cut -c1-330
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for c in 1 3 4 5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_c$c -o cfg$c -- python3 tests/bench_configs.py --steps 5 --only $c > $out/prof_c$c.log 2>&1
done
