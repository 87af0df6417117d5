# round 3, final measurements: bench line under the kernel tracer, PMC passes over the config-2 transform and over the
# single-pass N = 8192 transform, per-config lines and kernel stats.  Outputs under gpurun_out/r03final/.
set -e
out=gpurun_out/r03final
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o bench -- python3 bench.py > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.json; echo
bash tools/pmc_ntt.sh r3final 1024 2 > $out/pmc_ntt.log 2>&1
bash tools/pmc_generic.sh r3n8192 python3 tools/r03/ntt_sizes.py 13 > $out/pmc_n8192.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_r3n8192 $((1024 * 16 * 8192)) ntt > $out/pmc_n8192_summary.txt 2>&1 || true
bash tools/r03/run_f.sh $out/configs
