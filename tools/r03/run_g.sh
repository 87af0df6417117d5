set -e
out=${1:-gpurun_out/r03g}
mkdir -p $out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest.txt 2>&1 || { tail -40 $out/pytest.txt; exit 1; }
tail -3 $out/pytest.txt
timeout -k 10 600 python3 tests/bench_configs.py --steps 10 > $out/configs.jsonl 2> $out/configs.err
python3 - $out/configs.jsonl <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    r=json.loads(ln); print(r["config"], {k:round(v,1) for k,v in r.items() if k.endswith("_per_s")}, "frac %.3f"%r["roofline"]["frac"], r["parity_checked"])
PY
