set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 800 python -m pytest tests -x -q -m gpu > gpurun_out/r3x_tests.log 2>&1 || { tail -30 gpurun_out/r3x_tests.log; exit 1; }
tail -3 gpurun_out/r3x_tests.log
timeout -k 10 420 python bench.py > gpurun_out/r3x_bench.json 2> gpurun_out/r3x_bench.err || { tail -20 gpurun_out/r3x_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3x_bench.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for l in d["legs"]:
    print(l["n"], l["ms_per_step"], l["window_tables"]["window_bits_z"], l["window_tables"]["window_bits_h"], l["window_tables"]["plain_key_ms_per_proof"], l.get("throughput_in_flight",{}).get("value"), l.get("oracle_match"))
PY
