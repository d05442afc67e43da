set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 700 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r3r_tests.log 2>&1 || { tail -30 gpurun_out/r3r_tests.log; exit 1; }
tail -14 gpurun_out/r3r_tests.log
timeout -k 10 420 python bench.py > gpurun_out/r3r_bench.json 2> gpurun_out/r3r_bench.err || { tail -20 gpurun_out/r3r_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3r_bench.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], json.dumps(d.get("end_to_end",{}).get("request_with_setup")), d.get("setup_s"))
PY
