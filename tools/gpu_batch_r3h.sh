set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3h_tests.log 2>&1; echo "tests_exit=$?"
tail -5 gpurun_out/r3h_tests.log
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3h_bench.json 2> gpurun_out/r3h_bench.err; echo "bench_exit=$?"
timeout -k 10 300 python tools/verify_timing.py > gpurun_out/r3h_verify.log 2>&1; echo "verify_exit=$?"
cat gpurun_out/r3h_verify.log
timeout -k 10 600 python bench.py --gpus 2 --dist-backend gloo --matrix-n 32 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --legs "" > gpurun_out/r3h_2rank.json 2> gpurun_out/r3h_2rank.err; echo "2rank_exit=$?"
tail -c 1500 gpurun_out/r3h_2rank.json
