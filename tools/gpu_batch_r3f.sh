set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3f_tests.log 2>&1; echo "tests_exit=$?"
tail -8 gpurun_out/r3f_tests.log
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3f_bench.json 2> gpurun_out/r3f_bench.err; echo "bench_exit=$?"
tail -c 400 gpurun_out/r3f_bench.err
