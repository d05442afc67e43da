set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "r1cs_matrix or without_host_synthesis or request or handler" > gpurun_out/r3i_tests.log 2>&1; echo "tests_exit=$?"
tail -15 gpurun_out/r3i_tests.log
timeout -k 10 900 python bench.py --steps 5 --warmup 2 --legs "" --no-cpu-baseline --in-flight 0 > gpurun_out/r3i_bench.json 2> gpurun_out/r3i_bench.err; echo "bench_exit=$?"
tail -c 300 gpurun_out/r3i_bench.err
