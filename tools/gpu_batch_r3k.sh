set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "error_paths or two_streamed or concurrent" --durations=5 > gpurun_out/r3k_tests.log 2>&1; echo "tests_exit=$?"
tail -15 gpurun_out/r3k_tests.log
