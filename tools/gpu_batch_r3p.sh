set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "setup or handler_mirrors or whole_request" > gpurun_out/r3p_tests.log 2>&1 || { tail -30 gpurun_out/r3p_tests.log; exit 1; }
tail -3 gpurun_out/r3p_tests.log
timeout -k 10 200 python tools/setup_loop.py 128 4 > gpurun_out/r3p_setup.log 2>&1 || { tail gpurun_out/r3p_setup.log; exit 1; }
cat gpurun_out/r3p_setup.log
timeout -k 10 200 python tools/setup_loop.py 32 4 >> gpurun_out/r3p_setup.log 2>&1
tail -4 gpurun_out/r3p_setup.log
