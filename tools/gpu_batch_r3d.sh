set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "prove_matrix or witness_matrix or device_witness" > gpurun_out/r3d_tests.log 2>&1; echo "tests_exit=$?"
tail -15 gpurun_out/r3d_tests.log
ZKG16_TRACE_HOST=1 timeout -k 10 600 python tools/e2e_witness.py 128 3 on > gpurun_out/r3d_e2e.log 2>&1; echo "e2e_exit=$?"
cat gpurun_out/r3d_e2e.log
