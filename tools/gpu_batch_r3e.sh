set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "request or handler or prove_matrix" > gpurun_out/r3e_tests.log 2>&1; echo "tests_exit=$?"
tail -5 gpurun_out/r3e_tests.log
timeout -k 10 600 python tools/e2e_witness.py 128 3 on > gpurun_out/r3e_e2e.log 2>&1; echo "e2e_exit=$?"
grep prove_matrix gpurun_out/r3e_e2e.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format rocpd -d $R/gpurun_out/r3e_tl -o tl -- python3 $R/tools/prove_matrix_loop.py 128 3 on > $R/gpurun_out/r3e_tl.log 2>&1
cd $R
python tools/timeline.py $(find gpurun_out/r3e_tl -name "*.db" | head -1) 0.0 > gpurun_out/r3e_timeline.txt; rm -rf gpurun_out/r3e_tl
cat gpurun_out/r3e_tl.log | tail -5
