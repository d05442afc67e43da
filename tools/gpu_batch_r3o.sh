set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fixed_base or setup or error_paths or handler_mirrors" > gpurun_out/r3o_tests.log 2>&1 || { tail -30 gpurun_out/r3o_tests.log; exit 1; }
tail -3 gpurun_out/r3o_tests.log
for bits in 14 16 18 20 0; do
  echo "== fixed_base_bits $bits" >> gpurun_out/r3o_setup.log
  timeout -k 10 200 python tools/setup_loop.py 128 3 $bits >> gpurun_out/r3o_setup.log 2>&1 || { tail gpurun_out/r3o_setup.log; exit 1; }
done
cat gpurun_out/r3o_setup.log
