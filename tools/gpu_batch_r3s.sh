set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "msm or window_tables or prove_random or prove_reference or prove_golden" > gpurun_out/r3s_tests.log 2>&1 || { tail -30 gpurun_out/r3s_tests.log; exit 1; }
tail -3 gpurun_out/r3s_tests.log
for n in 32 46 64; do
  timeout -k 10 300 python tools/ab_option.py $n tables=0,0 reduce_mode=6,0,6,0 >> gpurun_out/r3s_ab.log 2>&1 || { tail gpurun_out/r3s_ab.log; exit 1; }
done
cat gpurun_out/r3s_ab.log
