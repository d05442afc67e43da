set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "g2_lazy or random_vs_oracle_and_exponent or msm_golden" > gpurun_out/r3g_tests.log 2>&1; echo "tests_exit=$?"
tail -12 gpurun_out/r3g_tests.log
timeout -k 10 600 python tools/ab_option.py 128 tables=0,0 g2_lazy=0,1 > gpurun_out/r3g_ab128.log 2>&1; echo "ab_exit=$?"
cat gpurun_out/r3g_ab128.log
timeout -k 10 600 python tools/ab_option.py 32 tables=0,0 g2_lazy=0,1 > gpurun_out/r3g_ab32.log 2>&1; echo "ab_exit=$?"
cat gpurun_out/r3g_ab32.log
