set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "msm_golden or window_tables or prove_random or prove_reference" > gpurun_out/r3u_tests.log 2>&1 || { tail -30 gpurun_out/r3u_tests.log; exit 1; }
tail -3 gpurun_out/r3u_tests.log
for k in prime 32 46 64; do python tools/spans.py $k reduce_mode 6,0,6,0 >> gpurun_out/r3u_spans.log 2>&1; done
for k in 8 16 32 64 128; do python tools/spans.py $k reduce_mode 6,0,6,0 tables=0 >> gpurun_out/r3u_spans.log 2>&1; done
grep -v "^{" gpurun_out/r3u_spans.log
