set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "msm or window_tables or prove_random or prove_reference or prove_golden or edge_shapes" > gpurun_out/r3t_tests.log 2>&1 || { tail -30 gpurun_out/r3t_tests.log; exit 1; }
tail -3 gpurun_out/r3t_tests.log
for k in prime 32 46; do python tools/spans.py $k reduce_mode 6,0 >> gpurun_out/r3t_spans.log 2>&1; done
for k in 8 32 64 128; do python tools/spans.py $k reduce_mode 6,0 tables=0 >> gpurun_out/r3t_spans.log 2>&1; done
python tools/spans.py fib1000 reduce_mode 6,0,5 tables=0 >> gpurun_out/r3t_spans.log 2>&1
grep -v "^{" gpurun_out/r3t_spans.log
