set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "prove_random or prove_reference or concurrent or error_paths" > gpurun_out/r3v_tests.log 2>&1 || { tail -30 gpurun_out/r3v_tests.log; exit 1; }
tail -3 gpurun_out/r3v_tests.log
for k in prime fib1000; do python tools/spans.py $k reduce_mode 6,0,6,0 >> gpurun_out/r3v_spans.log 2>&1; done
for k in 2 8 16 32 64; do python tools/spans.py $k reduce_mode 6,0,6,0 tables=0 >> gpurun_out/r3v_spans.log 2>&1; done
python tools/spans.py 8 collect_threads 2,0,2,0 tables=0 >> gpurun_out/r3v_spans.log 2>&1
python tools/spans.py fib1000 collect_threads 2,0,2,0 tables=0 >> gpurun_out/r3v_spans.log 2>&1
grep -v "^{" gpurun_out/r3v_spans.log
