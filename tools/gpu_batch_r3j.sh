set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python tools/sweep.py matrix 7 > gpurun_out/r3_sweep_matrix.csv 2> gpurun_out/r3_sweep_matrix.err; echo "m_exit=$?"
cat gpurun_out/r3_sweep_matrix.csv
timeout -k 10 600 python tools/sweep.py fib 31 > gpurun_out/r3_sweep_fib.csv 2> gpurun_out/r3_sweep_fib.err; echo "f_exit=$?"
cat gpurun_out/r3_sweep_fib.csv
timeout -k 10 600 python tools/sweep.py prime 2 > gpurun_out/r3_sweep_prime.csv 2> gpurun_out/r3_sweep_prime.err; echo "p_exit=$?"
cat gpurun_out/r3_sweep_prime.csv
