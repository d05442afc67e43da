set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_fetch5 -o f -- python3 $R/bench.py --steps 2 --warmup 1 --tables on --no-cpu-baseline --no-e2e --legs "" > $R/gpurun_out/r2_pmc_fetch5.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_write5 -o w -- python3 $R/bench.py --steps 2 --warmup 1 --tables on --no-cpu-baseline --no-e2e --legs "" > $R/gpurun_out/r2_pmc_write5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_stats32t -o st -- python3 $R/tools/prove_loop.py 32 11 tables=0,0 > $R/gpurun_out/r2_stats32t.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_stats128t -o st -- python3 $R/tools/prove_loop.py 128 6 tables=0,0 > $R/gpurun_out/r2_stats128t.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_stats_bencht -o st -- python3 $R/bench.py --steps 3 --warmup 1 --tables on --no-cpu-baseline --no-e2e --legs 46,32 > $R/gpurun_out/r2_stats_bencht.log 2>&1
rocprofv3 --kernel-trace --output-format rocpd -d $R/gpurun_out/r2_tl128t -o tl -- python3 $R/tools/prove_loop.py 128 3 tables=0,0 > $R/gpurun_out/r2_tl128t.log 2>&1
rocprofv3 --kernel-trace --output-format rocpd -d $R/gpurun_out/r2_tl32t -o tl -- python3 $R/tools/prove_loop.py 32 6 tables=0,0 > $R/gpurun_out/r2_tl32t.log 2>&1
cd $R
python tools/timeline.py $(find gpurun_out/r2_tl128t -name "*.db" | head -1) 0.0 > gpurun_out/r2_tl128t.txt; rm -rf gpurun_out/r2_tl128t
python tools/timeline.py $(find gpurun_out/r2_tl32t -name "*.db" | head -1) 0.0 > gpurun_out/r2_tl32t.txt; rm -rf gpurun_out/r2_tl32t
python bench.py > gpurun_out/r2_bench7.json 2> gpurun_out/r2_bench7.err; echo "bench_exit=$?"
python bench.py --workload prime --legs "" --no-cpu-baseline > gpurun_out/r2_bench7_prime.json 2> gpurun_out/r2_bench7_prime.err; echo "prime_exit=$?"
python bench.py --tables off --legs "46,32" --no-cpu-baseline --no-e2e > gpurun_out/r2_bench7_plain.json 2> gpurun_out/r2_bench7_plain.err; echo "plain_exit=$?"
python tools/sweep.py matrix 7 > gpurun_out/sweep_matrix.csv 2> gpurun_out/sweep_matrix.err
python tools/sweep.py fib 31 > gpurun_out/sweep_fib.csv 2> gpurun_out/sweep_fib.err
python tools/sweep.py prime 2 > gpurun_out/sweep_prime.csv 2> gpurun_out/sweep_prime.err
