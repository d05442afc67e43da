set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_fetch3 -o f -- python3 $R/bench.py --steps 2 --warmup 1 --tables on --no-cpu-baseline --no-e2e --legs "" > $R/gpurun_out/r2_pmc_fetch3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_write3 -o w -- python3 $R/bench.py --steps 2 --warmup 1 --tables on --no-cpu-baseline --no-e2e --legs "" > $R/gpurun_out/r2_pmc_write3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_stats32t -o st -- python3 $R/tools/prove_loop.py 32 11 tables=0,0 > $R/gpurun_out/r2_stats32t.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_stats128t -o st -- python3 $R/tools/prove_loop.py 128 6 tables=0,0 > $R/gpurun_out/r2_stats128t.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_stats_bencht -o st -- python3 $R/bench.py --steps 3 --warmup 1 --tables on --no-cpu-baseline --no-e2e --legs 46,32 > $R/gpurun_out/r2_stats_bencht.log 2>&1
cd $R
python bench.py > gpurun_out/r2_bench6.json 2> gpurun_out/r2_bench6.err; echo "bench_exit=$?"
