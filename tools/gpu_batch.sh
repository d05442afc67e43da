set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_fetch2 -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --legs "" > $R/gpurun_out/r2_pmc_fetch2.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_write2 -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --legs "" > $R/gpurun_out/r2_pmc_write2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_stats32 -o st -- python3 $R/tools/prove_loop.py 32 11 > $R/gpurun_out/r2_stats32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_stats128 -o st -- python3 $R/tools/prove_loop.py 128 6 > $R/gpurun_out/r2_stats128.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_stats_bench -o st -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --legs 46,32 > $R/gpurun_out/r2_stats_bench.log 2>&1
cd $R
python bench.py > gpurun_out/r2_bench5.json 2> gpurun_out/r2_bench5.err; echo "bench_exit=$?"
