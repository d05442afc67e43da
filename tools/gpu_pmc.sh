set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_fetch4 -o f -- python3 $R/bench.py --steps 2 --warmup 1 --tables on --no-cpu-baseline --no-e2e --legs "" > $R/gpurun_out/r2_pmc_fetch4.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_write4 -o w -- python3 $R/bench.py --steps 2 --warmup 1 --tables on --no-cpu-baseline --no-e2e --legs "" > $R/gpurun_out/r2_pmc_write4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_stats_bencht -o st -- python3 $R/bench.py --steps 3 --warmup 1 --tables on --no-cpu-baseline --no-e2e --legs 46,32 > $R/gpurun_out/r2_stats_bencht.log 2>&1
