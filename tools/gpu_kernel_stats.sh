set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3f_stats128t -o st -- python3 $R/tools/prove_loop.py 128 6 tables=0,0 > $R/gpurun_out/r3f_stats128t.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3f_bench -o st -- python3 $R/bench.py > $R/gpurun_out/r3f_bench_under_rocprof.json 2> $R/gpurun_out/r3f_bench.err
cd $R
find gpurun_out/r3f_stats128t gpurun_out/r3f_bench -type f ! -name "*kernel_stats.csv" -delete
ls -la gpurun_out/r3f_stats128t gpurun_out/r3f_bench
tail -c 600 gpurun_out/r3f_bench_under_rocprof.json
