set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3z_setup32 -o st -- python3 $R/tools/setup_loop.py 32 6 > $R/gpurun_out/r3z_setup32.log 2>&1
cd $R
grep request gpurun_out/r3z_setup32.log
find gpurun_out/r3z_setup32 -type f ! -name "*kernel_stats.csv" -delete
python3 tools/setup_loop.py 32 6 14 > gpurun_out/r3z_b.log 2>&1; tail -2 gpurun_out/r3z_b.log
python3 tools/setup_loop.py 32 6 16 > gpurun_out/r3z_b.log 2>&1; tail -2 gpurun_out/r3z_b.log
python3 tools/setup_loop.py 64 4 > gpurun_out/r3z_b.log 2>&1; tail -2 gpurun_out/r3z_b.log
python3 tools/setup_loop.py 64 4 16 > gpurun_out/r3z_b.log 2>&1; tail -2 gpurun_out/r3z_b.log
python3 tools/setup_loop.py 64 4 18 > gpurun_out/r3z_b.log 2>&1; tail -2 gpurun_out/r3z_b.log
