set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3n_setup -o st -- python3 $R/tools/setup_loop.py 128 3 > $R/gpurun_out/r3n_setup.log 2>&1
cd $R
cat gpurun_out/r3n_setup.log | tail -5
find gpurun_out/r3n_setup -type f ! -name "*kernel_stats.csv" -delete
cut -c1-140 $(find gpurun_out/r3n_setup -name "*kernel_stats.csv") | head -25
