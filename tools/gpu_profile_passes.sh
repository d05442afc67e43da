# rocprofv3 passes of the 128x128 prove loop: kernel stats, SQ counters (two passes), FETCH_SIZE, WRITE_SIZE (each PMC pass on its own)
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
$R/tools/bin/microbench > $R/gpurun_out/r3_microbench.txt 2>&1; echo "microbench_exit=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_stats128t -o st -- python3 $R/tools/prove_loop.py 128 6 tables=0,0 > $R/gpurun_out/r3_stats128t.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/r3_sq128t_a -o sq -- python3 $R/tools/prove_loop.py 128 3 tables=0,0 > $R/gpurun_out/r3_sq128t_a.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/r3_sq128t_b -o sq -- python3 $R/tools/prove_loop.py 128 3 tables=0,0 > $R/gpurun_out/r3_sq128t_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r3_fetch128t -o f -- python3 $R/tools/prove_loop.py 128 3 tables=0,0 > $R/gpurun_out/r3_fetch128t.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r3_write128t -o w -- python3 $R/tools/prove_loop.py 128 3 tables=0,0 > $R/gpurun_out/r3_write128t.log 2>&1
cd $R
find gpurun_out/r3_sq128t_a gpurun_out/r3_sq128t_b gpurun_out/r3_stats128t gpurun_out/r3_fetch128t gpurun_out/r3_write128t -name "*.csv" | head -40
# keep only the csv summaries (the raw agent-info files are not needed)
find gpurun_out/r3_stats128t -type f ! -name "*kernel_stats.csv" -delete
find gpurun_out/r3_sq128t_a gpurun_out/r3_sq128t_b gpurun_out/r3_fetch128t gpurun_out/r3_write128t -type f ! -name "*counter_collection.csv" -delete
du -sh gpurun_out/r3_*
