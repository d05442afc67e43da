set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3f_stats128t -o st -- python3 $R/tools/prove_loop.py 128 6 tables=0,0 > $R/gpurun_out/r3f_stats128t.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/r3f_sq_a -o sq -- python3 $R/tools/prove_loop.py 128 3 tables=0,0 > $R/gpurun_out/r3f_sq_a.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/r3f_sq_b -o sq -- python3 $R/tools/prove_loop.py 128 3 tables=0,0 > $R/gpurun_out/r3f_sq_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r3f_fetch -o f -- python3 $R/tools/prove_loop.py 128 3 tables=0,0 > $R/gpurun_out/r3f_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r3f_write -o w -- python3 $R/tools/prove_loop.py 128 3 tables=0,0 > $R/gpurun_out/r3f_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3f_stats_bench -o st -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --legs 46,32 > $R/gpurun_out/r3f_stats_bench.log 2>&1
rocprofv3 --kernel-trace --output-format rocpd -d $R/gpurun_out/r3f_tl -o tl -- python3 $R/tools/prove_loop.py 128 3 tables=0,0 > $R/gpurun_out/r3f_tl.log 2>&1
rocprofv3 --kernel-trace --output-format rocpd -d $R/gpurun_out/r3f_tlm -o tl -- python3 $R/tools/prove_matrix_loop.py 128 3 on > $R/gpurun_out/r3f_tlm.log 2>&1
cd $R
python tools/timeline.py $(find gpurun_out/r3f_tl -name "*.db" | head -1) 0.0 > gpurun_out/r3f_timeline_prove.txt; rm -rf gpurun_out/r3f_tl
python tools/timeline.py $(find gpurun_out/r3f_tlm -name "*.db" | head -1) 0.0 8 > gpurun_out/r3f_timeline_prove_matrix.txt; rm -rf gpurun_out/r3f_tlm
find gpurun_out/r3f_stats128t gpurun_out/r3f_stats_bench -type f ! -name "*kernel_stats.csv" -delete
find gpurun_out/r3f_sq_a gpurun_out/r3f_sq_b gpurun_out/r3f_fetch gpurun_out/r3f_write -type f ! -name "*counter_collection.csv" -delete
du -sh gpurun_out/r3f_*
