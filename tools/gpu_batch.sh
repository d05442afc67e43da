set -x
cd $GRAFT_REPO_ROOT
python bench.py --gpus 2 --dist-backend gloo --matrix-n 32 --steps 5 --warmup 2 > gpurun_out/r2_bench_2rank_gloo.json 2> gpurun_out/r2_bench_2rank_gloo.err; echo "2rank_exit=$?"
tail -c 600 gpurun_out/r2_bench_2rank_gloo.err
python bench.py --gpus 4 --dist-backend gloo --matrix-n 46 --steps 5 --warmup 2 > gpurun_out/r2_bench_4rank_gloo.json 2> gpurun_out/r2_bench_4rank_gloo.err; echo "4rank_exit=$?"
python tools/shard_timing.py 128 1,2,4,8 2>&1 | tee gpurun_out/r2_shard128b.txt
python tools/shard_timing.py 46 1,8 2>&1 | tee gpurun_out/r2_shard46b.txt
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_sq32 -o sq -- python3 $GRAFT_REPO_ROOT/tools/prove_loop.py 32 3 > $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_sq32.log 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_sq128 -o sq -- python3 $GRAFT_REPO_ROOT/tools/prove_loop.py 128 2 > $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_sq128.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_sq32 $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_sq128
