R=$GRAFT_REPO_ROOT
cd $R
for n in 32 46 64 128; do echo "n=$n"; python3 tools/setup_loop.py $n 4 > gpurun_out/r3z_b.log 2>&1; tail -2 gpurun_out/r3z_b.log; done
