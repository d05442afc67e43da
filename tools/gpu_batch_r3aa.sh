R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size_request or handler or resident_setup or whole_request or setup_prove_verify" > gpurun_out/r3aa_tests.log 2>&1 || { tail -30 gpurun_out/r3aa_tests.log; exit 1; }
tail -3 gpurun_out/r3aa_tests.log
python tools/sweep.py matrix 7 > gpurun_out/sweep_r3_matrix.csv 2>&1; cat gpurun_out/sweep_r3_matrix.csv
