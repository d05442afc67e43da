set -x
R=$GRAFT_REPO_ROOT
cd $R
lscpu | grep -i "model name\|^CPU(s)\|MHz" > gpurun_out/r3a_host.txt
hipcc -O3 -std=c++17 --offload-host-only -x hip -Izksnark-finalproject_amd/csrc tools/bench_poseidon_host.cpp -o /tmp/bench_poseidon_host 2> gpurun_out/r3a_hostbench.err && /tmp/bench_poseidon_host >> gpurun_out/r3a_host.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "witness_matrix or device_witness or with_tables" > gpurun_out/r3a_tests.log 2>&1; echo "tests_exit=$?"
tail -5 gpurun_out/r3a_tests.log
timeout -k 10 600 python tools/e2e_witness.py 128 3 on > gpurun_out/r3a_e2e.log 2>&1; echo "e2e_exit=$?"
cat gpurun_out/r3a_e2e.log
