set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1500 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r3m_tests.log 2>&1; echo "tests_exit=$?"
tail -16 gpurun_out/r3m_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3m_smoke.log 2>&1; echo "smoke_exit=$?"
tail -2 gpurun_out/r3m_smoke.log
