cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/k_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/k_tests.log
tail -12 gpurun_out/k_tests.log
