cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/ -x -q -m gpu > gpurun_out/g_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/g_tests.log
tail -15 gpurun_out/g_tests.log
