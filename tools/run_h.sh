cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/ -x -q -m gpu -k "sharded or rank_communicator or cli or device_finish" > gpurun_out/h_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/h_tests.log
tail -30 gpurun_out/h_tests.log
