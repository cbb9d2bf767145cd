cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -x -q -m gpu -k "target_coverage or test_cli or cli_reference" > gpurun_out/s_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/s_tests.log
tail -15 gpurun_out/s_tests.log
