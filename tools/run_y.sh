cd $GRAFT_REPO_ROOT
timeout 600 python tools/cli_reference_report.py --extra > gpurun_out/y_cli_report.txt 2>&1; tail -18 gpurun_out/y_cli_report.txt
timeout 1500 python -m pytest tests/ -x -q -m gpu -k "bloom or reads or cli or target_coverage" > gpurun_out/y_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/y_tests.log
tail -15 gpurun_out/y_tests.log
