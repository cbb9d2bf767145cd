cd $GRAFT_REPO_ROOT
timeout 900 python tools/cli_fuzz.py --n 700 --seed 2 --seconds 400 > gpurun_out/z_fuzz2.txt 2>&1; echo rc=$?
tail -40 gpurun_out/z_fuzz2.txt | cut -c1-600
timeout 900 python -m pytest tests/test_cli.py -x -q -m gpu > gpurun_out/z_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/z_tests.log
