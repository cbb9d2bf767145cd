cd $GRAFT_REPO_ROOT
timeout 600 python tools/compare_fuzz.py --n 3000 --seed 3 --seconds 150 > gpurun_out/z_cfuzz3.txt 2>&1; echo rc=$?
tail -12 gpurun_out/z_cfuzz3.txt | cut -c1-400
timeout 300 python tools/compare_fuzz.py --n 2047 --seed 3 --only 2046 --seconds 1000 > gpurun_out/z_cfuzz3b.txt 2>&1; echo rc=$?
tail -12 gpurun_out/z_cfuzz3b.txt | cut -c1-400
