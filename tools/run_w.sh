cd $GRAFT_REPO_ROOT
timeout 900 python tools/sweep_compare.py --n 100000 --rounds 2 win win500 win520 win555 win575 win600 win537k2 win537k4 win640k4 win700k4 > gpurun_out/w_sweep.txt 2>&1; echo rc=$?
cat gpurun_out/w_sweep.txt | tail -14
