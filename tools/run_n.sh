cd $GRAFT_REPO_ROOT
timeout 900 python tools/sweep_engines.py tiles > gpurun_out/n_sweep.txt 2> gpurun_out/n_sweep.err
cat gpurun_out/n_sweep.txt; tail -c 300 gpurun_out/n_sweep.err
