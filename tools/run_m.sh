cd $GRAFT_REPO_ROOT
timeout 900 python tools/sweep_engines.py > gpurun_out/m_sweep.txt 2> gpurun_out/m_sweep.err
cat gpurun_out/m_sweep.txt; tail -c 300 gpurun_out/m_sweep.err
