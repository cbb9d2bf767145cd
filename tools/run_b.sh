cd $GRAFT_REPO_ROOT
MASHGPU_COMPARE_DBG=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-sketch --no-screen --no-cpu > gpurun_out/b_dbg.json 2> gpurun_out/b_dbg.err
grep "compare dbg" gpurun_out/b_dbg.err | tail -8
MASHGPU_COMPARE_DBG=1 MASHGPU_COMPARE_WINDOWS=0 timeout 300 python bench.py --steps 1 --warmup 1 --no-sketch --no-screen --no-cpu > gpurun_out/b_dbg_plain.json 2> gpurun_out/b_dbg_plain.err
grep "compare dbg" gpurun_out/b_dbg_plain.err | tail -3
