cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "compare or c3" > gpurun_out/d_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/d_tests.log
tail -3 gpurun_out/d_tests.log
for v in 3 2; do
  MASHGPU_COMPARE_VARIANT=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-sketch --no-screen --no-cpu > gpurun_out/d_bench_v$v.json 2> gpurun_out/d_bench_v$v.err
  python -c "import json;d=json.load(open('gpurun_out/d_bench_v$v.json'));print('v$v %.3e'%d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['launches'])"
done
MASHGPU_COMPARE_DBG=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-sketch --no-screen --no-cpu > gpurun_out/d_dbg.json 2> gpurun_out/d_dbg.err
grep "compare dbg" gpurun_out/d_dbg.err | tail -2
