cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "compare or c3" > gpurun_out/c_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/c_tests.log
tail -3 gpurun_out/c_tests.log
for v in 3 4; do
  MASHGPU_COMPARE_VARIANT=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-sketch --no-screen --no-cpu > gpurun_out/c_bench_v$v.json 2> gpurun_out/c_bench_v$v.err
  python -c "import json;d=json.load(open('gpurun_out/c_bench_v$v.json'));print('v$v %.3e'%d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['launches'])"
done
MASHGPU_COMPARE_DBG=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-sketch --no-screen --no-cpu > gpurun_out/c_dbg.json 2> gpurun_out/c_dbg.err
grep "compare dbg" gpurun_out/c_dbg.err | tail -4
MASHGPU_COMPARE_WINDOWS=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-sketch --no-screen --no-cpu > gpurun_out/c_bench_plain.json 2> gpurun_out/c_bench_plain.err
python -c "import json;d=json.load(open('gpurun_out/c_bench_plain.json'));print('plain %.3e'%d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['launches'])"
