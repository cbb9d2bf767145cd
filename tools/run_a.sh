set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "compare or c3" > gpurun_out/a_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/a_tests.log
tail -5 gpurun_out/a_tests.log
for v in 3 2 4; do
  MASHGPU_COMPARE_VARIANT=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-sketch --no-screen --no-cpu > gpurun_out/a_bench_v$v.json 2> gpurun_out/a_bench_v$v.err; tail -c 600 gpurun_out/a_bench_v$v.json
done
MASHGPU_COMPARE_WINDOWS=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-sketch --no-screen --no-cpu > gpurun_out/a_bench_plain.json 2> gpurun_out/a_bench_plain.err; tail -c 400 gpurun_out/a_bench_plain.json
