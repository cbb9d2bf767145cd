cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/u_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/u_tests.log
tail -5 gpurun_out/u_tests.log
timeout 600 python bench.py --no-cpu --no-h2h --no-sketch --steps 5 --warmup 2 > gpurun_out/u_bench.json 2> gpurun_out/u_bench.err; echo "bench rc=$?"
cat gpurun_out/u_bench.json
timeout 600 python tools/related_bench.py > gpurun_out/u_related.json 2> gpurun_out/u_related.err; echo "related rc=$?"
cat gpurun_out/u_related.json
