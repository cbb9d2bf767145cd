cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "compare or c3 or config5 or streamed or sharded" > gpurun_out/j_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/j_tests.log
tail -3 gpurun_out/j_tests.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-sketch --no-screen --no-cpu --no-h2h > gpurun_out/j_bench.json 2> gpurun_out/j_bench.err
python -c "
import json;d=json.load(open('gpurun_out/j_bench.json'));print('win %.3e'%d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['launches'], d['config']['output_checksum']); print('c5 %.3e'%d['c5']['value'], d['c5']['config']['output_checksum'])"
tail -c 300 gpurun_out/j_bench.err
MASHGPU_COMPARE_WINDOWS=0 timeout 600 python bench.py --steps 1 --warmup 0 --no-sketch --no-screen --no-cpu --no-h2h > gpurun_out/j_bench_plain.json 2> gpurun_out/j_bench_plain.err
python -c "
import json;d=json.load(open('gpurun_out/j_bench_plain.json'));print('plain %.3e'%d['value'], d['config']['output_checksum']); print('c5 plain', d['c5'].get('value'), d['c5'].get('config',{}).get('output_checksum'), d['c5'].get('error'))"
tail -c 300 gpurun_out/j_bench_plain.err
timeout 600 python -m pytest tests/test_cli.py -x -q -m gpu 2>&1 | tail -3
