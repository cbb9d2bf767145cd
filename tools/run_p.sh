cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "compare" > gpurun_out/p_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/p_tests.log
tail -25 gpurun_out/p_tests.log
for v in 3 4 2; do
MASHGPU_COMPARE_DIRECT=1 MASHGPU_COMPARE_VARIANT=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-sketch --no-screen --no-cpu --no-h2h > gpurun_out/p_bench_direct$v.json 2> gpurun_out/p_bench_direct$v.err
python -c "
import json;d=json.load(open('gpurun_out/p_bench_direct$v.json'));print('direct v$v %.3e'%d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['launches'], d['config']['output_checksum']); print('c5 %.3e'%d['c5'].get('value',0), d['c5'].get('error'))"
tail -c 300 gpurun_out/p_bench_direct$v.err
done
timeout 600 python tools/related_bench.py --n 20000 --engines default,windows,direct,plain > gpurun_out/p_related.json 2> gpurun_out/p_related.err
python -c "
import json;d=json.load(open('gpurun_out/p_related.json'))
for k,v in d['cases'].items(): print(k, {e:('%.3e'%x['pairs_per_s']) for e,x in v.items() if isinstance(x,dict)}, v['engines_agree'])"
tail -c 300 gpurun_out/p_related.err
