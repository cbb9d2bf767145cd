#!/bin/bash
# round-3 GPU run P: run dedupe in the index (clades), pack merge default; parity + fuzz + brackets
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_refcli_gpu.py -m gpu -q -x -k "compare or sparse or triangle or rect or identical or copies or runs or finish or survivor or sharded_compare or bound" ; echo "rc=$?" ) > gpurun_out/p_tests.log 2>&1; tail -4 gpurun_out/p_tests.log
( timeout 300 python tools/compare_fuzz.py --n 100000 --seconds 60 --seed 94 ) > gpurun_out/p_compare_fuzz.txt 2>&1; tail -2 gpurun_out/p_compare_fuzz.txt
( MASHGPU_SPARSE_DBG=1 timeout 900 python bench.py --no-h2h --no-cpu --no-sketch --no-cli --no-screen ) > gpurun_out/p_bench.json 2> gpurun_out/p_bench.err; grep "compare sparse: index" gpurun_out/p_bench.err | head -8; tail -c 300 gpurun_out/p_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/p_bench.json') if l.startswith('{')][-1])
print('c3', d['value'], d['ms_per_step'], d['config'].get('first_call_ms'), d['roofline']['phases'])
for k,v in d.get('brackets',{}).items():
    if isinstance(v,dict): print(k, v.get('value'), v.get('ms_per_step'), {a:b['ms_per_pass'] for a,b in v.get('roofline',{}).get('phases',{}).items()})
c=d.get('c5',{}); print('c5', c.get('value'), c.get('ms_per_step'), {a:b['ms_per_pass'] for a,b in c.get('roofline',{}).get('phases',{}).items()})
PY
( timeout 600 python tools/related_bench.py ) > gpurun_out/p_related.json 2> gpurun_out/p_related.err; cat gpurun_out/p_related.json | cut -c1-1500
