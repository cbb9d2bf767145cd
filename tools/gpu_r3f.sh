#!/bin/bash
# round-3 GPU run F: merge fast path, discover visiting order, refcli-gpu test
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x ; echo "rc=$?" ) > gpurun_out/f_pytest_all.log 2>&1
tail -6 gpurun_out/f_pytest_all.log
V="noorder:MASHGPU_SPARSE_NO_ORDER=1;lanes:MASHGPU_SPARSE_MERGE=lanes"
( MASHGPU_SPARSE_DBG=1 timeout 600 python tools/sparse_probe.py --n 100000 --no-dense --variants "$V" ) > gpurun_out/f_probe_c3.json 2> gpurun_out/f_probe_c3.err
cat > /tmp/show.py <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if isinstance(v,dict) and 'ms_per_step' in v:
        print(k, 'cold',round(v['cold_ms'],1),'step', round(v['ms_per_step'],2), 'ms', '%.3g'%v['pairs_per_s'], {p:round(v[p]['avg_ms'],2) for p in v if p.startswith('compare')}, v['checksum_after_steps'])
PY
python /tmp/show.py gpurun_out/f_probe_c3.json; tail -2 gpurun_out/f_probe_c3.err
( timeout 900 python tools/related_bench.py --n 20000 --engines default,merged ) > gpurun_out/f_related.json 2> gpurun_out/f_related.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/f_related.json'))
for k,v in d['cases'].items():
    print(k, {e:('%.3g'%v[e]['pairs_per_s'], {p:v[e][p] for p in v[e] if p.endswith('_ms')}) for e in v if isinstance(v[e],dict)}, v['engines_agree'])
PY
tail -3 gpurun_out/f_related.err
( timeout 1200 python bench.py --no-sketch --no-screen --no-h2h --no-cpu ) > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err
tail -c 400 gpurun_out/f_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/f_bench.json').read().strip().splitlines()[-1])
print('value %.4g ms/step %.2f'%(d['value'],d['ms_per_step']), d['config'].get('first_call_ms'))
print('phases', {k:v['ms_per_pass'] for k,v in d['roofline'].get('phases',{}).items()})
for k,v in d.get('brackets',{}).items():
    if isinstance(v,dict): print(k, '%.4g'%v.get('value',0), v.get('ms_per_step'), v.get('error'), {p:q['ms_per_pass'] for p,q in v.get('roofline',{}).get('phases',{}).items()})
print('c5', d.get('c5',{}).get('value'), d.get('c5',{}).get('error'), {p:q['ms_per_pass'] for p,q in d.get('c5',{}).get('roofline',{}).get('phases',{}).items()})
PY
