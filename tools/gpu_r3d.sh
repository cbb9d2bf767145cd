#!/bin/bash
# round-3 GPU run D: dedupe + new tests + bench + a kernel trace of the overlapped pass
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x ; echo "rc=$?" ) > gpurun_out/d_pytest_all.log 2>&1
tail -6 gpurun_out/d_pytest_all.log
( timeout 900 python tools/related_bench.py --n 20000 --engines default,sparse,merged ) > gpurun_out/d_related.json 2> gpurun_out/d_related.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/d_related.json'))
for k,v in d['cases'].items():
    print(k, {e:('%.3g'%v[e]['pairs_per_s'], {p:v[e][p] for p in v[e] if p.endswith('_ms')}) for e in v if isinstance(v[e],dict)}, v['engines_agree'])
PY
tail -3 gpurun_out/d_related.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/d_trace -o p -- python $GRAFT_REPO_ROOT/tools/prof_leg.py --leg c3 --steps 2 > $GRAFT_REPO_ROOT/gpurun_out/d_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/d_trace/**/*kernel_trace.csv',recursive=True)
rows=[r for p in f for r in csv.DictReader(open(p))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
sel=[r for r in rows if 'sp_' in r['Kernel_Name'] or 'rocprim' in r['Kernel_Name']]
for r in sel[-24:]:
    print(r['Kernel_Name'].split('(')[0][-48:], 'q', r.get('Queue_Id'), 'start %.3f ms dur %.3f ms'%((int(r['Start_Timestamp'])-t0)/1e6,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
PY
( timeout 1200 python bench.py ) > gpurun_out/d_bench.json 2> gpurun_out/d_bench.err
tail -c 1500 gpurun_out/d_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/d_bench.json').read().strip().splitlines()[-1])
print('value %.4g ms/step %.2f'%(d['value'],d['ms_per_step']), d['config'].get('first_call_ms'))
print('phases', d['roofline'].get('phases'))
for k,v in d.get('brackets',{}).items():
    if isinstance(v,dict): print(k, v.get('value'), v.get('ms_per_step'), v.get('error'), v.get('roofline',{}).get('engine'))
print('c5', d.get('c5',{}).get('value'), d.get('c5',{}).get('error'))
print('sketch', d.get('sketch',{}).get('value'), 'screen', d.get('screen',{}).get('value'), d.get('screen',{}).get('error'))
print('h2h', d.get('host_to_host'))
PY
