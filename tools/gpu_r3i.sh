#!/bin/bash
# round-3 GPU run I: after the engine clean-up and the two merge kernels: tests, fuzzers, profile round, bench
mkdir -p gpurun_out
( timeout 1800 python -m pytest tests -m gpu -q -x ; echo "rc=$?" ) > gpurun_out/i_pytest_all.log 2>&1
tail -4 gpurun_out/i_pytest_all.log
( timeout 300 python tools/compare_fuzz.py --n 100000 --seconds 90 --seed 31 ) > gpurun_out/r03_compare_fuzz.txt 2>&1; tail -3 gpurun_out/r03_compare_fuzz.txt
( timeout 500 python tests/fuzz_cli.py --n 100000 --seconds 120 --seed 77; timeout 300 python tests/fuzz_cli.py --n 100000 --seconds 60 --seed 78 --env MASH_GPU_DEVICES=0,0 ) > gpurun_out/r03_cli_fuzz.txt 2>&1; tail -3 gpurun_out/r03_cli_fuzz.txt
( timeout 300 python tests/fuzz_sketch.py --n 100000 --seconds 60 --seed 5 ) > gpurun_out/r03_sketch_fuzz.txt 2>&1; tail -2 gpurun_out/r03_sketch_fuzz.txt
bash tools/profile_round3.sh > gpurun_out/i_profile.log 2>&1; grep -E "rc=|total" gpurun_out/i_profile.log
( timeout 1500 python bench.py ) > gpurun_out/i_bench.json 2> gpurun_out/i_bench.err
tail -c 300 gpurun_out/i_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/i_bench.json').read().strip().splitlines()[-1])
print('value %.4g ms/step %.2f'%(d['value'],d['ms_per_step']), d['config'].get('first_call_ms'))
print('phases', {k:v['ms_per_pass'] for k,v in d['roofline'].get('phases',{}).items()}, 'traffic', d['roofline'].get('traffic'), d['roofline'].get('traffic_over_compulsory'))
for k,v in d.get('brackets',{}).items():
    if isinstance(v,dict): print(k, '%.4g'%v.get('value',0), v.get('ms_per_step'), v.get('error'), {p:q['ms_per_pass'] for p,q in v.get('roofline',{}).get('phases',{}).items()})
print('c5', d.get('c5',{}).get('value'), d.get('c5',{}).get('error'), {p:q['ms_per_pass'] for p,q in d.get('c5',{}).get('roofline',{}).get('phases',{}).items()})
print('sketch', d.get('sketch',{}).get('value'), 'screen', d.get('screen',{}).get('value'), d.get('screen',{}).get('error'))
print('h2h', {k:(v.get('value'), v.get('ms')) for k,v in d.get('host_to_host',{}).items() if isinstance(v,dict)})
PY
