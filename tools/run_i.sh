cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/ -x -q -m gpu -k "sharded or rank_communicator or test_cli" > gpurun_out/i_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/i_tests.log
tail -5 gpurun_out/i_tests.log
( time timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/i_bench.json 2> gpurun_out/i_bench.err ) 2>&1 | tail -3
tail -c 300 gpurun_out/i_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/i_bench.json'))
print('value %.3e'%d['value'], d['ms_per_step'], d['config'].get('output_checksum'))
for k in ('host_to_host','c5'):
    print(k, json.dumps(d.get(k))[:900])
s=d.get('sketch',{})
print('sketch %.3e'%s.get('value',0), {k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('value','cores','sample','error',1,16,'1','16')}) for k,v in s.items() if k.startswith('cpu') or k=='host_to_host'})
print('screen', d.get('screen',{}).get('value'), d.get('screen',{}).get('error'))
PY
bash tools/profile_round.sh r02 2>&1 | tail -20
