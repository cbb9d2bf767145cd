cd $GRAFT_REPO_ROOT
timeout 600 python tools/related_bench.py --n 20000 > gpurun_out/l_related.json 2> gpurun_out/l_related.err
python -c "
import json;d=json.load(open('gpurun_out/l_related.json'))
for k,v in d['cases'].items(): print(k, {e:('%.3e'%x['pairs_per_s']) for e,x in v.items() if isinstance(x,dict)}, v['engines_agree'], v['mean_shared'])"
tail -c 300 gpurun_out/l_related.err
