#!/bin/bash
# round-3 GPU run C: probe variants (fixed), related bench with phases, new tests
mkdir -p gpurun_out
V="seq:MASHGPU_SPARSE_OVERLAP=0;seq_lanes:MASHGPU_SPARSE_OVERLAP=0,MASHGPU_SPARSE_MERGE=lanes;bpc2:MASHGPU_SPARSE_FILL_BPC=2;bpc8:MASHGPU_SPARSE_FILL_BPC=8;bpc16:MASHGPU_SPARSE_FILL_BPC=16"
( MASHGPU_SPARSE_DBG=1 timeout 600 python tools/sparse_probe.py --n 100000 --no-dense --variants "$V" ) > gpurun_out/c_probe_c3.json 2> gpurun_out/c_probe_c3.err
cat > /tmp/show.py <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if isinstance(v,dict) and 'ms_per_step' in v:
        print(k, 'cold',round(v['cold_ms'],1),'step', round(v['ms_per_step'],2), 'ms', '%.3g'%v['pairs_per_s'], {p:round(v[p]['avg_ms'],2) for p in v if p.startswith('compare')}, v['checksum_after_steps'])
PY
python /tmp/show.py gpurun_out/c_probe_c3.json; tail -2 gpurun_out/c_probe_c3.err
( MASHGPU_SPARSE_DBG=1 timeout 900 python tools/sparse_probe.py --n 100000 --s 10000 --steps 2 --variants "seq:MASHGPU_SPARSE_OVERLAP=0" ) > gpurun_out/c_probe_c5.json 2> gpurun_out/c_probe_c5.err
python /tmp/show.py gpurun_out/c_probe_c5.json; tail -3 gpurun_out/c_probe_c5.err
( timeout 900 python tools/related_bench.py --n 20000 --engines default,sparse,merged ) > gpurun_out/c_related.json 2> gpurun_out/c_related.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/c_related.json'))
for k,v in d['cases'].items():
    print(k, {e:('%.3g'%v[e]['pairs_per_s'], {p:v[e][p] for p in v[e] if p.endswith('_ms')}) for e in v if isinstance(v[e],dict)}, v['engines_agree'])
PY
( timeout 1500 python -m pytest tests -m gpu -q -x ; echo "rc=$?" ) > gpurun_out/c_pytest_all.log 2>&1
tail -4 gpurun_out/c_pytest_all.log
