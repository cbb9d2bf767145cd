#!/bin/bash
# round-3 GPU run M: screen phases after the planning fix; fill {c,c} for tables of copies; ctx timing
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "screen or compare or identical or sparse or triangle" ; echo "rc=$?" ) > gpurun_out/m_tests.log 2>&1; tail -4 gpurun_out/m_tests.log
( timeout 600 python tools/prof_leg.py --leg screen --steps 4 ) > gpurun_out/m_screen.json 2> gpurun_out/m_screen.err; cat gpurun_out/m_screen.json; tail -2 gpurun_out/m_screen.err
( MASHGPU_SCREEN_TIERS=1 timeout 600 python tools/prof_leg.py --leg screen --steps 4 ) > gpurun_out/m_screen_t1.json 2> gpurun_out/m_screen_t1.err; cat gpurun_out/m_screen_t1.json
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/m_screen_stats -o p -- python $GRAFT_REPO_ROOT/tools/prof_leg.py --leg screen --steps 4 ) > gpurun_out/m_screen_stats.log 2>&1
find gpurun_out/m_screen_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/m_kernel_stats_screen.csv \;
head -14 gpurun_out/m_kernel_stats_screen.csv | cut -c1-160
for i in 1 2 3; do python tools/ctx_timing.py; done 2>&1 | grep "^{" 
( timeout 600 python bench.py --no-c5 --no-h2h --no-cpu --no-sketch --no-cli --no-screen ) > gpurun_out/m_bench.json 2> gpurun_out/m_bench.err; tail -c 300 gpurun_out/m_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/m_bench.json') if l.startswith('{')][-1])
print('c3', d['value'], d['ms_per_step'])
for k,v in d.get('brackets',{}).items():
    if isinstance(v,dict): print(k, v.get('value'), v.get('ms_per_step'), v.get('roofline',{}).get('phases'))
PY
