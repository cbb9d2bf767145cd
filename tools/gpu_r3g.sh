#!/bin/bash
# round-3 GPU run G: discover batching, fill variant, start-up timing, CLI e2e
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_refcli_gpu.py -m gpu -q -x ; echo "rc=$?" ) > gpurun_out/g_pytest.log 2>&1
tail -5 gpurun_out/g_pytest.log
V="fillwave:MASHGPU_SPARSE_FILL_MODE=1;fillwave8:MASHGPU_SPARSE_FILL_MODE=1,MASHGPU_SPARSE_FILL_BPC=8;bpc8:MASHGPU_SPARSE_FILL_BPC=8;bpc32:MASHGPU_SPARSE_FILL_BPC=32"
( timeout 600 python tools/sparse_probe.py --n 100000 --no-dense --variants "$V" ) > gpurun_out/g_probe_c3.json 2> gpurun_out/g_probe_c3.err
cat > /tmp/show.py <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if isinstance(v,dict) and 'ms_per_step' in v:
        print(k, 'cold',round(v['cold_ms'],1),'step', round(v['ms_per_step'],2), 'ms', '%.3g'%v['pairs_per_s'], {p:round(v[p]['avg_ms'],2) for p in v if p.startswith('compare')}, v['checksum_after_steps'])
PY
python /tmp/show.py gpurun_out/g_probe_c3.json; tail -2 gpurun_out/g_probe_c3.err
for i in 1 2 3; do python tools/ctx_timing.py 2>/dev/null | tail -1; done > gpurun_out/g_ctx_timing.txt; cat gpurun_out/g_ctx_timing.txt
( timeout 600 python tools/cli_e2e.py --genomes 12000 --len 50000 --threads 16 --only-sketch ) > gpurun_out/g_cli_sketch.json 2> gpurun_out/g_cli_sketch.err; cat gpurun_out/g_cli_sketch.json; tail -2 gpurun_out/g_cli_sketch.err
nproc
