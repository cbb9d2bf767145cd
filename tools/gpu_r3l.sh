#!/bin/bash
# round-3 GPU run L: constant-memory reads mode, resident screen database + sparse hits + two-tier bound
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "screen or reads or target_coverage or bloom or streamed" ; echo "rc=$?" ) > gpurun_out/l_tests.log 2>&1; tail -5 gpurun_out/l_tests.log
( timeout 900 python -m pytest tests/test_cli.py tests/test_abi_cpu.py -q -x ; echo "rc=$?" ) > gpurun_out/l_cli_tests.log 2>&1; tail -3 gpurun_out/l_cli_tests.log
( timeout 300 python tests/fuzz_sketch.py --n 100000 --seconds 45 --seed 11 ) > gpurun_out/l_sketch_fuzz.txt 2>&1; tail -2 gpurun_out/l_sketch_fuzz.txt
( timeout 300 python tests/fuzz_cli.py --n 100000 --seconds 45 --seed 303 ) > gpurun_out/l_cli_fuzz.txt 2>&1; tail -2 gpurun_out/l_cli_fuzz.txt
( timeout 600 python tools/reads_e2e.py ) > gpurun_out/l_reads_e2e.json 2> gpurun_out/l_reads_e2e.err; cat gpurun_out/l_reads_e2e.json; tail -3 gpurun_out/l_reads_e2e.err
( MASH_AMD_EARLY_PARSE=1 timeout 600 python tools/sketch_e2e.py --genomes 300 --len 4000000 --reps 3 ) > gpurun_out/l_bact_early.json 2>&1; cat gpurun_out/l_bact_early.json
( timeout 600 python tools/sketch_e2e.py --genomes 300 --len 4000000 --reps 3 ) > gpurun_out/l_bact.json 2>&1; cat gpurun_out/l_bact.json
( timeout 900 python bench.py --no-c5 --no-h2h --no-cpu --no-sketch --no-cli ) > gpurun_out/l_bench.json 2> gpurun_out/l_bench.err; tail -c 400 gpurun_out/l_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/l_bench.json') if l.startswith('{')][-1])
print('c3', d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('traffic'), d['roofline']['pass'])
s=d.get('screen',{})
print('screen', {k:v for k,v in s.items() if k not in ('config','roofline')})
PY
