#!/bin/bash
# Round-end recipe on the GPU box (through gpurun): profile passes of the round, the stamped PMC
# summaries put where bench.py reads them, one default bench line, the whole GPU test suite.
#   gpurun --timeout 2700 -- 'bash tools/round_end.sh r02'
# Copy afterwards: gpurun_out/<tag>_{compare,sketch,screen}_pmc.json -> profiles/*_pmc_latest.json,
# <tag>_kernel_stats.csv, <tag>_pmc_digest.txt, round_bench.json -> profiles/<tag>_bench_full.json.
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh $TAG 2>&1 | tail -8
cp gpurun_out/${TAG}_compare_pmc.json profiles/compare_pmc_latest.json
cp gpurun_out/${TAG}_sketch_pmc.json profiles/sketch_pmc_latest.json
cp gpurun_out/${TAG}_screen_pmc.json profiles/screen_pmc_latest.json
( time timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/round_bench.json 2> gpurun_out/round_bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/round_bench.json').read().strip().splitlines()[-1])
print('value %.3e'%d['value'], d['ms_per_step'], d['roofline']['issue'], d['roofline']['measured_hbm_frac'], d['roofline']['traffic'])
PY
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/round_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/round_tests.log
tail -4 gpurun_out/round_tests.log
