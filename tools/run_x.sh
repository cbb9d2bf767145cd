cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r02 2>&1 | tail -8
cp gpurun_out/r02_compare_pmc.json profiles/compare_pmc_latest.json
cp gpurun_out/r02_sketch_pmc.json profiles/sketch_pmc_latest.json
cp gpurun_out/r02_screen_pmc.json profiles/screen_pmc_latest.json
( time timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/x_bench.json 2> gpurun_out/x_bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/x_bench.json').read().strip().splitlines()[-1])
print('value %.3e'%d['value'], d['ms_per_step'], d['roofline']['issue'], d['roofline']['measured_hbm_frac'], d['roofline']['traffic'])
PY
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/x_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/x_tests.log
tail -4 gpurun_out/x_tests.log
