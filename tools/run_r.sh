cd $GRAFT_REPO_ROOT
timeout 600 python tools/cli_e2e.py --genomes 12000 --len 50000 --only-sketch > gpurun_out/r_cli_sketch.json 2>/dev/null; cat gpurun_out/r_cli_sketch.json
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r_tests.log
tail -4 gpurun_out/r_tests.log
bash tools/profile_round.sh r02 2>&1 | tail -8
( time timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r_bench.json 2> gpurun_out/r_bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.load(open('gpurun_out/r_bench.json'))
print('value %.3e'%d['value'], d['ms_per_step'], d['roofline']['issue'], d['roofline']['measured_hbm_frac'])
print('c5 %.3e'%d['c5']['value'], 'sketch %.3e'%d['sketch']['value'], d['sketch']['roofline'].get('issue'))
PY
