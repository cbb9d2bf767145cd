#!/bin/bash
# first GPU contact of round 5: the tile-built index against the sorted one, then what it costs
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "index_built_by_tiles or many_genome_sizes or leading_bits or dense_groups or invalidate" > gpurun_out/r5_first_tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r5_first_tests.log
for mode in tiles sort; do
    MASHGPU_SPARSE_INDEX=$mode MASHGPU_SPARSE_DBG=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-sketch --no-screen --no-h2h --no-cli --no-brackets --no-cpu --detail gpurun_out/r5_first_bench_$mode.json > gpurun_out/r5_first_bench_$mode.log 2>&1
    echo "bench $mode rc=$?"; grep -h "index by tiles\|index of" gpurun_out/r5_first_bench_$mode.log | sort | uniq -c | head -8; tail -2 gpurun_out/r5_first_bench_$mode.log | cut -c1-1500
done
STATS_ONLY=1 STEPS=4 bash tools/profile_round5.sh c3_cold c5_cold
for t in c3_cold c5_cold; do echo "== $t"; python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/r05_kernel_stats_$t.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:22]: print(f'{float(r["TotalDurationNs"])/1e6/5:9.3f} ms/step {int(r["Calls"]):6d} calls  {r["Name"][:90]}')
PY
done
