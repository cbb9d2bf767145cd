#!/bin/bash
# the round's closing run on one box: every GPU test, the smoke entry, the profile legs (kernel statistics + HBM counters),
# then the bench line with the PMC files of THIS tree
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_tests.log 2>&1
echo "gpu tests rc=$? $(tail -1 gpurun_out/r05_gpu_tests.log)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 bash tools/profile_round5.sh c3_cold c3 c5_cold c5 clades one_clade random identical 2>&1 | tail -30
cp gpurun_out/compare_*_pmc.json profiles/ 2>/dev/null
timeout 1200 python bench.py --detail gpurun_out/r05_bench_detail.json > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_stderr.log
echo "bench rc=$?"; cut -c1-3000 gpurun_out/r05_bench_line.json
