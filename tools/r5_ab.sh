#!/bin/bash
# tests (TESTS=...), then per-table time of the legs (LEGS, REPS) without a profiler, then one profiled cold c3 step's timeline
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
if [ -n "${TESTS:-}" ]; then
    timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$TESTS" > gpurun_out/r5_ab_tests.log 2>&1
    echo "tests rc=$?"; tail -3 gpurun_out/r5_ab_tests.log
fi
for leg in ${LEGS:-c3 c5}; do
    for rep in $(seq ${REPS:-3}); do
        timeout 300 python tools/prof_leg.py --leg $leg --steps 6 --cold > gpurun_out/r5_ab_${leg}.log 2>&1
        echo "$leg rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5_ab_${leg}.log | head -1)"
    done
done
STATS_ONLY=1 STEPS=4 bash tools/profile_round5.sh c3_cold
python tools/r5_timeline.py gpurun_out/r05_c3_cold_stats/p_kernel_trace.csv > gpurun_out/r5_timeline_c3.txt 2>&1
head -${LINES_TL:-48} gpurun_out/r5_timeline_c3.txt
