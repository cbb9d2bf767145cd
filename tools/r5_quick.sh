#!/bin/bash
# quick look: per-table C3 (+ C5) with the kernels' times
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
LEGS=${LEGS:-c3_cold c5_cold}
if [ -n "${TESTS:-}" ]; then
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$TESTS" > gpurun_out/r5_quick_tests.log 2>&1
    echo "tests rc=$?"; tail -4 gpurun_out/r5_quick_tests.log
fi
STATS_ONLY=1 STEPS=4 bash tools/profile_round5.sh $LEGS
for t in $LEGS; do echo "== $t"; python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/r05_kernel_stats_$t.csv")))
rows=[r for r in rows if "at::native" not in r["Name"] and "rocclr" not in r["Name"] and "segmented_sort" not in r["Name"]]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows if "mg::ix_" in r["Name"])/1e6/5
print(f"ix_ kernels: {tot:.3f} ms/step; all mg/rocprim kernels: {sum(float(r['TotalDurationNs']) for r in rows)/1e6/5:.3f} ms/step")
for r in rows[:${TOP:-18}]: print(f'{float(r["TotalDurationNs"])/1e6/5:9.3f} ms/step {int(r["Calls"]):6d} calls  {r["Name"][:80]}')
PY
done
