#!/bin/bash
# Per-dispatch durations of the windowed compare launches (kernel trace of one sweep run).
#   tools/window_trace.sh [n] [s]   -> prints the durations (ms) of the last call's window launches
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-6000}; S=${2:-10000}
OUT=$ROOT/gpurun_out/wintrace
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python $ROOT/tools/sweep_compare.py --n $N --s $S --rounds 1 win > "$OUT/log.txt" 2>&1
tail -3 "$OUT/log.txt"
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "compare_merged_kernel" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print(len(d), "launches; last call per window (ms):")
half = len(d) // 2
print(" ".join("%.2f" % x for x in d[half:]))
print("sum %.2f ms" % sum(d[half:]))
PY
