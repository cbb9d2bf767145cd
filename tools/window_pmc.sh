#!/bin/bash
# SQ counters of the windowed compare launches (one sweep call), summed over the dispatches.
#   tools/window_pmc.sh [n] [s]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-20000}; S=${2:-10000}
OUT=$ROOT/gpurun_out/winpmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/sweep_compare.py --n $N --s $S --rounds 0 win"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/a" -o p -- $CMD > "$OUT/a.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/b" -o p -- $CMD > "$OUT/b.log" 2>&1
python - "$OUT" $N <<'PY'
import csv, glob, sys, collections
out, n = sys.argv[1], int(sys.argv[2])
pairs = n * (n - 1) // 2
for sub in ("a", "b"):
    fs = glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(sub, "no counter file"); continue
    tot = collections.Counter(); disp = set()
    for r in csv.DictReader(open(fs[0])):
        if "compare_merged_kernel" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    print(sub, "dispatches", len(disp))
    for k, v in sorted(tot.items()):
        print("  %-24s %.4g   per pair %.2f" % (k, v, v / pairs))
PY
