#!/bin/bash
# tools/r4_prof.sh <leg> [cold] -- kernel trace + stats of one leg (tools/prof_leg.py) -> gpurun_out/r04_kernel_stats_<leg>[_cold].csv
leg=$1; cold=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
tag=$leg${cold:+_cold}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_${tag}_stats -o p -- python $ROOT/tools/prof_leg.py --leg $leg --steps 4 ${cold:+--cold} > $OUT/r04_${tag}_stats.log 2>&1
echo "$tag rc=$? $(grep -o '"ms_per_step": [0-9.]*' $OUT/r04_${tag}_stats.log | head -1)"
find $OUT/r04_${tag}_stats -name "*kernel_stats.csv" -exec cp {} $OUT/r04_kernel_stats_${tag}.csv \;
python - <<PY
import csv
for r in csv.DictReader(open("$OUT/r04_kernel_stats_${tag}.csv")):
    n=r['Name']
    if 'at::native' in n or 'elementwise' in n: continue
    print(r['Calls'], round(float(r['AverageNs'])/1e6,3), round(float(r['TotalDurationNs'])/1e6,3), n[:110])
PY
