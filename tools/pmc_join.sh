#!/bin/bash
# SQ counters of ONE kernel of a leg (default: the join engine's tile kernel on the one_species bracket) -- kernel stats, then
# the counter groups, each pass its own run.   LEG=one_clade KERNEL=dn_pairs tools/pmc_join.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
LEG=${LEG:-one_species}
KERNEL=${KERNEL:-jn_tile}
cd /tmp && export TMPDIR=/tmp
run() {
    local name=$1; shift
    timeout 600 rocprofv3 "$@" --output-format csv -d "$OUT/${TAG:-r06}_sq_${LEG}_${name}" -o p -- python $ROOT/tools/prof_leg.py --leg $LEG --steps ${STEPS:-3} ${COLD:-} > "$OUT/${TAG:-r06}_sq_${LEG}_${name}.log" 2>&1
    echo "$name rc=$? $(grep -o '"ms_per_step": [0-9.]*' $OUT/${TAG:-r06}_sq_${LEG}_${name}.log | head -1)"
}
run stats --kernel-trace --stats
find $OUT/${TAG:-r06}_sq_${LEG}_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG:-r06}_sq_kernel_stats_${LEG}.csv \;
rm -rf $OUT/${TAG:-r06}_sq_${LEG}_stats/*/*.db 2>/dev/null
find $OUT/${TAG:-r06}_sq_${LEG}_stats -name "*kernel_trace.csv" -delete 2>/dev/null
run sqa --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run sqb --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run sqc --pmc SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
[ "${TA:-0}" = 1 ] && run sqd --pmc TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
cd $ROOT
python - <<PY
import csv, glob, collections
rows=list(csv.DictReader(open("gpurun_out/${TAG:-r06}_sq_kernel_stats_${LEG}.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:12]: print(f'{float(r["TotalDurationNs"])/1e6:9.3f} ms total {int(r["Calls"]):5d} calls {float(r["AverageNs"])/1e6:9.3f} ms avg  {r["Name"][:90]}')
for tag in ("sqa","sqb","sqc","sqd"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob(f"gpurun_out/${TAG:-r06}_sq_${LEG}_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "${KERNEL}" not in k: continue
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in acc.items():
        print(tag, k[:40], {a: f"{b:.4g}" for a,b in v.items()})
PY
