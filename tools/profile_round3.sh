#!/bin/bash
# Round-3 profile passes, one leg at a time (tools/prof_leg.py), through gpurun:
#   kernel trace + stats per leg  -> gpurun_out/r03_<leg>_stats/  -> profiles/r03_kernel_stats_<leg>.csv
#   PMC passes (each its own run, never combined with tracing domains): FETCH_SIZE, WRITE_SIZE for every
#   compare leg; SQ instruction / LDS counters for c3.
# usage: tools/profile_round3.sh [legs...]      default: c3 c5 random identical clades sketch screen
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
LEGS=${@:-c3 c5 random identical clades sketch screen}
STEPS=6
cd /tmp && export TMPDIR=/tmp
run() { local leg=$1 name=$2; shift 2; timeout 900 rocprofv3 "$@" --output-format csv -d "$OUT/r03_${leg}_${name}" -o p -- python $ROOT/tools/prof_leg.py --leg $leg --steps $STEPS > "$OUT/r03_${leg}_${name}.log" 2>&1; echo "$leg $name rc=$? $(grep -o '"ms_per_step": [0-9.]*' $OUT/r03_${leg}_${name}.log | head -1)"; }
for leg in $LEGS; do
  run $leg stats --kernel-trace --stats
  find $OUT/r03_${leg}_stats -name "*kernel_stats.csv" -exec cp {} $OUT/r03_kernel_stats_${leg}.csv \;
  case $leg in
    sketch|screen) ;;
    *) run $leg fetch --pmc FETCH_SIZE
       run $leg write --pmc WRITE_SIZE ;;
  esac
  if [ $leg = c3 ]; then
    run $leg sqa --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
    run $leg sqb --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
  fi
done
cd $ROOT
SRC="mash_amd/csrc/compare_sparse.hip mash_amd/csrc/compare_merged.hip mash_amd/csrc/compare_internal.h"
PAIRS=4999950000
for leg in $LEGS; do
  case $leg in
    sketch|screen) ;;
    *) python tools/make_pmc_json3.py gpurun_out/r03_${leg}_ $((STEPS + 1)) $PAIRS pair gpurun_out/compare_${leg}_pmc.json "sp_fill_const,sp_fill_short,sp_discover_kernel<false,sp_chunks,sp_merge_rows,sp_merge_kernel,sp_scatter_kernel,compare_merged" "sp_fill_entries,sp_heads,sp_gstart,sp_index_scatter,sp_fill_u32,sp_row_digest,sp_row_equal,sp_discover_kernel<true,rocprim" $SRC > gpurun_out/r03_${leg}_pmc.txt 2>&1; tail -1 gpurun_out/r03_${leg}_pmc.txt ;;
  esac
  head -8 gpurun_out/r03_kernel_stats_${leg}.csv
done
