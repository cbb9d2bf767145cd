#!/bin/bash
# A round's profile passes, one leg at a time (tools/prof_leg.py), through gpurun (TAG=r06: the prefix of every file):
#   kernel trace + stats per leg  -> gpurun_out/<TAG>_<leg>_stats/  -> gpurun_out/<TAG>_kernel_stats_<leg>.csv
#   PMC passes (each its own run, never combined with tracing domains): FETCH_SIZE, WRITE_SIZE for every leg;
#   SQ instruction / LDS counters for the headline leg (c3_cold).
# A leg named <x>_cold runs every step from an invalidated table (the per-table job: index build included).
# usage: TAG=r06 tools/profile_round.sh [legs...]      default: c3_cold c3 c5_cold c5
#        legs: c3 c5 random identical clades one_clade one_species, each also as <leg>_cold
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
LEGS=${@:-c3_cold c3 c5_cold c5}
STEPS=${STEPS:-4}
TAG=${TAG:-r06}
cd /tmp && export TMPDIR=/tmp
run() {
    local tag=$1 name=$2; shift 2
    local leg=${tag%_cold} cold=""; [ "$leg" != "$tag" ] && cold="--cold"
    timeout 600 rocprofv3 "$@" --output-format csv -d "$OUT/${TAG}_${tag}_${name}" -o p -- python $ROOT/tools/prof_leg.py --leg $leg --steps $STEPS $cold > "$OUT/${TAG}_${tag}_${name}.log" 2>&1
    echo "$tag $name rc=$? $(grep -o '"ms_per_step": [0-9.]*' $OUT/${TAG}_${tag}_${name}.log | head -1)"
}
for tag in $LEGS; do
    run $tag stats --kernel-trace --stats
    find $OUT/${TAG}_${tag}_stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats_${tag}.csv \;
    [ "${STATS_ONLY:-0}" = 1 ] && continue
    run $tag fetch --pmc FETCH_SIZE
    run $tag write --pmc WRITE_SIZE
    if [ $tag = c3_cold ] && [ "${NO_SQ:-0}" != 1 ]; then
        run $tag sqa --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
        run $tag sqb --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
    fi
    rm -rf $OUT/${TAG}_${tag}_stats/*/*.db 2>/dev/null
done
# (gpurun brings back 64 MiB at most: the kernel traces -- a row per launch -- stay on the box, the per-kernel statistics travel)
find $OUT -name "*kernel_trace.csv" -path "*${TAG}_*" -delete 2>/dev/null
[ "${STATS_ONLY:-0}" = 1 ] && exit 0
cd $ROOT
SRC="mash_amd/csrc/compare_sparse.hip mash_amd/csrc/compare_dense.hip mash_amd/csrc/compare_merged.hip mash_amd/csrc/compare_internal.h mash_amd/csrc/index_build.hip mash_amd/csrc/compare_join.hip"
PASS="jn_tile_kernel,sp_fill_value,sp_fill_chunks,sp_fill_short,sp_class_pairs,sp_discover_kernel,sp_chunks,sp_pack_costs,sp_merge_pack,sp_merge_rows,sp_merge_kernel,sp_scatter_kernel,dn_pairs,compare_merged"
BUILD="mg::ix_,jn_emit,jn_heads,jn_groups,jn_gend,jn_levels,jn_labels,jn_jump,jn_order,jn_shared,sp_fill_entries,sp_tie_,sp_heads,sp_index_scatter,sp_stat_reduce,sp_fill_u32,sp_row_digest,sp_dup_flags,sp_row_equal,sp_row_key,sp_order_from_keys,sp_entry_counts,sp_offsets,dn_group_rows,row_classes,cl_emit,cl_minrow,cl_jump,cl_order_keys,cl_split_keys,cl_gather_rows,dn_neighbor,dn_leader,dn_sublists,dn_universe,dn_encode,ROCPRIM_400200"
for tag in $LEGS; do
    leg=${tag%_cold}
    case $leg in one_clade|one_species) PAIRS=536854528 ;; *) PAIRS=4999950000 ;; esac
    # a cold leg builds the index in every pass: its kernels belong to the pass; a warm leg builds it once (reported per run, outside the totals)
    if [ "$leg" != "$tag" ]; then P="$PASS,$BUILD"; C=""; else P="$PASS"; C="$BUILD"; fi
    python tools/make_pmc_json3.py gpurun_out/${TAG}_${tag}_ $((STEPS + 1)) $PAIRS pair gpurun_out/compare_${tag}_pmc.json "$P" "$C" $SRC > gpurun_out/${TAG}_${tag}_pmc.txt 2>&1
    tail -1 gpurun_out/${TAG}_${tag}_pmc.txt
done
