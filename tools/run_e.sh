ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu --no-sketch --no-screen --steps 1 --warmup 0"
run() { local name=$1; shift; timeout 300 rocprofv3 "$@" > "$OUT/e_${name}.log" 2>&1; echo "$name rc=$?"; }
run stats --kernel-trace --stats --output-format csv -d "$OUT/e_stats" -o p -- $B
run sqa --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/e_sqa" -o p -- $B
run sqb --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/e_sqb" -o p -- $B
run sqc --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM --output-format csv -d "$OUT/e_sqc" -o p -- $B
run fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/e_fetch" -o p -- $B
run write --pmc WRITE_SIZE --output-format csv -d "$OUT/e_write" -o p -- $B
cd $ROOT
python - <<'PY'
import csv, glob, collections
for d in ("sqa","sqb","sqc","fetch","write"):
    for p in glob.glob(f"gpurun_out/e_{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            if "compare_merged" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for c, v in sorted(agg.items()):
            print(d, c, " ".join("%.4g" % x for _, x in sorted(v)))
for p in glob.glob("gpurun_out/e_stats/**/*kernel_stats.csv", recursive=True):
    print(open(p).read()[:1500])
PY
