#!/bin/bash
# counters of the index-build kernels on the per-table C3 leg (each --pmc set its own run)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
LEG=${1:-c3}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rm -rf $OUT/r05_ixpmc_$i
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/r05_ixpmc_$i -o p -- python $ROOT/tools/prof_leg.py --leg $LEG --steps 3 --cold > $OUT/r05_ixpmc_$i.log 2>&1
    echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for i in (1, 2, 3, 4):
    for p in glob.glob("$OUT/r05_ixpmc_%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not any(s in k for s in ("ix_", "dn_leaders", "dn_encode", "cl_gather", "sp_row_digest", "sp_fill_value")):
                continue
            name = r["Counter_Name"]
            if name == "GRBM_GUI_ACTIVE" and i == 2: name = "GUI2"
            acc[k][name] += float(r["Counter_Value"])
            if name == "FETCH_SIZE": calls[k] += 1
for k, c in sorted(acc.items()):
    n = max(calls[k], 1)
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8 / n
    cyc2 = c.get("GUI2", 0) / 8 / n
    g = lambda x: c.get(x, 0) / n
    print(f"{k[:44]:44s} launches {n:3d} cyc {cyc:9.0f} ({cyc/2.4e6:6.3f} ms@2.4GHz) waves {g('SQ_WAVES'):9.0f} "
          f"valu/simd-cyc {g('SQ_INSTS_VALU')*4/1024/max(cyc,1):5.3f} salu {g('SQ_INSTS_SALU')/1024/max(cyc,1):5.3f} ldsinst {g('SQ_INSTS_LDS')/1024/max(cyc,1):5.3f} "
          f"lds_active {g('SQ_LDS_IDX_ACTIVE')/256/max(cyc2,1):5.3f} conflict_share {g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_LDS_IDX_ACTIVE'),1):5.3f} "
          f"wave_cyc/cu-cyc {g('SQ_WAVE_CYCLES')/256/max(cyc,1):6.2f} wait_inst {g('SQ_WAIT_INST_ANY')/max(g('SQ_WAVE_CYCLES'),1):5.3f} wait_any {g('SQ_WAIT_ANY')/max(g('SQ_WAVE_CYCLES'),1):5.3f} "
          f"read {2*g('FETCH_SIZE')*1024/1e9:6.3f} GB write {g('WRITE_SIZE')*1024/1e9:6.3f} GB")
PY
