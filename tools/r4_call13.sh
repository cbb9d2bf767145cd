#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/r04_dn_sq -o p -- python $ROOT/tools/prof_leg.py --leg c3 --steps 3 > $OUT/r04_dn_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/r04_dn_sq2 -o p -- python $ROOT/tools/prof_leg.py --leg c3 --steps 3 > $OUT/r04_dn_sq2.log 2>&1
cd $ROOT
python - <<'PY'
import csv,glob,collections
for d in ('gpurun_out/r04_dn_sq','gpurun_out/r04_dn_sq2'):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for p in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(p)):
            n=r['Kernel_Name']
            for key in ('dn_pairs','sp_fill_value','sp_discover','dn_encode'):
                if key in n:
                    acc[key][r['Counter_Name']]+=float(r['Counter_Value'])
                    if r['Counter_Name']=='GRBM_GUI_ACTIVE': cnt[key]+=1
    for k,v in acc.items(): print(d[-6:],k,cnt[k],{a:round(b/max(cnt[k],1)) for a,b in v.items()})
PY
