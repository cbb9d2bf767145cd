cd $GRAFT_REPO_ROOT
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; TAG=v
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu --no-h2h --no-screen --no-c5 --no-sketch --steps 1 --warmup 0"
run() { local name=$1; shift; timeout 600 rocprofv3 "$@" > "$OUT/${TAG}_${name}.log" 2>&1; echo "$name rc=$?"; }
run sqa --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${TAG}_sqa" -o p -- $B
run sqb --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/${TAG}_sqb" -o p -- $B
cd $ROOT
python tools/pmc_digest.py gpurun_out/${TAG}_sqa gpurun_out/${TAG}_sqb --kernels=compare_merged > gpurun_out/${TAG}_pmc_digest.txt
cat gpurun_out/${TAG}_pmc_digest.txt
