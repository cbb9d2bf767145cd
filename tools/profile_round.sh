#!/bin/bash
# Profile passes of the round over bench.py on the GPU box (run through gpurun):
#   tools/profile_round.sh <tag>     e.g. r02  -> gpurun_out/<tag>_{stats,fetch,write,sqa,sqb}/ + digests
# kernel stats of a whole default bench run; PMC passes (each its own run, never combined with tracing
# domains) over ONE compare step / ONE sketch step.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu --no-h2h --no-screen --no-c5 --steps 1 --warmup 0"
run() { local name=$1; shift; timeout 600 rocprofv3 "$@" > "$OUT/${TAG}_${name}.log" 2>&1; echo "$name rc=$?"; }
run stats --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -o p -- python $ROOT/bench.py --no-cpu --no-h2h --steps 3 --warmup 1
run fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_fetch" -o p -- $B
run write --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_write" -o p -- $B
run sqa --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${TAG}_sqa" -o p -- $B
run sqb --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/${TAG}_sqb" -o p -- $B
# the screen leg (C4): HBM traffic of its kernels over warm-up + 2 steps = 3 x 10^7 reads
BS="python $ROOT/bench.py --no-cpu --no-h2h --no-sketch --no-c5 --steps 1 --warmup 0"
run sfetch --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_s_fetch" -o p -- $BS
run swrite --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_s_write" -o p -- $BS
cd $ROOT
python tools/make_pmc_json.py gpurun_out/${TAG}_s_ "256, true>" 30000000 read gpurun_out/${TAG}_screen_pmc.json mash_amd/csrc/sketch.hip mash_amd/csrc/kmer_hash.h mash_amd/csrc/screen.hip
python tools/make_pmc_json.py gpurun_out/${TAG}_ compare_merged 4999950000 pair gpurun_out/${TAG}_compare_pmc.json mash_amd/csrc/compare_merged.hip mash_amd/csrc/compare_internal.h
# the sketch leg of that run: warm-up + 2 timed steps = 3 launches of 10^4 x (10^6 - 20) k-mers
python tools/make_pmc_json.py gpurun_out/${TAG}_ sketch_chunks 29999400000 kmer gpurun_out/${TAG}_sketch_pmc.json mash_amd/csrc/sketch.hip mash_amd/csrc/kmer_hash.h
python tools/pmc_digest.py gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_sqa gpurun_out/${TAG}_sqb --kernels=compare_merged,sketch_chunks,merge_chunks,finish > gpurun_out/${TAG}_pmc_digest.txt
find gpurun_out/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
head -12 gpurun_out/${TAG}_kernel_stats.csv
