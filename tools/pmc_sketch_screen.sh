#!/bin/bash
# HBM and issue counters of the sketch kernel and the screen kernel (to be run again whenever sketch.hip, kmer_hash.h or screen.hip
# change: bench.py drops the files when the sources' hash differs): PMC passes, each its own run, over bench.py's sketch leg / screen leg -> gpurun_out/{sketch,screen}_pmc_latest.json
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu --no-h2h --no-screen --no-c5 --no-cli --no-brackets --steps 1 --warmup 0 --detail /tmp/d1.json"
BS="python $ROOT/bench.py --no-cpu --no-h2h --no-sketch --no-c5 --no-cli --no-brackets --steps 1 --warmup 0 --detail /tmp/d2.json"
run() { local name=$1; shift; rm -rf "$OUT/r05k_${name}"; timeout 300 rocprofv3 "$@" > "$OUT/r05k_${name}.log" 2>&1; echo "$name rc=$?"; }
run fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/r05k_fetch" -o p -- $B
run write --pmc WRITE_SIZE --output-format csv -d "$OUT/r05k_write" -o p -- $B
run sqa --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/r05k_sqa" -o p -- $B
run sqb --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/r05k_sqb" -o p -- $B
run s_fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/r05k_s_fetch" -o p -- $BS
run s_write --pmc WRITE_SIZE --output-format csv -d "$OUT/r05k_s_write" -o p -- $BS
cd $ROOT
python tools/make_pmc_json.py gpurun_out/r05k_ sketch_chunks 29999400000 kmer gpurun_out/sketch_pmc_latest.json mash_amd/csrc/sketch.hip mash_amd/csrc/kmer_hash.h | cut -c1-700
python tools/make_pmc_json.py gpurun_out/r05k_s_ "256, true>" 30000000 read gpurun_out/screen_pmc_latest.json mash_amd/csrc/sketch.hip mash_amd/csrc/kmer_hash.h mash_amd/csrc/screen.hip | cut -c1-500
rm -rf $OUT/r05k_fetch $OUT/r05k_write $OUT/r05k_sqa $OUT/r05k_sqb $OUT/r05k_s_fetch $OUT/r05k_s_write
