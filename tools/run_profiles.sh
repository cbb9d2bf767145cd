#!/bin/bash
# Profile passes over bench.py on the GPU box (run through gpurun): kernel stats, HBM traffic
# counters (separate --pmc passes, never combined with tracing domains), SQ activity counters.
#   tools/run_profiles.sh <tag>      -> gpurun_out/<tag>_{stats,fetch,write,sqa,sqb}/
set -u
TAG=${1:-r01}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --no-cpu"
run() { # name, rocprof args..., -- bench args
  local name=$1; shift
  timeout 600 rocprofv3 "$@" > "$OUT/${TAG}_${name}.log" 2>&1
  echo "$name rc=$?"
}
run stats --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_stats" -o p -- $B --steps 2 --warmup 1
run fetch --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_fetch" -o p -- $B --no-screen --steps 1 --warmup 0
run write --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_write" -o p -- $B --no-screen --steps 1 --warmup 0
run sqa --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${TAG}_sqa" -o p -- $B --no-screen --steps 1 --warmup 0
run sqb --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/${TAG}_sqb" -o p -- $B --no-screen --steps 1 --warmup 0
ls "$OUT" | grep "^${TAG}_"
