#!/bin/bash
# tools/profile_screen.sh -- HBM traffic of the screen leg (C4): two PMC passes (FETCH_SIZE, WRITE_SIZE, each its own run,
# no tracing domains) over warm-up + 2 steps = 3 x 10^7 reads -> gpurun_out/screen_pmc_latest.json (copy to profiles/)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BS="python $ROOT/bench.py --no-cpu --no-h2h --no-sketch --no-c5 --no-cli --no-brackets --steps 1 --warmup 0 --detail /tmp/screen_detail.json"
rm -rf $OUT/r04_s_fetch $OUT/r04_s_write
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/r04_s_fetch -o p -- $BS > $OUT/r04_s_fetch.log 2>&1; echo "fetch rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/r04_s_write -o p -- $BS > $OUT/r04_s_write.log 2>&1; echo "write rc=$?"
cd $ROOT
python tools/make_pmc_json.py gpurun_out/r04_s_ "256, true>" 30000000 read gpurun_out/screen_pmc_latest.json mash_amd/csrc/sketch.hip mash_amd/csrc/kmer_hash.h mash_amd/csrc/screen.hip | cut -c1-600
# (the counter files themselves are large: keep the summary only)
rm -rf $OUT/r04_s_fetch $OUT/r04_s_write
