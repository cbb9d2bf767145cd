ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; TAG=r02
cd /tmp && export TMPDIR=/tmp
BS="python $ROOT/bench.py --no-cpu --no-h2h --no-sketch --no-c5 --steps 1 --warmup 0"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${TAG}_s_fetch" -o p -- $BS > $OUT/${TAG}_sfetch.log 2>&1; echo rc=$?
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${TAG}_s_write" -o p -- $BS > $OUT/${TAG}_swrite.log 2>&1; echo rc=$?
cd $ROOT
python tools/make_pmc_json.py gpurun_out/${TAG}_s_ "256, true>" 30000000 read gpurun_out/${TAG}_screen_pmc.json mash_amd/csrc/sketch.hip mash_amd/csrc/kmer_hash.h mash_amd/csrc/screen.hip
