#!/bin/bash
# tools/r4_final.sh -- the round's closing GPU call: full -m gpu suite, compare fuzzer, full bench, profile passes (kernel stats + PMC)
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q ; echo "rc=$?" ) > gpurun_out/r4_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r4_tests.log | tail -2
( timeout 200 python tools/compare_fuzz.py --n 100000 --seconds 100 --seed 2026 ) > gpurun_out/r4_compare_fuzz.txt 2>&1; tail -1 gpurun_out/r4_compare_fuzz.txt
timeout 1500 bash tools/profile_round4.sh 2>&1 | tail -30
cp gpurun_out/compare_*_pmc.json profiles/ 2>/dev/null      # (the bench reads the counters of THESE sources from profiles/)
( timeout 900 python bench.py --steps 20 --warmup 5 ; echo "rc=$?" ) > gpurun_out/r4_bench.log 2> gpurun_out/r4_bench.err
tail -c 3300 gpurun_out/r4_bench.log; tail -3 gpurun_out/r4_bench.err
