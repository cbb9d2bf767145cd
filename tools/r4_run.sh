#!/bin/bash
# tools/r4_run.sh [tests|bench|both] -- one GPU call of round 4: the -m gpu suite and/or the full bench, each under its own timeout,
# logs under gpurun_out/
what=${1:-both}
mkdir -p gpurun_out
if [ "$what" = tests ] || [ "$what" = both ]; then
    ( timeout 900 python -m pytest tests -m gpu -x -q ; echo "rc=$?" ) > gpurun_out/r4_tests.log 2>&1
    tail -15 gpurun_out/r4_tests.log
fi
if [ "$what" = fuzz ] || [ "$what" = both ]; then
    ( timeout 200 python tools/compare_fuzz.py --n 100000 --seconds 100 --seed 411 ) > gpurun_out/r4_compare_fuzz.txt 2>&1; tail -2 gpurun_out/r4_compare_fuzz.txt
fi
if [ "$what" = bench ] || [ "$what" = both ]; then
    ( timeout 900 python bench.py --steps 5 --warmup 2 ; echo "rc=$?" ) > gpurun_out/r4_bench.log 2> gpurun_out/r4_bench.err
    tail -c 6000 gpurun_out/r4_bench.log; tail -5 gpurun_out/r4_bench.err
fi
