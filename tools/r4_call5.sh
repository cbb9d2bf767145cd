#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q ; echo "rc=$?" ) > gpurun_out/r4_tests.log 2>&1
tail -6 gpurun_out/r4_tests.log
( timeout 200 python tools/compare_fuzz.py --n 100000 --seconds 100 --seed 411 ) > gpurun_out/r4_compare_fuzz.txt 2>&1; tail -3 gpurun_out/r4_compare_fuzz.txt
