#!/bin/bash
# round-3 GPU run R (the last 40 s): compare fuzzer with the run dedupe on; then whatever fits of the compare tests in default mode
mkdir -p gpurun_out
( MASHGPU_SPARSE_RUN_DEDUP=1 timeout 16 python tools/compare_fuzz.py --n 100000 --seconds 10 --seed 95 ) > gpurun_out/r_dedup_fuzz.txt 2>&1; tail -2 gpurun_out/r_dedup_fuzz.txt
( timeout 20 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "compare" ; echo "rc=$?" ) > gpurun_out/r_tests.log 2>&1; tail -3 gpurun_out/r_tests.log
