#!/bin/bash
mkdir -p gpurun_out
bash tools/r4_prof.sh c3 cold
bash tools/r4_prof.sh random cold | head -30
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "compare or sparse or triangle or rect or invalidate or reads_session or identical or copies or finish or sharded" ; echo "rc=$?" ) > gpurun_out/r4_tests_subset.log 2>&1
tail -4 gpurun_out/r4_tests_subset.log
