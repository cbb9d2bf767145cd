#!/bin/bash
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense or invalidate or sparse" ; echo "rc=$?" ) > gpurun_out/r4_tests_subset.log 2>&1
tail -3 gpurun_out/r4_tests_subset.log
bash tools/r4_prof.sh c5 cold | head -12
bash tools/r4_prof.sh c3 cold | head -12
