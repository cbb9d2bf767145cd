#!/bin/bash
# round-3 GPU run Q (70 s of budget left): the final default tree on a handful of compare tests; the rewritten run dedupe under a 20 s guard
mkdir -p gpurun_out
( timeout 45 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "table_of_copies or runs_that_name or extremes or compare_matches_oracle or rect_matches" ; echo "rc=$?" ) > gpurun_out/q_tests.log 2>&1; tail -3 gpurun_out/q_tests.log
( MASHGPU_TEST_RUN_DEDUP=1 timeout 20 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "runs_that_name" ; echo "rc=$?" ) > gpurun_out/q_dedup.log 2>&1; tail -3 gpurun_out/q_dedup.log
