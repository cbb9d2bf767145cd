#!/bin/bash
# the dense pairs kernel: its tests, then the brackets that live on it (cold and warm) and c3
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${TESTS:-dense or clade or species or cluster or sparse_matrix or thresholded}" > gpurun_out/r5_dense_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r5_dense_tests.log
for leg in one_clade clades c3; do
    for cold in "--cold" ""; do
        timeout 300 python tools/prof_leg.py --leg $leg --steps 5 $cold > gpurun_out/r5_dense_${leg}${cold}.log 2>&1
        echo "$leg $cold rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5_dense_${leg}${cold}.log | head -1)"
    done
done
LEGS="one_clade_cold c3_cold" TOP=8 bash tools/r5_quick.sh 2>&1 | grep -v "^tests"
