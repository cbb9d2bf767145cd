#!/bin/bash
# the big-bucket path: its tests, then the clade brackets with the index built by tiles and by the sort
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${TESTS:-index_built_by_tiles or dense_groups or clade}" > gpurun_out/r5_big_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r5_big_tests.log
for leg in one_clade clades; do
    for mode in tiles sort; do
        for cold in "--cold" ""; do
            MASHGPU_SPARSE_DBG=1 MASHGPU_SPARSE_INDEX=$mode timeout 300 python tools/prof_leg.py --leg $leg --steps 4 $cold > gpurun_out/r5_big_${leg}_${mode}${cold}.log 2>&1
            echo "$leg $mode $cold rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5_big_${leg}_${mode}${cold}.log | head -1) $(grep -m1 'index by tiles' gpurun_out/r5_big_${leg}_${mode}${cold}.log)"
        done
    done
done
LEGS="one_clade_cold clades_cold" TOP=14 MASHGPU_SPARSE_INDEX=tiles bash tools/r5_quick.sh 2>&1 | grep -v "^tests"
