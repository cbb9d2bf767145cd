#!/bin/bash
# the per-table job with the index built by tiles and by the round-4 sort, same box, same process order
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
: > gpurun_out/r05_tiles_vs_sort.txt
for leg in c3 c5; do
    for rep in 1 2; do
        for mode in tiles sort; do
            MASHGPU_SPARSE_INDEX=$mode timeout 300 python tools/prof_leg.py --leg $leg --steps 6 --cold > gpurun_out/r5_tvs.log 2>&1
            echo "$leg index=$mode $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5_tvs.log | head -1)" | tee -a gpurun_out/r05_tiles_vs_sort.txt
        done
    done
done
