#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "compare or sparse or triangle or rect or invalidate or identical or copies or finish or sharded or dense" ; echo "rc=$?" ) > gpurun_out/r4_tests_subset.log 2>&1
tail -8 gpurun_out/r4_tests_subset.log
( timeout 200 python tools/compare_fuzz.py --n 100000 --seconds 60 --seed 512 ) > gpurun_out/r4_compare_fuzz.txt 2>&1; tail -3 gpurun_out/r4_compare_fuzz.txt
( MASHGPU_SPARSE_DBG=1 timeout 600 python bench.py --steps 3 --warmup 2 --no-sketch --no-screen --no-cli --no-h2h --no-cpu ; echo "rc=$?" ) > gpurun_out/r4_bench.log 2> gpurun_out/r4_bench.err
tail -c 2500 gpurun_out/r4_bench.log; grep -E "compare dense|index of" gpurun_out/r4_bench.err | sort | uniq -c | head -20
