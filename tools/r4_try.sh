#!/bin/bash
# tools/r4_try.sh -- a quick GPU check of a change to the compare engine: its tests, a short fuzz, the cold legs with a kernel trace
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "compare or invalidate or sparse or dense" ; echo "rc=$?" ) > gpurun_out/try_tests.log 2>&1
tail -8 gpurun_out/try_tests.log
( timeout 120 python tools/compare_fuzz.py --n 100000 --seconds 50 --seed 977 ) > gpurun_out/try_fuzz.txt 2>&1; tail -2 gpurun_out/try_fuzz.txt
( timeout 300 python tools/prof_leg.py --leg c3 --cold --steps 5 ) > gpurun_out/try_c3.log 2>&1; tail -1 gpurun_out/try_c3.log
( timeout 300 python tools/prof_leg.py --leg c5 --cold --steps 3 ) > gpurun_out/try_c5.log 2>&1; tail -1 gpurun_out/try_c5.log
( MASHGPU_SPARSE_SORT_ALL_BITS=1 timeout 300 python tools/prof_leg.py --leg c3 --cold --steps 5 ) > gpurun_out/try_c3_all.log 2>&1; tail -1 gpurun_out/try_c3_all.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_leg.py --leg c3 --cold --steps 5 > /dev/null 2>&1
f=$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/gpurun_out/try_c3_kernel_stats.csv && head -25 $f | cut -c1-150
