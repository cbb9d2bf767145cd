#!/bin/bash
mkdir -p gpurun_out
( timeout 900 python bench.py --steps 5 --warmup 2 ; echo "rc=$?" ) > gpurun_out/r4_bench.log 2> gpurun_out/r4_bench.err
tail -c 3500 gpurun_out/r4_bench.log; tail -3 gpurun_out/r4_bench.err
timeout 1500 bash tools/profile_round4.sh 2>&1 | tail -40
