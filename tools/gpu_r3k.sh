#!/bin/bash
# round-3 GPU run K: ingest after the wake-up fix, small batches, parallel hand-out
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_cli.py -m gpu -q -x ; echo "rc=$?" ) > gpurun_out/k_cli_tests.log 2>&1; tail -3 gpurun_out/k_cli_tests.log
( timeout 300 python tests/fuzz_cli.py --n 100000 --seconds 40 --seed 202 ) > gpurun_out/k_cli_fuzz.txt 2>&1; tail -2 gpurun_out/k_cli_fuzz.txt
( timeout 900 python tools/sketch_e2e.py --variants --ref-threads 64 ) > gpurun_out/k_sketch_e2e.json 2> gpurun_out/k_sketch_e2e.err; cat gpurun_out/k_sketch_e2e.json; tail -3 gpurun_out/k_sketch_e2e.err
( timeout 600 python tools/sketch_e2e.py --genomes 300 --len 4000000 --reps 2 ) > gpurun_out/k_sketch_e2e_bact.json 2> gpurun_out/k_sketch_e2e_bact.err; cat gpurun_out/k_sketch_e2e_bact.json; tail -3 gpurun_out/k_sketch_e2e_bact.err
