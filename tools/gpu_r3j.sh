#!/bin/bash
# round-3 GPU run J: ingest rework (early parse start, grouped staging windows, progress log, fast exit): CLI tests + e2e timing
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_cli.py tests/test_refcli_gpu.py -m gpu -q -x ; echo "rc=$?" ) > gpurun_out/j_cli_tests.log 2>&1; tail -3 gpurun_out/j_cli_tests.log
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "streamed or sharded_sketch" ; echo "rc=$?" ) > gpurun_out/j_stream_tests.log 2>&1; tail -3 gpurun_out/j_stream_tests.log
( timeout 300 python tests/fuzz_cli.py --n 100000 --seconds 60 --seed 101 ) > gpurun_out/j_cli_fuzz.txt 2>&1; tail -2 gpurun_out/j_cli_fuzz.txt
nproc
( timeout 600 python tools/sketch_e2e.py --variants --ref-threads 64 ) > gpurun_out/j_sketch_e2e.json 2> gpurun_out/j_sketch_e2e.err; cat gpurun_out/j_sketch_e2e.json; tail -3 gpurun_out/j_sketch_e2e.err
( timeout 600 python tools/sketch_e2e.py --genomes 300 --len 4000000 --reps 2 --ref-threads 64 ) > gpurun_out/j_sketch_e2e_bact.json 2> gpurun_out/j_sketch_e2e_bact.err; cat gpurun_out/j_sketch_e2e_bact.json; tail -3 gpurun_out/j_sketch_e2e_bact.err
