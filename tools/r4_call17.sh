#!/bin/bash
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense or invalidate or sparse or triangle" ; echo "rc=$?" ) > gpurun_out/r4_tests_subset.log 2>&1
tail -2 gpurun_out/r4_tests_subset.log
for leg in c3 one_clade c5 clades; do bash tools/r4_prof.sh $leg | grep -E "rc=|dn_pairs"; done
( timeout 100 python tools/compare_fuzz.py --n 100000 --seconds 40 --seed 77 ) > gpurun_out/r4_compare_fuzz.txt 2>&1; tail -1 gpurun_out/r4_compare_fuzz.txt
