#!/bin/bash
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dense or invalidate" ; echo "rc=$?" ) > gpurun_out/r4_tests_subset.log 2>&1
tail -2 gpurun_out/r4_tests_subset.log
MASHGPU_DENSE_STREAM=1 bash tools/r4_prof.sh c3 | grep -E "rc=|dn_pairs"
bash tools/r4_prof.sh c3 | grep -E "rc=|dn_pairs"
bash tools/r4_prof.sh one_clade | grep -E "rc=|dn_pairs"
bash tools/r4_prof.sh c5 | grep -E "rc=|dn_pairs"
