#!/bin/bash
# round-3 GPU run O: pack variants A/B; regression test for tables of copies; compare fuzz with the pack kernel
mkdir -p gpurun_out
( timeout 900 python tools/merge_pack_ab.py ) > gpurun_out/o_pack_ab.json 2> gpurun_out/o_pack_ab.err; python -c "
import json
d=json.load(open('gpurun_out/o_pack_ab.json'))
for k,v in d.items():
    print(k, {c: [(x['ms'],x['compare_merge']) for x in r] for c,r in v.items()})
"; tail -3 gpurun_out/o_pack_ab.err
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "copies or extremes" ; echo "rc=$?" ) > gpurun_out/o_tests.log 2>&1; tail -3 gpurun_out/o_tests.log
( MASHGPU_SPARSE_MERGE_PACK=1 timeout 300 python tools/compare_fuzz.py --n 100000 --seconds 50 --seed 92 ) > gpurun_out/o_compare_fuzz.txt 2>&1; tail -2 gpurun_out/o_compare_fuzz.txt
( MASHGPU_SPARSE_MERGE_PACK=1 MASHGPU_SPARSE_PACK_MIN=43 timeout 300 python tools/compare_fuzz.py --n 100000 --seconds 30 --seed 93 ) > gpurun_out/o_compare_fuzz43.txt 2>&1; tail -1 gpurun_out/o_compare_fuzz43.txt
