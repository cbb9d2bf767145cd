#!/bin/bash
# round-3 GPU run N: merge with several rows per item, A/B + parity
mkdir -p gpurun_out
( timeout 900 python tools/merge_pack_ab.py ) > gpurun_out/n_pack_ab.json 2> gpurun_out/n_pack_ab.err; python -c "
import json
d=json.load(open('gpurun_out/n_pack_ab.json'))
for k,v in d.items():
    print(k, 'pack0', [(x['ms'],x['compare_merge']) for x in v['pack0']], 'pack1', [(x['ms'],x['compare_merge']) for x in v['pack1']])
"; tail -3 gpurun_out/n_pack_ab.err
( MASHGPU_SPARSE_MERGE_PACK=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "compare or sparse or triangle or rect or identical or sharded_compare or finish or survivor" ; echo "rc=$?" ) > gpurun_out/n_tests.log 2>&1; tail -4 gpurun_out/n_tests.log
( MASHGPU_SPARSE_MERGE_PACK=1 timeout 300 python tools/compare_fuzz.py --n 100000 --seconds 60 --seed 91 ) > gpurun_out/n_compare_fuzz.txt 2>&1; tail -2 gpurun_out/n_compare_fuzz.txt
