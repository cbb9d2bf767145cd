#!/bin/bash
# tools/next_round_first_run.sh -- the first GPU call of the next round (about 12 GPU-minutes): validate and measure the three
# opt-in additions of compare_sparse_x.hip, so that they can become defaults (and be profiled) right away.
#   1. every compare test + the new-run test with each addition forced, each under its own timeout (a hang costs 5 minutes,
#      not the round: the first version of the run dedupe cost 40);
#   2. the compare fuzzer with all three on;
#   3. A/B on the bench tables: default / pack / pack + one-class / all three (C3, clades, identical, s = 400), phases from the
#      library's HIP events, checksums compared;
# Output: gpurun_out/nr_*.  Then: make them the default in mashgpu.cpp, move the kernels into compare_sparse.hip,
# tools/profile_round3.sh for fresh kernel stats + PMC, full bench.
mkdir -p gpurun_out
K="compare or sparse or triangle or rect or identical or copies or runs or finish or survivor or sharded_compare"
for cfg in "MASHGPU_SPARSE_MERGE_PACK=1" "MASHGPU_SPARSE_ONE_CLASS=1" "MASHGPU_SPARSE_RUN_DEDUP=1 MASHGPU_TEST_RUN_DEDUP=1" \
           "MASHGPU_SPARSE_MERGE_PACK=1 MASHGPU_SPARSE_ONE_CLASS=1 MASHGPU_SPARSE_RUN_DEDUP=1"; do
    tag=$(echo "$cfg" | tr -c 'A-Z0-9_=\n' '_' | cut -c1-60)
    ( env $cfg timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "$K" ; echo "rc=$?" ) > gpurun_out/nr_tests_$tag.log 2>&1
    echo "$cfg: $(tail -2 gpurun_out/nr_tests_$tag.log | tr '\n' ' ')"
done
( MASHGPU_SPARSE_MERGE_PACK=1 MASHGPU_SPARSE_ONE_CLASS=1 MASHGPU_SPARSE_RUN_DEDUP=1 timeout 150 python tools/compare_fuzz.py --n 100000 --seconds 120 --seed 401 ) \
    > gpurun_out/nr_compare_fuzz.txt 2>&1; tail -2 gpurun_out/nr_compare_fuzz.txt
( timeout 600 python tools/next_round_ab.py ) > gpurun_out/nr_ab.json 2> gpurun_out/nr_ab.err; cat gpurun_out/nr_ab.json | cut -c1-3000; tail -2 gpurun_out/nr_ab.err
