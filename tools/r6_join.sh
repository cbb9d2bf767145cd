#!/bin/bash
# round 6, first contact of the join engine: its GPU tests, then the one_species bracket through every engine
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${TESTS:-join or one_species}" > gpurun_out/r6_join_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r6_join_tests.log
MASHGPU_SPARSE_DBG=1 timeout 900 python tools/r6_species.py ${ENGINES:-default join join+MASHGPU_JOIN_NO_EARLY_STOP=1 sparse} > gpurun_out/r6_species.txt 2> gpurun_out/r6_species.err
echo "species rc=$?"; cat gpurun_out/r6_species.txt; grep -v "^$" gpurun_out/r6_species.err | tail -20
