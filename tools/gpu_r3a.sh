#!/bin/bash
# round-3 GPU run A: first contact of the sparse engine with hardware
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sparse" ; echo "rc=$?" ) > gpurun_out/a_pytest_sparse.log 2>&1
tail -5 gpurun_out/a_pytest_sparse.log
( MASHGPU_SPARSE_DBG=1 timeout 600 python tools/sparse_probe.py --n 100000 ) > gpurun_out/a_probe_c3.json 2> gpurun_out/a_probe_c3.err
tail -40 gpurun_out/a_probe_c3.json; tail -5 gpurun_out/a_probe_c3.err
( MASHGPU_SPARSE_DBG=1 timeout 600 python tools/sparse_probe.py --n 20000 --steps 10 ) > gpurun_out/a_probe_20k.json 2> gpurun_out/a_probe_20k.err
( timeout 900 python tools/related_bench.py --n 20000 --engines default,sparse,merged ) > gpurun_out/a_related.json 2> gpurun_out/a_related.err
tail -60 gpurun_out/a_related.json; tail -3 gpurun_out/a_related.err
( timeout 900 python -m pytest tests -m gpu -q -x ; echo "rc=$?" ) > gpurun_out/a_pytest_all.log 2>&1
tail -5 gpurun_out/a_pytest_all.log
