#!/usr/bin/env python3
"""A/B of the opt-in additions of compare_sparse_x.hip on the bench tables (tools/next_round_first_run.sh): per table the
default engine, then each combination; a fresh table object per configuration (the run dedupe acts at index build),
phases from the library's HIP events, first-call time (index build), checksums compared."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mash_amd import abi
from workloads import synth_torch

dev = torch.device("cuda", 0)
eng = abi.MashGpu(0, stream=torch.cuda.current_stream().cuda_stream)
CONFIGS = [("default", {}), ("pack", {"MASHGPU_SPARSE_MERGE_PACK": "1"}), ("one_class", {"MASHGPU_SPARSE_ONE_CLASS": "1"}),
           ("run_dedup", {"MASHGPU_SPARSE_RUN_DEDUP": "1"}),
           ("all", {"MASHGPU_SPARSE_MERGE_PACK": "1", "MASHGPU_SPARSE_ONE_CLASS": "1", "MASHGPU_SPARSE_RUN_DEDUP": "1"})]
TABLES = [("c3", lambda: synth_torch.clustered_sketch_table(100000, 1000, clusters=1000, device=dev)),
          ("clades", lambda: synth_torch.clade_sketch_table(100000, 1000, device=dev)),
          ("identical", lambda: synth_torch.identical_sketch_table(100000, 1000, device=dev)),
          ("s400", lambda: synth_torch.clustered_sketch_table(100000, 400, clusters=1000, pool=600, private=160, device=dev))]
res = {}
for name, make in TABLES:
    h, nh, ln = make()
    n, s = h.shape
    pairs = n * (n - 1) // 2
    out = torch.empty((pairs, 2), dtype=torch.int32, device=dev)
    r = {}
    for tag, env in CONFIGS:
        for k, v in env.items():
            os.environ[k] = v
        torch.cuda.synchronize()
        t = eng.table_wrap(h.data_ptr(), nh.data_ptr(), ln.data_ptr(), n, s, keep=(h, nh, ln))
        t0 = time.perf_counter()
        eng.compare_tri_dev(t, 0, n, out.data_ptr())
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        eng.prof_enable(True); eng.prof_reset()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.compare_tri_dev(t, 0, n, out.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        ph = {k: round(eng.prof_avg_ms(k)[0], 3) for k in ("compare_fill", "compare_discover", "compare_merge")}
        eng.prof_enable(False)
        r[tag] = {"ms": round(dt * 1e3, 3), "first_call_ms": round(first * 1e3, 1), **ph,
                  "checksum": [int(out[:, 0].sum(dtype=torch.int64)), int(out[:, 1].sum(dtype=torch.int64))]}
        t.free()
        for k in env:
            del os.environ[k]
    assert len({tuple(v["checksum"]) for v in r.values()}) == 1, (name, r)
    res[name] = r
    del out, h
    torch.cuda.empty_cache()
print(json.dumps(res))
