#!/usr/bin/env python3
"""A/B of the merge's work distribution on the bench tables: one row per item (MASHGPU_SPARSE_MERGE_PACK=0) against
several rows per item (=1), alternating in one process; phases from the library's HIP events, checksums compared."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mash_amd import abi
from workloads import synth_torch

dev = torch.device("cuda", 0)
eng = abi.MashGpu(0, stream=torch.cuda.current_stream().cuda_stream)
res = {}
for name, make in (("c3", lambda: synth_torch.clustered_sketch_table(100000, 1000, clusters=1000, device=dev)),
                   ("c3_n20000", lambda: synth_torch.clustered_sketch_table(20000, 1000, clusters=200, device=dev)),
                   ("s400", lambda: synth_torch.clustered_sketch_table(100000, 400, clusters=1000, pool=600, private=160, device=dev))):
    h, nh, ln = make()
    n, s = h.shape
    torch.cuda.synchronize()
    t = eng.table_wrap(h.data_ptr(), nh.data_ptr(), ln.data_ptr(), n, s, keep=(h, nh, ln))
    pairs = n * (n - 1) // 2
    out = torch.empty((pairs, 2), dtype=torch.int32, device=dev)
    eng.compare_tri_dev(t, 0, n, out.data_ptr())
    r = {}
    for rep in range(2):
        for pack, pmin in (("0", "32"), ("1", "32"), ("1", "43"), ("1", "64")):
            os.environ["MASHGPU_SPARSE_MERGE_PACK"] = pack
            os.environ["MASHGPU_SPARSE_PACK_MIN"] = pmin
            out.zero_()
            torch.cuda.synchronize()
            eng.prof_enable(True); eng.prof_reset()
            t0 = time.perf_counter()
            for _ in range(3):
                eng.compare_tri_dev(t, 0, n, out.data_ptr())
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            ph = {k: round(eng.prof_avg_ms(k)[0], 3) for k in ("compare_fill", "compare_discover", "compare_merge")}
            eng.prof_enable(False)
            cs = [int(out[:, 0].sum(dtype=torch.int64)), int(out[:, 1].sum(dtype=torch.int64))]
            r.setdefault("pack" + pack + ("_min" + pmin if pack == "1" else ""), []).append({"ms": round(dt * 1e3, 3), **ph, "checksum": cs})
    assert len({tuple(x["checksum"]) for v in r.values() for x in v}) == 1, (name, r)
    del os.environ["MASHGPU_SPARSE_MERGE_PACK"], os.environ["MASHGPU_SPARSE_PACK_MIN"]
    res[name] = r
    t.free(); del out, h
    torch.cuda.empty_cache()
print(json.dumps(res))
