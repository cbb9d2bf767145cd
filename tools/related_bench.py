#!/usr/bin/env python3
"""Throughput of the compare engines on RELATED sketches (the expensive case: every shared hash is
ranked exactly).  Cases: all sketches identical; clades of near-identical sketches laid out as runs of
consecutive rows (taxonomically sorted collections) and interleaved; the unrelated C3-style table for
scale.  usage: python tools/related_bench.py [--n 20000] > profiles/r02_related.json"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mash_amd import abi
from workloads import synth_torch

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20000)
ap.add_argument("--engines", default="default,sparse,merged,plain")
args = ap.parse_args()
torch.cuda.init()
dev = torch.device("cuda", 0)
eng = abi.MashGpu(0, stream=torch.cuda.current_stream().cuda_stream)
S = 1000
n = args.n

def tables():
    h, nh, ln = synth_torch.clustered_sketch_table(n, S, clusters=max(1, n // 100), device=dev)
    yield "unrelated_c3_style", h, nh, ln
    h1 = h[:1].repeat(n, 1).contiguous()
    yield "all_identical", h1, torch.full((n,), S, dtype=torch.int32, device=dev), ln
    for contiguous in (True, False):
        hc, nc, lc = synth_torch.clustered_sketch_table(n, S, clusters=max(1, n // 1000), pool=1030, private=20, keep_p=0.97,
                                                        device=dev, contiguous=contiguous)
        yield ("clades_of_1000_contiguous" if contiguous else "clades_of_1000_interleaved"), hc, nc, lc

res = {"n": n, "s": S, "cases": {}}
pairs = n * (n - 1) // 2
out = torch.empty((pairs, 2), dtype=torch.int32, device=dev)
for name, h, nh, ln in tables():
    torch.cuda.synchronize()
    t = eng.table_wrap(h.data_ptr(), nh.data_ptr(), ln.data_ptr(), n, S, keep=(h, nh, ln))
    row = {}
    sums = {}
    for e in args.engines.split(","):
        for k in ("MASHGPU_COMPARE_KERNEL", "MASHGPU_COMPARE_WINDOWS"):
            os.environ.pop(k, None)
        if e == "plain":
            os.environ["MASHGPU_COMPARE_WINDOWS"] = "0"
        elif e == "windows":
            os.environ["MASHGPU_COMPARE_WINDOWS"] = "1"
        elif e != "default":
            os.environ["MASHGPU_COMPARE_KERNEL"] = e
        m = n
        mp = m * (m - 1) // 2
        eng.compare_tri_dev(t, 0, m, out.data_ptr())
        torch.cuda.synchronize()
        eng.prof_enable(True)
        eng.prof_reset()
        t0 = time.perf_counter()
        eng.compare_tri_dev(t, 0, m, out.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        row[e] = {"pairs_per_s": mp / dt, "ms": dt * 1e3, "rows": m}
        for phase in ("compare", "compare_fill", "compare_discover", "compare_merge"):
            ms, k = eng.prof_avg_ms(phase)
            if k:
                row[e][phase + "_ms"] = round(ms * k, 3)
        eng.prof_enable(False)
        sums[e] = (int(out[:mp, 0].sum(dtype=torch.int64).item()), int(out[:mp, 1].sum(dtype=torch.int64).item())) if m == n else None
    full = [v for v in sums.values() if v is not None]
    row["engines_agree"] = all(v == full[0] for v in full)
    row["mean_shared"] = full[0][0] / pairs if full else None
    res["cases"][name] = row
    t.free()
print(json.dumps(res, indent=1))
