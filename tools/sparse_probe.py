#!/usr/bin/env python3
"""Phases of the inverted-index compare engine on a C3-style table: index build (first call),
counting pass, then per step fill / discover / merge, against the tile engine on the same table,
with the full-output checksum of both.  usage: python tools/sparse_probe.py [--n 100000] [--s 1000]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mash_amd import abi
from workloads import synth_torch

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100000)
ap.add_argument("--s", type=int, default=1000)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--no-dense", action="store_true")
ap.add_argument("--variants", default="", help="extra sparse runs: name:ENV=V,ENV=V;name2:...")
args = ap.parse_args()
torch.cuda.init()
dev = torch.device("cuda", 0)
eng = abi.MashGpu(0, stream=torch.cuda.current_stream().cuda_stream)
n, S = args.n, args.s
kw = dict(pool=15000, private=4000, block=2000) if S == 10000 else {}
h, nh, ln = synth_torch.clustered_sketch_table(n, S, clusters=max(1, n // 100), device=dev, **kw)
torch.cuda.synchronize()
t = eng.table_wrap(h.data_ptr(), nh.data_ptr(), ln.data_ptr(), n, S, keep=(h, nh, ln))
pairs = n * (n - 1) // 2
out = torch.empty((pairs, 2), dtype=torch.int32, device=dev)
res = {"n": n, "s": S, "pairs": pairs}


def sums():
    return [int(out[:, 0].sum(dtype=torch.int64).item()), int(out[:, 1].sum(dtype=torch.int64).item())]


runs = [("sparse", "sparse", {})] + ([] if args.no_dense else [("merged", "merged", {})])
for v in filter(None, args.variants.split(";")):
    name, _, envs = v.partition(":")
    runs.append((name, "sparse", dict(kv.split("=") for kv in envs.split(",") if kv)))
for name, engine, extra in runs:
    os.environ["MASHGPU_COMPARE_KERNEL"] = engine
    for k_, v_ in extra.items():
        os.environ[k_] = v_
    out.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.compare_tri_dev(t, 0, n, out.data_ptr())               # cold: index / prefix images are built here
    torch.cuda.synchronize()
    cold = time.perf_counter() - t0
    row = {"cold_ms": cold * 1e3, "checksum": sums()}
    eng.prof_enable(True)
    eng.prof_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.compare_tri_dev(t, 0, n, out.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    row["ms_per_step"] = dt * 1e3
    row["pairs_per_s"] = pairs / dt
    for phase in ("compare", "compare_fill", "compare_discover", "compare_merge", "compare_index"):
        ms, k = eng.prof_avg_ms(phase)
        if k:
            row[phase] = {"avg_ms": ms, "launches": k}
    eng.prof_enable(False)
    row["checksum_after_steps"] = sums()
    res[name] = row
    for k_ in extra:
        os.environ.pop(k_, None)
os.environ.pop("MASHGPU_COMPARE_KERNEL", None)
# what the default dispatch picks
eng.prof_enable(True)
eng.prof_reset()
eng.compare_tri_dev(t, 0, n, out.data_ptr())
res["default_engine"] = "sparse" if eng.prof_avg_ms("compare_fill")[1] else "tiles"
eng.prof_enable(False)
print(json.dumps(res, indent=1))
