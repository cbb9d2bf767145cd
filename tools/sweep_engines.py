#!/usr/bin/env python3
"""Window engine vs plain tiles over the number of sketches (unrelated C3-style tables, s = 1000) and
over sketch sizes: where each wins decides the default dispatch in run_compare_merged."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mash_amd import abi
from workloads import synth_torch
torch.cuda.init()
dev = torch.device("cuda", 0)
eng = abi.MashGpu(0, stream=torch.cuda.current_stream().cuda_stream)
res = []
import itertools
CASES = ((1000, (4000, 10000, 20000, 40000, 70000)), (400, (20000, 60000)), (2000, (10000, 30000)), (5000, (10000,)))
if len(sys.argv) > 1 and sys.argv[1] == "tiles":
    CASES = ((1000, (4000, 10000, 20000, 40000)), (2000, (10000,)))
for s, ns in CASES:
    for n in ns:
        h, nh, ln = synth_torch.clustered_sketch_table(n, s, clusters=max(1, n // 100), pool=int(1.5 * s), private=int(0.4 * s), device=dev, block=max(500, 20000000 // (2 * s)))
        torch.cuda.synchronize()
        t = eng.table_wrap(h.data_ptr(), nh.data_ptr(), ln.data_ptr(), n, s, keep=(h, nh, ln))
        pairs = n * (n - 1) // 2
        out = torch.empty((pairs, 2), dtype=torch.int32, device=dev)
        row = {"s": s, "n": n}
        variants = ("windows", "plain")
        if len(sys.argv) > 1 and sys.argv[1] == "tiles":
            variants = ("windows", "windows@2048", "windows@6000", "plain", "plain@2048", "plain@6000")
        for e in variants:
            os.environ["MASHGPU_COMPARE_WINDOWS"] = "1" if e.startswith("windows") else "0"
            os.environ.pop("MASHGPU_COMPARE_MIN_TILES", None)
            if "@" in e:
                os.environ["MASHGPU_COMPARE_MIN_TILES"] = e.split("@")[1]
            eng.compare_tri_dev(t, 0, n, out.data_ptr())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                eng.compare_tri_dev(t, 0, n, out.data_ptr())
            torch.cuda.synchronize()
            row[e] = pairs * 2 / (time.perf_counter() - t0)
        res.append(row)
        print(json.dumps(row), flush=True)
        t.free()
        del out, h
