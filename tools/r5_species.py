#!/usr/bin/env python3
"""the one_species bracket through each engine (MASHGPU_COMPARE_KERNEL), per table and per further pass"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mash_amd import abi
from workloads import synth_torch
torch.cuda.init()
dev = torch.device("cuda", 0)
eng = abi.MashGpu(0, stream=torch.cuda.current_stream().cuda_stream)
n, S = int(os.environ.get("N", 32768)), 1000
h, nh, ln = synth_torch.species_sketch_table(n, S, device=dev)
torch.cuda.synchronize()
t = eng.table_wrap(h.data_ptr(), nh.data_ptr(), ln.data_ptr(), n, S, keep=(h, nh, ln))
pairs = n * (n - 1) // 2
out = torch.empty((pairs, 2), dtype=torch.int32, device=dev)
ref = None
for k in sys.argv[1:] or ["default", "sparse", "merged"]:
    if k != "default":
        os.environ["MASHGPU_COMPARE_KERNEL"] = k
    else:
        os.environ.pop("MASHGPU_COMPARE_KERNEL", None)
    t.invalidate()
    eng.compare_tri_dev(t, 0, n, out.data_ptr())
    torch.cuda.synchronize()
    t.invalidate()
    t0 = time.perf_counter()
    eng.compare_tri_dev(t, 0, n, out.data_ptr())
    torch.cuda.synchronize()
    cold = time.perf_counter() - t0
    t0 = time.perf_counter()
    eng.compare_tri_dev(t, 0, n, out.data_ptr())
    torch.cuda.synchronize()
    warm = time.perf_counter() - t0
    sums = [int(out[:, 0].sum(dtype=torch.int64).item()), int(out[:, 1].sum(dtype=torch.int64).item())]
    ref = ref or sums
    print(json.dumps({"engine": k, "pairs_s_per_table": pairs / cold, "pairs_s_warm": pairs / warm, "ms": [cold * 1e3, warm * 1e3], "sums": sums, "same": sums == ref}))
