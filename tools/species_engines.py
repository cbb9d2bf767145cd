#!/usr/bin/env python3
"""the one_species bracket through each engine (MASHGPU_COMPARE_KERNEL), per table and per further pass, with the library's
own phase times (mg_prof_*); args: engines (default join sparse merged); env N, S, EXTRA (name=value,... set for every run)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mash_amd import abi
from workloads import synth_torch
torch.cuda.init()
dev = torch.device("cuda", 0)
eng = abi.MashGpu(0, stream=torch.cuda.current_stream().cuda_stream)
n, S = int(os.environ.get("N", 32768)), int(os.environ.get("S", 1000))
h, nh, ln = synth_torch.species_sketch_table(n, S, device=dev)
torch.cuda.synchronize()
t = eng.table_wrap(h.data_ptr(), nh.data_ptr(), ln.data_ptr(), n, S, keep=(h, nh, ln))
pairs = n * (n - 1) // 2
out = torch.empty((pairs, 2), dtype=torch.int32, device=dev)
ref = None
for k in sys.argv[1:] or ["default", "join", "sparse", "merged"]:
    name, _, extra = k.partition("+")
    if name != "default":
        os.environ["MASHGPU_COMPARE_KERNEL"] = name
    else:
        os.environ.pop("MASHGPU_COMPARE_KERNEL", None)
    sets = [kv.split("=") for kv in extra.split(",") if kv]
    for a, b in sets:
        os.environ[a] = b
    t.invalidate()
    eng.compare_tri_dev(t, 0, n, out.data_ptr())
    torch.cuda.synchronize()
    eng.prof_enable(True)
    eng.prof_reset()
    t.invalidate()
    t0 = time.perf_counter()
    eng.compare_tri_dev(t, 0, n, out.data_ptr())
    torch.cuda.synchronize()
    cold = time.perf_counter() - t0
    ph = {p: [round(x, 3) for x in eng.prof_avg_ms("compare_" + p)] for p in ("index", "discover", "fill", "dense", "merge", "join")}
    eng.prof_reset()
    t0 = time.perf_counter()
    eng.compare_tri_dev(t, 0, n, out.data_ptr())
    torch.cuda.synchronize()
    warm = time.perf_counter() - t0
    eng.prof_enable(False)
    sums = [int(out[:, 0].sum(dtype=torch.int64).item()), int(out[:, 1].sum(dtype=torch.int64).item())]
    ref = ref or sums
    print(json.dumps({"engine": k, "pairs_s_per_table": pairs / cold, "pairs_s_warm": pairs / warm, "ms": [round(cold * 1e3, 3), round(warm * 1e3, 3)],
                      "phases_cold": ph, "sums": sums, "same": sums == ref}), flush=True)
    for a, b in sets:
        os.environ.pop(a, None)
