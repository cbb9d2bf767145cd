#!/usr/bin/env python3
"""a row range of the C5 (or C3) triangle, per table and per further pass: python tools/range_check.py [c5|c3] [rb] [re]
(C5 rows 50 000 - 100 000 warm within twice their share of the pairs of the whole triangle's pass)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mash_amd import abi, shard
from workloads import synth_torch
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
n = 100000
rb = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
re = int(sys.argv[3]) if len(sys.argv) > 3 else n
torch.cuda.init()
dev = torch.device("cuda", 0)
eng = abi.MashGpu(0, stream=torch.cuda.current_stream().cuda_stream)
if which == "c5":
    S = 10000
    h, nh, ln = synth_torch.clustered_sketch_table(n, S, clusters=n // 100, pool=15000, private=4000, device=dev, block=2000)
else:
    S = 1000
    h, nh, ln = synth_torch.clustered_sketch_table(n, S, clusters=n // 100, device=dev)
torch.cuda.synchronize()
t = eng.table_wrap(h.data_ptr(), nh.data_ptr(), ln.data_ptr(), n, S, keep=(h, nh, ln))
out = torch.empty((n * (n - 1) // 2, 2), dtype=torch.int32, device=dev)
res = {}
for name, a, b in (("whole", 0, n), ("range", rb, re)):
    t.invalidate()
    eng.prof_enable(True)
    eng.prof_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.compare_tri_dev(t, a, b, out.data_ptr())
    torch.cuda.synchronize()
    cold = (time.perf_counter() - t0) * 1e3
    ph = {p: round(eng.prof_avg_ms("compare_" + p)[0] * eng.prof_avg_ms("compare_" + p)[1], 3) for p in ("index", "discover", "fill", "fill_aside", "dense", "merge", "join")}
    eng.prof_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.compare_tri_dev(t, a, b, out.data_ptr())
    torch.cuda.synchronize()
    warm = (time.perf_counter() - t0) * 1e3 / 3
    phw = {p: round(eng.prof_avg_ms("compare_" + p)[0], 3) for p in ("discover", "fill", "dense", "merge", "join")}
    eng.prof_enable(False)
    pairs = shard.tri_pairs(a, b)
    res[name] = {"rows": [a, b], "pairs": pairs, "per_table_ms": round(cold, 3), "per_pass_ms": round(warm, 3), "phases_per_table": ph, "phases_per_pass": phw,
                 "sums": [int(out[:pairs, 0].sum(dtype=torch.int64).item()), int(out[:pairs, 1].sum(dtype=torch.int64).item())]}
    print(json.dumps({name: res[name]}), flush=True)
share = res["range"]["pairs"] / res["whole"]["pairs"]
print(json.dumps({"workload": which, "share_of_pairs": round(share, 4), "range_pass_over_its_share_of_the_whole_pass": round(res["range"]["per_pass_ms"] / (share * res["whole"]["per_pass_ms"]), 3)}))
