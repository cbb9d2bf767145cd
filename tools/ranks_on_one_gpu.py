#!/usr/bin/env python3
"""What each of G ranks would pay for its block of the per-table job, run one after the other on ONE GPU (no 8-GPU box in
reach): for every rank the table is invalidated, its block [rb, re) of the equal-area cut is computed through mg_compare_tri_dev
-- the view of the rows below re, the clustered index with the block's rows in a segment of their own -- and the library's
phase times are read.  max over ranks = what the sharded job would take (without the table's broadcast).
   python tools/r6_ranks.py [c3|one_species] [G]      env PREFIX=0: views off (every rank indexes the whole table)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mash_amd import abi, shard
from workloads import synth_torch
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if os.environ.get("PREFIX") == "0":
    os.environ["MASHGPU_TRI_PREFIX"] = "0"
torch.cuda.init()
dev = torch.device("cuda", 0)
eng = abi.MashGpu(0, stream=torch.cuda.current_stream().cuda_stream)
S = 1000
if which == "c3":
    n = 100000
    h, nh, ln = synth_torch.clustered_sketch_table(n, S, clusters=n // 100, device=dev)
else:
    n = 32768
    h, nh, ln = synth_torch.species_sketch_table(n, S, device=dev)
torch.cuda.synchronize()
t = eng.table_wrap(h.data_ptr(), nh.data_ptr(), ln.data_ptr(), n, S, keep=(h, nh, ln))
blocks = shard.equal_area_row_blocks(n, G)
out = torch.empty((max(shard.tri_pairs(blocks[g], blocks[g + 1]) for g in range(G)), 2), dtype=torch.int32, device=dev)
eng.compare_tri_dev(t, 0, n, torch.empty((n * (n - 1) // 2, 2), dtype=torch.int32, device=dev).data_ptr())     # warm the pool
def run_blocks(blocks):
  res = []
  for g in range(G):
      rb, re = blocks[g], blocks[g + 1]
      best = None
      for rep in range(3):
          t.invalidate()
          eng.prof_enable(True)
          eng.prof_reset()
          torch.cuda.synchronize()
          t0 = time.perf_counter()
          eng.compare_tri_dev(t, rb, re, out.data_ptr())
          torch.cuda.synchronize()
          ms = (time.perf_counter() - t0) * 1e3
          ph = {p: round(eng.prof_avg_ms("compare_" + p)[0] * eng.prof_avg_ms("compare_" + p)[1], 3) for p in ("index", "discover", "fill", "fill_aside", "dense", "merge", "join")}
          eng.prof_enable(False)
          if best is None or ms < best[0]:
              best = (ms, ph)
      pairs = shard.tri_pairs(rb, re)
      sums = [int(out[:pairs, 0].sum(dtype=torch.int64).item()), int(out[:pairs, 1].sum(dtype=torch.int64).item())]
      res.append({"rank": g, "rows": [rb, re], "pairs": pairs, "ms": round(best[0], 3), "phases_ms": best[1], "sums": sums})
      print(json.dumps(res[-1]), flush=True)
  return res


res = run_blocks(blocks)
tot = n * (n - 1) // 2
print(json.dumps({"workload": which, "ranks": G, "cut": "equal areas", "prefix_views": os.environ.get("PREFIX") != "0", "max_ms": max(r["ms"] for r in res),
                  "pairs_s_if_sharded": tot / (max(r["ms"] for r in res) * 1e-3), "checksum": [sum(r["sums"][0] for r in res), sum(r["sums"][1] for r in res)]}))
# the cut bench.py makes from one measured step (mg_shard_tri_rows_costed): per pair what fill / join cost, per row discover + merge,
# per row of the view the index
fill = sum(r["phases_ms"]["fill"] + r["phases_ms"].get("fill_aside", 0.0) + r["phases_ms"]["join"] for r in res)
dm = sum(r["phases_ms"]["discover"] + r["phases_ms"]["merge"] for r in res)
ix = sum(r["phases_ms"]["index"] for r in res)
per_pair = fill / tot
w = (dm / n) / per_pair
v = (ix / sum(r["rows"][1] for r in res)) / per_pair if os.environ.get("PREFIX") != "0" else 0.0
blocks2 = [abi.shard_tri_rows_costed(eng.lib, 0, n, G, g, w, v)[0] for g in range(G)] + [n]
out = torch.empty((max(shard.tri_pairs(blocks2[g], blocks2[g + 1]) for g in range(G)), 2), dtype=torch.int32, device=dev)
res2 = run_blocks(blocks2)
print(json.dumps({"workload": which, "ranks": G, "cut": "costed", "row_weight": round(w, 1), "prefix_weight": round(v, 1), "blocks": blocks2,
                  "max_ms": max(r["ms"] for r in res2), "pairs_s_if_sharded": tot / (max(r["ms"] for r in res2) * 1e-3),
                  "checksum": [sum(r["sums"][0] for r in res2), sum(r["sums"][1] for r in res2)]}))
