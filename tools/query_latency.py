#!/usr/bin/env python3
"""Latency of the serving shape of `mash dist`: a few query sketches against a resident
database table (rows = queries, columns = references).
    python tools/query_latency.py [--refs 100000] [--s 1000]
Prints one JSON object: per query count the wall time of mg_compare_rect_dev (device
resident in/out, synchronous call) and the same with full-length column chunks
(MASHGPU_COMPARE_COLS=16384, the pre-adaptive tiling) for comparison; results are compared."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mash_amd import abi, synth_torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--refs", type=int, default=100000)
    ap.add_argument("--s", type=int, default=1000)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    eng = abi.MashGpu(0, stream=torch.cuda.current_stream().cuda_stream)
    h, nh, ln = synth_torch.clustered_sketch_table(a.refs, a.s, clusters=max(1, a.refs // 100), device=dev)
    torch.cuda.synchronize()
    ref = eng.table_wrap(h.data_ptr(), nh.data_ptr(), ln.data_ptr(), a.refs, a.s, keep=(h, nh, ln))
    res = {"refs": a.refs, "s": a.s, "cases": []}
    for nq in (1, 4, 16, 64, 256, 1024):
        idx = torch.randperm(a.refs, device=dev)[:nq]
        qh, qn, ql = h[idx].contiguous(), nh[idx].contiguous(), ln[idx].contiguous()
        qry = eng.table_wrap(qh.data_ptr(), qn.data_ptr(), ql.data_ptr(), nq, a.s, keep=(qh, qn, ql))
        out = torch.zeros((nq * a.refs, 2), dtype=torch.int32, device=dev)
        case = {"queries": nq}
        keep = None
        for tag, env in (("adaptive_ms", None), ("full_chunks_ms", "16384")):
            if env:
                os.environ["MASHGPU_COMPARE_COLS"] = env
            else:
                os.environ.pop("MASHGPU_COMPARE_COLS", None)
            out.zero_()
            eng.compare_rect_dev(ref, qry, 0, nq, out.data_ptr())       # warm (prefix images, classes)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t = time.perf_counter()
                eng.compare_rect_dev(ref, qry, 0, nq, out.data_ptr())
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t)
            case[tag] = round(1e3 * sorted(ts)[len(ts) // 2], 4)
            if keep is None:
                keep = out.clone()
            else:
                assert torch.equal(keep, out), "tilings disagree"
        os.environ.pop("MASHGPU_COMPARE_COLS", None)
        # a query is one of the references: its own column must be a full match
        o = keep.view(nq, a.refs, 2)
        assert all(int(o[i, int(idx[i]), 0]) == int(o[i, int(idx[i]), 1]) for i in range(min(nq, 16)))
        case["pairs_per_s"] = round(nq * a.refs / (case["adaptive_ms"] * 1e-3))
        res["cases"].append(case)
        qry.free()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
