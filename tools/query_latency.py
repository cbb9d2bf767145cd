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
from mash_amd import abi
from workloads import synth_torch


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
            if not env:                                          # kernel time inside the call (HIP events)
                eng.prof_enable(True); eng.prof_reset()
                eng.compare_rect_dev(ref, qry, 0, nq, out.data_ptr())
                torch.cuda.synchronize()
                case["kernel_ms"] = round(eng.prof_avg_ms("compare")[0], 4)
                eng.prof_enable(False)
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
    # the other half of a query: sketching ONE genome handed over in host memory (mg_sketch_host)
    import numpy as np
    rng = np.random.default_rng(3)
    prm = eng.params(k=21, s=a.s)
    res["sketch_one"] = []
    for nb in (100_000, 5_000_000, 100_000_000):
        bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, nb)]
        off = np.array([0, nb], dtype=np.uint64)
        eng.sketch_host_raw(bases, off, prm)
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            eng.sketch_host_raw(bases, off, prm)
            ts.append(time.perf_counter() - t)
        eng.prof_enable(True); eng.prof_reset()
        eng.sketch_host_raw(bases, off, prm)
        kms = eng.prof_avg_ms("sketch")[0]
        eng.prof_enable(False)
        dbuf = torch.from_numpy(bases).to(dev)
        hb = torch.empty((1, a.s), dtype=torch.int64, device=dev); nb_ = torch.empty(1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        td = []
        for _ in range(5):
            t = time.perf_counter()
            eng.sketch_dev(dbuf.data_ptr(), nb, off, prm, hb.data_ptr(), nb_.data_ptr())
            torch.cuda.synchronize()
            td.append(time.perf_counter() - t)
        res["sketch_one"].append({"bases": nb, "ms": round(1e3 * sorted(ts)[2], 4), "chunk_kernel_ms": round(kms, 4),
                                  "dev_call_ms": round(1e3 * sorted(td)[2], 4)})
    # whole query, host to host: sketch one 5 Mbp genome, upload it as a 1-row table, compare it
    # with the resident database, bring the counts back, distances + p-values on the host
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 5_000_000)]
    off = np.array([0, len(bases)], dtype=np.uint64)
    ref_len = ln.cpu().numpy().astype(np.uint64)
    def one_query():
        qh, qn = eng.sketch_host_raw(bases, off, prm)
        qt = eng.table_upload(qh, qn, np.array([len(bases)], np.uint64))
        counts = eng.compare_rect_host(ref, qt)
        fin = eng.finish_rect(counts, ref_len, np.array([len(bases)], np.uint64), 21, 4.0 ** 21, max_d=0.1)
        qt.free()
        return fin
    one_query()
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        one_query()
        ts.append(time.perf_counter() - t)
    res["query_host_to_host_ms"] = round(1e3 * sorted(ts)[2], 4)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
