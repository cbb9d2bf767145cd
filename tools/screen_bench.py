#!/usr/bin/env python3
"""BASELINE config 4 on one GPU: screen 10^7 synthetic 150-bp reads against a 100 000-sketch
database (first 1000 rows = real sketches of the C2 genomes the reads come from, the rest =
C3 synthetic sketches that fill the hash table to RefSeq scale).

    python tools/screen_bench.py [--reads 10000000] [--db 100000] [--steps 3]

Prints one JSON object (reads/s, bp/s, kernel-only split via HIP events around the calls).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from workloads import synth_torch  # noqa: E402
from mash_amd.abi import MashGpu  # noqa: E402

K, S, L, RL = 21, 1000, 1_000_000, 150


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--db", type=int, default=100_000)
    ap.add_argument("--src", type=int, default=1000, help="genomes the reads are sampled from")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2_000_000, help="reads per mg_screen_add_dev call")
    a = ap.parse_args()
    torch.cuda.init()
    eng = MashGpu(0)
    p = eng.params(k=K, s=S)
    # database
    genomes = synth_torch.synthetic_genomes(0, a.src, L, device="cuda", stride=40000)   # unrelated genomes
    gh = torch.empty((a.src, S), dtype=torch.int64, device="cuda")
    gn = torch.empty(a.src, dtype=torch.int32, device="cuda")
    off = np.arange(a.src + 1, dtype=np.uint64) * np.uint64(L)
    torch.cuda.synchronize()
    eng.sketch_dev(genomes.data_ptr(), a.src * L, off, p, gh.data_ptr(), gn.data_ptr())
    eng.synchronize()
    rest = max(0, a.db - a.src)
    fh, fn, fl = synth_torch.clustered_sketch_table(max(rest, 1), S, clusters=max(1, rest // 100), device="cuda")
    hashes = torch.cat([gh, fh[:rest]], 0).contiguous()
    nhash = torch.cat([gn, fn[:rest].to(torch.int32)], 0).contiguous()
    lengths = torch.full((a.src + rest,), L, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    db = eng.table_wrap(hashes.data_ptr(), nhash.data_ptr(), lengths.data_ptr(), a.src + rest, S)
    # reads, resident in HBM, in batches
    batches = []
    for b0 in range(0, a.reads, a.batch):
        n = min(a.batch, a.reads - b0)
        batches.append(synth_torch.synthetic_reads(genomes, n, RL, seed=1000 + b0))
    torch.cuda.synchronize()
    nb = sum(int(b.numel()) for b in batches)
    res = {"reads": a.reads, "read_len": RL, "db_sketches": a.src + rest, "bytes": nb}
    t0 = time.perf_counter()
    sc = eng.screen_open(db, p)
    eng.synchronize()
    res["table_build_s"] = time.perf_counter() - t0
    sc.close()
    times = []
    for step in range(a.steps + 1):
        sc = eng.screen_open(db, p)
        eng.synchronize()
        t0 = time.perf_counter()
        for b in batches:
            sc.add_dev(b.data_ptr(), int(b.numel()))
        counts = torch.empty((a.src + rest) * S, dtype=torch.int32, device="cuda")
        sc.counts_dev(counts.data_ptr())
        eng.synchronize()
        dt = time.perf_counter() - t0
        _, mix, _ = sc.finish(want_counts=False, want_distinct=False)
        sc.close()
        if step:
            times.append(dt)
    dt = min(times)
    c = counts.view(a.src + rest, S)
    shared = (c[: a.src] > 0).sum(1).float()
    res.update({"seconds": dt, "reads_per_s": a.reads / dt, "bp_per_s": a.reads * RL / dt,
                "mean_shared_src": float(shared.mean()), "shared_rest_max": int((c[a.src:] > 0).sum(1).max()) if rest else 0,
                "mix_n": int(len(mix))})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
