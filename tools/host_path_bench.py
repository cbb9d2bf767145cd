#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points, and the
thresholded all-pairs path on BASELINE config 3.

    python tools/host_path_bench.py [--n 100000] [--genomes 256]

Prints one JSON object.  bench.py's `value` never includes these transfers.
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
from mash_amd.shard import tri_pairs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--genomes", type=int, default=256)
    ap.add_argument("--max-d", type=float, default=0.05)
    a = ap.parse_args()
    torch.cuda.init()
    eng = MashGpu(0)
    res = {}
    n, s, k = a.n, 1000, 21
    hashes, nhash, lengths = synth_torch.clustered_sketch_table(n, s, clusters=max(1, n // 100), device="cuda")
    hn, nn, ln = hashes.cpu().numpy(), nhash.cpu().numpy(), lengths.cpu().numpy()
    t0 = time.perf_counter()
    t = eng.table_upload(hn, nn, ln)
    res["table_upload_s"] = time.perf_counter() - t0
    # thresholded triangle: whole job, host edge list out
    for rep in range(2):
        t0 = time.perf_counter()
        edges = eng.compare_tri_filter(t, k, a.max_d, capacity=1 << 26)
        dt = time.perf_counter() - t0
    res["filter"] = {"max_d": a.max_d, "pairs": tri_pairs(0, n), "edges": int(len(edges)), "seconds": dt,
                     "pairs_per_s": tri_pairs(0, n) / dt}
    # full host-output path on the last rows (about 2.5e8 pairs)
    rb = int((n * n - 5e8) ** 0.5) if n * n > 5e8 else 0
    out = None
    for rep in range(3):                                   # rep 0 also faults the output pages in
        t0 = time.perf_counter()
        out = eng.compare_tri_host(t, rb, n, out=out)
        dt = time.perf_counter() - t0
    res["tri_host"] = {"rows": [rb, n], "pairs": int(len(out)), "seconds": dt, "pairs_per_s": len(out) / dt}
    t.free()
    # sketch from host buffers
    g, L = a.genomes, 1_000_000
    bases = synth_torch.synthetic_genomes(0, g, L, device="cuda").cpu().numpy().reshape(-1)
    off = np.arange(g + 1, dtype=np.uint64) * np.uint64(bases.size // g)
    p = eng.params(k=k, s=s)
    for rep in range(2):
        t0 = time.perf_counter()
        eng.sketch_host_raw(bases, off, p)
        dt = time.perf_counter() - t0
    res["sketch_host"] = {"bases": int(bases.size), "seconds": dt, "bp_per_s": bases.size / dt}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
