#!/usr/bin/env python3
"""Workload for rocprofv3 passes: the two hot kernels at BASELINE shapes (scaled so a
profiled pass stays short) + a calibration copy of known size for FETCH_SIZE/WRITE_SIZE.
  compare: triangle on N sketches (default 40000 -> 8.0e8 pairs/launch), 2 launches
  sketch : G genomes x 1 Mbp (default 2000), 2 launches
  calib  : torch copy of 2 GiB (known 2 GiB read + 2 GiB written), 2 launches"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=40000)
    ap.add_argument("--genomes", type=int, default=2000)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    import torch
    from mash_amd import abi
    from workloads import synth_torch
    dev = torch.device("cuda", 0)
    eng = abi.MashGpu(0)
    hashes, nhash, lengths = synth_torch.clustered_sketch_table(a.n, 1000, clusters=max(1, a.n // 100), device=dev)
    table = eng.table_wrap(hashes.data_ptr(), nhash.data_ptr(), lengths.data_ptr(), a.n, 1000)
    out = torch.empty((a.n * (a.n - 1) // 2, 2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for _ in range(a.reps):
        eng.compare_tri_dev(table, 0, a.n, out.data_ptr())
    eng.synchronize()
    del out
    bases = synth_torch.synthetic_genomes(0, a.genomes, 1_000_000, device=dev)
    off = np.arange(a.genomes + 1, dtype=np.uint64) * np.uint64(1_000_000)
    sk = torch.empty((a.genomes, 1000), dtype=torch.int64, device=dev)
    nh = torch.empty(a.genomes, dtype=torch.int32, device=dev)
    p = eng.params(k=21, s=1000)
    torch.cuda.synchronize()
    for _ in range(a.reps):
        eng.sketch_dev(bases.data_ptr(), a.genomes * 1_000_000, off, p, sk.data_ptr(), nh.data_ptr())
    eng.synchronize()
    x = torch.empty(2 << 30, dtype=torch.uint8, device=dev).view(torch.int32)
    y = torch.empty_like(x)
    x.fill_(1)
    torch.cuda.synchronize()
    for _ in range(a.reps):
        y.copy_(x)
    torch.cuda.synchronize()
    print("prof workload done: compare pairs/launch", a.n * (a.n - 1) // 2, "sketch bases/launch", a.genomes * 1_000_000)


if __name__ == "__main__":
    main()
