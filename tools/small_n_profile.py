#!/usr/bin/env python3
"""Where a triangle call spends its time at small and medium n: wall per call against the summed
kernel time of the compare launches (library profiling events), for the default dispatch.
usage: python tools/small_n_profile.py [--s 1000] n..."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mash_amd import abi
from workloads import synth_torch

ap = argparse.ArgumentParser()
ap.add_argument("--s", type=int, default=1000)
ap.add_argument("n", type=int, nargs="*", default=[1000, 3000, 10000, 20000, 40000])
a = ap.parse_args()
dev = torch.device("cuda", 0)
eng = abi.MashGpu(0)
for n in a.n:
    h, nh, ln = synth_torch.clustered_sketch_table(n, a.s, clusters=max(1, n // 100), device=dev)
    torch.cuda.synchronize()
    t = eng.table_wrap(h.data_ptr(), nh.data_ptr(), ln.data_ptr(), n, a.s, keep=(h, nh, ln))
    pairs = n * (n - 1) // 2
    out = torch.empty((pairs, 2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng.compare_tri_dev(t, 0, n, out.data_ptr()); eng.synchronize()          # first call: classes, prefix images
    eng.prof_enable(True)
    best = None
    for _ in range(5):
        eng.prof_reset()
        t0 = time.perf_counter()
        eng.compare_tri_dev(t, 0, n, out.data_ptr())
        t1 = time.perf_counter()
        eng.synchronize()
        t2 = time.perf_counter()
        k_ms, launches = eng.prof_avg_ms("compare")
        rec = ((t2 - t0) * 1e3, (t1 - t0) * 1e3, k_ms * launches, launches)
        if best is None or rec[0] < best[0]:
            best = rec
    eng.prof_enable(False)
    print("n %6d s %5d: wall %8.3f ms (call returns after %8.3f ms), kernels %8.3f ms in %d launches -> %.2f Gpairs/s wall, %.2f kernel-only"
          % (n, a.s, best[0], best[1], best[2], best[3], pairs / best[0] / 1e6, pairs / max(best[2], 1e-9) / 1e6))
    t.free()
