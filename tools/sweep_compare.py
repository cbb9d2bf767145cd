#!/usr/bin/env python3
"""A/B sweep of compare-kernel variants in ONE process (interleaved rounds).
usage: tools/sweep_compare.py [--n 30000] [--s 1000] [--rounds 3] variant[:ROWS[:COLS]] ...
variant: 0|2|3|4 (group size), merged|sparse|generic (engine), win|winNNN|nowin (value windows)"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=30000)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--mode", default="clustered")
    ap.add_argument("--s", type=int, default=1000)
    ap.add_argument("variants", nargs="*", default=["0"])
    a = ap.parse_args()
    import torch
    from mash_amd import abi
    from workloads import synth_torch
    dev = torch.device("cuda", 0)
    eng = abi.MashGpu(0)
    n = a.n
    hashes, nhash, lengths = synth_torch.clustered_sketch_table(n, a.s, clusters=max(1, n // 100), device=dev, pool=int(1.5 * a.s), private=int(0.4 * a.s),
                                                                contiguous=(a.mode == "contiguous"))
    if a.mode == "identical":
        hashes[:] = hashes[0]
    if a.mode.startswith("mixed"):
        # every 10th (mixed) / every 97th (mixed97) row is a "small genome": its sketch spans the
        # whole 64-bit range instead of the bottom 2^54 (values << 9), like a virus next to bacteria
        step = 97 if a.mode == "mixed97" else 10
        sel = torch.arange(0, n, step, device=dev)
        hashes[sel] = hashes[sel] << 9
    table = eng.table_wrap(hashes.data_ptr(), nhash.data_ptr(), lengths.data_ptr(), n, a.s)
    pairs = n * (n - 1) // 2
    out = torch.empty((pairs, 2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ref_sum = None
    res = {v: [] for v in a.variants}
    for rd in range(a.rounds + 1):
        for v in a.variants:
            parts = v.split(":")
            for key in ("MASHGPU_COMPARE_WINDOWS", "MASHGPU_COMPARE_WIN_TARGET"):
                os.environ.pop(key, None)
            if parts[0].startswith("win") or parts[0] == "nowin":
                # value-window mode on (optionally winNNN = window target) / off
                os.environ.pop("MASHGPU_COMPARE_KERNEL", None)
                os.environ.pop("MASHGPU_COMPARE_VARIANT", None)
                os.environ["MASHGPU_COMPARE_WINDOWS"] = "0" if parts[0] == "nowin" else "1"
                if parts[0] not in ("win", "nowin"):
                    tgt = parts[0][3:]
                    if "k" in tgt:                                  # winNNNkK: window target NNN, group size K
                        tgt, ku = tgt.split("k")
                        os.environ["MASHGPU_COMPARE_VARIANT"] = ku
                    if tgt:
                        os.environ["MASHGPU_COMPARE_WIN_TARGET"] = tgt
            elif parts[0] in ("merged", "sparse", "generic"):
                os.environ["MASHGPU_COMPARE_KERNEL"] = parts[0]
                os.environ.pop("MASHGPU_COMPARE_VARIANT", None)
            else:
                os.environ.pop("MASHGPU_COMPARE_KERNEL", None)
                os.environ["MASHGPU_COMPARE_VARIANT"] = parts[0]
            for key, idx in (("MASHGPU_COMPARE_ROWS", 1), ("MASHGPU_COMPARE_COLS", 2)):
                if len(parts) > idx and parts[idx]:
                    os.environ[key] = parts[idx]
                else:
                    os.environ.pop(key, None)
            out.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.compare_tri_dev(table, 0, n, out.data_ptr())
            eng.synchronize()
            dt = time.perf_counter() - t0
            chk = (int(out[:, 0].to(torch.int64).sum().item()), int(out[:, 1].to(torch.int64).sum().item()))
            if ref_sum is None:
                ref_sum = chk
            assert chk == ref_sum, (v, chk, ref_sum)
            if rd > 0:
                res[v].append(pairs / dt)
    for v in a.variants:
        r = res[v]
        print(f"variant {v:16s} median {np.median(r)/1e6:9.1f} Mpairs/s  min {min(r)/1e6:9.1f}  max {max(r)/1e6:9.1f}")
    print("checksum", ref_sum)

if __name__ == "__main__":
    main()
