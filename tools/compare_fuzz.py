#!/usr/bin/env python3
"""Differential fuzz of the compare engines against each other on random tables built to hit the
rare paths of the tile table: values shared by many rows (duplicate runs), buckets with more than
four distinct prefixes (oversize), 64-bit values that agree in the probe prefix but differ below it
(the verifying loads and the non-clean tile path), ragged and empty sketches, tiny and wide value
ranges (density classes), triangle and rectangle.  The generic kernel (one wave per pair, binary
search in global memory, compare.hip) is the independent side; plain tiles, value windows, the
inverted-index engine must give the same {numer, denom} for every pair.

    python tools/compare_fuzz.py [--n 120] [--seed 1] [--seconds 200]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mash_amd import abi

PAD = 0xFFFFFFFFFFFFFFFF


def make_table(rng, n, s):
    """rows of up to s distinct ascending u64, padded; returns (hashes u64[n, s], nhash i32[n])"""
    style = rng.choice(["shared_pool", "prefix_twins", "dense_small", "mixed_density", "clades", "random"])
    out = np.full((n, s), PAD, dtype=np.uint64)
    nh = np.zeros(n, dtype=np.int32)
    if style == "shared_pool":                      # a pool a little larger than s: every value sits in most rows
        pool = rng.integers(0, 2 ** 63, int(s * rng.uniform(1.05, 2.0)) + 2, dtype=np.uint64)
    elif style == "prefix_twins":                   # few distinct high words, random low words: equal prefixes, different values
        hi = rng.integers(0, 2 ** 31, max(2, s // 3), dtype=np.uint64) << np.uint64(32)
        pool = (hi[rng.integers(0, len(hi), 3 * s + 8)] | rng.integers(0, 2 ** 32, 3 * s + 8, dtype=np.uint64))
    elif style == "dense_small":                    # values in a tiny range: buckets overflow, everything collides
        pool = rng.integers(0, 4 * s + 16, 3 * s + 8, dtype=np.uint64)
    elif style == "clades":
        pool = None
    else:
        pool = None
    shift = rng.choice([0, 0, 8, 20, 33])
    for i in range(n):
        k = s if rng.random() < 0.7 else int(rng.integers(0, s + 1))
        if style == "clades":
            base = np.random.default_rng(int(i // max(1, n // 6)) + 1000).integers(0, 2 ** 62, 2 * s + 4, dtype=np.uint64)
            v = np.concatenate([base[rng.random(len(base)) < 0.9], rng.integers(0, 2 ** 62, s // 10 + 1, dtype=np.uint64)])
        elif pool is not None:
            v = pool[rng.random(len(pool)) < rng.uniform(0.5, 1.0)]
            if style != "dense_small" and rng.random() < 0.5:
                v = np.concatenate([v, rng.integers(0, 2 ** 63, s // 4 + 1, dtype=np.uint64)])
        else:
            v = rng.integers(0, 2 ** 64 - 2, 2 * s + 4, dtype=np.uint64)
            if style == "mixed_density" and i % 3 == 0:
                v = v >> np.uint64(int(rng.integers(5, 30)))
        v = np.unique(v >> np.uint64(shift))[:k]
        out[i, : len(v)] = v
        nh[i] = len(v)
    return style, out, nh


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=200)
    ap.add_argument("--only", type=int, default=-1, help="replay one case of the seed (the others only consume random numbers)")
    a = ap.parse_args()
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    eng = abi.MashGpu(0, stream=torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(a.seed)
    engines = [("generic", {"MASHGPU_COMPARE_KERNEL": "generic"}), ("plain", {"MASHGPU_COMPARE_KERNEL": "merged", "MASHGPU_COMPARE_WINDOWS": "0"}),
               ("windows", {"MASHGPU_COMPARE_KERNEL": "merged", "MASHGPU_COMPARE_WINDOWS": "1"}), ("default", {}), ("sparse", {"MASHGPU_COMPARE_KERNEL": "sparse"}),
               ("join", {"MASHGPU_COMPARE_KERNEL": "join"}),
               # the per-table job of a large table: the table's derived data dropped, the fill beside the index build at some pace
               ("sparse, fill beside the build", {"MASHGPU_COMPARE_KERNEL": "sparse", "MASHGPU_FILL_ASIDE_MIN_PAIRS": "1", "MASHGPU_FILL_ASIDE": None}),
               ("default, fill beside the build", {"MASHGPU_FILL_ASIDE_MIN_PAIRS": "1", "MASHGPU_FILL_ASIDE": None})]
    knobs = ("MASHGPU_COMPARE_KERNEL", "MASHGPU_COMPARE_WINDOWS", "MASHGPU_FILL_ASIDE_MIN_PAIRS", "MASHGPU_FILL_ASIDE", "MASHGPU_FILL_ASIDE_AT_SORT")
    t0 = time.time()
    bad = ran = 0
    for case in range(a.n):
        if time.time() - t0 > a.seconds:
            break
        s = int(rng.choice([1, 2, 7, 33, 100, 257, 600, 1000, 1500, 2300, 4000]))
        n = int(rng.integers(2, max(3, min(900, 400000 // s))))
        style, h, nh = make_table(rng, n, s)
        rect = rng.random() < 0.3
        nq = int(rng.integers(1, n + 1)) if rect else 0
        perm = rng.permutation(n)[:nq] if rect else None
        if a.only >= 0 and case != a.only:
            continue
        ht = torch.from_numpy(h.view(np.int64)).to(dev)
        nt = torch.from_numpy(nh).to(dev)
        lt = torch.full((n,), 1000000, dtype=torch.int64, device=dev)
        t = eng.table_wrap(ht.data_ptr(), nt.data_ptr(), lt.data_ptr(), n, s, keep=(ht, nt, lt))
        if rect:
            qi = torch.from_numpy(perm).to(dev)
            hq, nq_t, lq = ht[qi].contiguous(), nt[qi].contiguous(), lt[qi].contiguous()
            tq = eng.table_wrap(hq.data_ptr(), nq_t.data_ptr(), lq.data_ptr(), nq, s, keep=(hq, nq_t, lq))
            pairs = nq * n
        else:
            pairs = n * (n - 1) // 2
        ref = None
        paces = [str(rng.choice(["1,0", "2,3", "7,0,64", "64,1", "64,40", "300,0,128", "4096,0"])) for _ in range(2)]
        late = [bool(rng.random() < 0.4) for _ in range(2)]
        for name, env in engines:
            for k in knobs:
                os.environ.pop(k, None)
            env = dict(env)
            if "MASHGPU_FILL_ASIDE" in env:
                env["MASHGPU_FILL_ASIDE"] = paces.pop()
                if late.pop():                             # (the fill waits for the bucket sorts, as behind a long build)
                    env["MASHGPU_FILL_ASIDE_AT_SORT"] = "1"
                t.invalidate()
            os.environ.update(env)
            out = torch.full((max(pairs, 1), 2), -1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()                             # the fill ran on torch's stream
            try:
                if rect:
                    eng.compare_rect_dev(t, tq, 0, nq, out.data_ptr())
                else:
                    eng.compare_tri_dev(t, 0, n, out.data_ptr())
                eng.synchronize()
            except abi.MashGpuError as e:
                if name in ("windows", "join") and ("unsupported" in str(e).lower() or "cannot take" in str(e).lower()):
                    continue
                print("ERROR case %d %s n=%d s=%d %s: %s" % (case, style, n, s, name, e))
                bad += 1
                continue
            got = out[:pairs].cpu().numpy()
            if ref is None:
                ref = got
            elif not np.array_equal(got, ref):
                w = np.nonzero((got != ref).any(axis=1))[0]
                print("DIFF case %d style %s n=%d s=%d %s engine %s: %d of %d pairs differ, first %d: got %s want %s"
                      % (case, style, n, s, "rect q=%d" % nq if rect else "tri", name, len(w), pairs, w[0], got[w[0]], ref[w[0]]))
                bad += 1
                if a.only >= 0:
                    for x in w[:12]:
                        if rect:
                            i, j = int(perm[x // n]), int(x % n)
                        else:
                            i = int((1 + (1 + 8 * int(x)) ** 0.5) / 2)
                            while i * (i - 1) // 2 > x: i -= 1
                            while (i + 1) * i // 2 <= x: i += 1
                            j = int(x) - i * (i - 1) // 2
                        print("   pair", x, "rows", i, j, "nh", nh[i], nh[j], "A", [hex(int(v)) for v in h[i, :4]], "B", [hex(int(v)) for v in h[j, :4]], "got", got[x], "want", ref[x])
        ran += 1
        t.free()
        if rect:
            tq.free()
    for k in knobs:
        os.environ.pop(k, None)
    print("tables: %d  engine disagreements: %d  [seed %d, %.0f s]" % (ran, bad, a.seed, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
