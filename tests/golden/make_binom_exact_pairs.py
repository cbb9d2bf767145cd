#!/usr/bin/env python3
"""Generates tests/golden/binom_exact_pairs.json: EXACT p-values of pValue(x, lenRef, lenQry, kmerSpace,
sketchSize) (CommandDistance.cpp:427-448) for the double r that function computes from the two genome
lengths -- the form in which the tail reaches the DEVICE (finish.hip, mg_finish_*_dev takes lengths
from the tables, not r).  Same construction as make_binom_exact.py (whose exact_tail it uses): the
finite binomial sum at 80 digits from the exact rational value of r, rounded once.

r is computed here with Python floats in the order of the C code (pX, pY, pX*pY/(pX+pY-pX*pY));
Python does not fuse multiply-adds, and the product compiles that expression with contraction off.

Run: python tests/golden/make_binom_exact_pairs.py     (needs mpmath; a few minutes)
"""
import json
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_binom_exact import exact_tail, to_double      # noqa: E402

mp.mp.dps = 80


def r_of(len_ref, len_qry, kmer_space):
    pX = 1.0 / (1.0 + kmer_space / float(len_ref))
    pY = 1.0 / (1.0 + kmer_space / float(len_qry))
    return pX * pY / (pX + pY - pX * pY)


def main():
    rng = np.random.default_rng(20260926)
    lengths = [1000, 4641, 50_000, 502_359, 1_000_000, 4_639_675, 100_000_000, 3_100_000_000, 2 ** 40]
    spaces = [4.0 ** 21, 4.0 ** 16, 4.0 ** 11, 4.0 ** 31, 20.0 ** 9, 4.0 ** 7]
    cases, seen = [], set()
    for ks in spaces:
        for a in range(len(lengths)):
            for b in range(a + 1):
                lr, lq = lengths[a], lengths[b]
                r = r_of(lr, lq, ks)
                if not (0.0 < r < 1.0):
                    continue
                for n in (1000, 400, 37, 10000):
                    if rng.random() < 0.55:
                        continue
                    mean = n * r
                    xs = {1, 2, 5, 41, n // 2, n}
                    xs |= {max(1, int(mean)), int(mean) + 1, int(mean + 3 * (mean * (1 - r)) ** 0.5) + 1}
                    xs |= set(int(v) for v in rng.integers(1, n + 1, 1))
                    # the underflow edge: where the tail crosses 1e-290 ... 1e-330
                    lo, hi = max(1, int(mean) + 1), n
                    if mp.log10(exact_tail(hi, n, r)) < -290:
                        while hi - lo > 1:
                            mid = (lo + hi) // 2
                            if mp.log10(exact_tail(mid, n, r)) < -290:
                                hi = mid
                            else:
                                lo = mid
                        for d in (-1, 0, 3, 6, 9):
                            xs.add(hi + d)
                    for x in sorted(xs):
                        if x < 1 or x > n or (x, n, r) in seen:
                            continue
                        seen.add((x, n, r))
                        e = exact_tail(x, n, r)
                        cases.append({"x": int(x), "n": int(n), "len_ref": int(lr), "len_qry": int(lq),
                                      "kmer_space": float(ks).hex(), "r": float(r).hex(), "exact": to_double(e).hex()})
    with open(os.path.join(HERE, "binom_exact_pairs.json"), "w") as f:
        json.dump(cases, f, indent=0)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
