#!/usr/bin/env python3
"""Generates tests/golden/binom_exact.json: EXACT binomial upper tails P[X >= x], X ~ Binomial(n, r),
for doubles r -- the quantity behind Mash's p-value (pValue, CommandDistance.cpp:427-448 calls
gsl_cdf_binomial_Q / Boost's complement CDF, neither of which is in the reference tree).

Every value is the finite sum of C(n,j) r^j (1-r)^(n-j) evaluated with mpmath at 80 digits from the
exact rational value of the double r, then rounded ONCE to double (subnormals and 0 included), so the
fixture is independent of any floating-point library.  scipy.stats.binom.sf (Boost-backed) is stored
next to it: the test reports how far a production library is from the exact value, which bounds what
"within 1 ulp of the reference" can mean.

r is handed to the product as set_size / kmer_space (mg_p_value_within); both are stored, and r is the
IEEE quotient of the two, as the C code computes it.

Run: python tests/golden/make_binom_exact.py     (needs mpmath + scipy; minutes)
"""
import json
import os

import mpmath as mp
import numpy as np
from scipy.stats import binom

HERE = os.path.dirname(os.path.abspath(__file__))
mp.mp.dps = 80


def exact_tail(x, n, r):
    """P[X >= x] as an mpf (x >= 1)."""
    r = mp.mpf(r)
    q = 1 - r
    if r == 0:
        return mp.mpf(0)
    if q == 0:
        return mp.mpf(1)
    ratio = r / q

    def term(j):
        return mp.exp(mp.loggamma(n + 1) - mp.loggamma(j + 1) - mp.loggamma(n - j + 1) + j * mp.log(r) + (n - j) * mp.log(q))

    if x > (n + 1) * r:                       # upper tail: terms fall from j = x on
        t = term(x)
        tot = mp.mpf(0)
        j = x
        while j <= n:
            tot += t
            if t < tot * mp.mpf(2) ** -300:
                break
            t = t * (n - j) / (j + 1) * ratio
            j += 1
        return tot
    t = term(x - 1)                           # lower sum downward, then the complement
    tot = mp.mpf(0)
    j = x - 1
    while j >= 0:
        tot += t
        if t < tot * mp.mpf(2) ** -300:
            break
        t = t * j / (n - j + 1) / ratio
        j -= 1
    return 1 - tot


def to_double(v):
    """Round an mpf to the nearest double, subnormals included (float() of mpmath does that)."""
    return float(v)


def main():
    rng = np.random.default_rng(20260925)
    set_size = 1 << 40
    cases = []
    seen = set()

    def add(x, n, r_target):
        if x < 1 or x > n:
            return
        kmer_space = set_size / r_target
        r = set_size / kmer_space           # what the C code computes
        key = (x, n, r)
        if key in seen:
            return
        seen.add(key)
        e = exact_tail(x, n, r)
        cases.append({"x": int(x), "n": int(n), "set_size": set_size, "kmer_space": float(kmer_space).hex(),
                      "r": float(r).hex(), "exact": to_double(e).hex(),
                      "log10_exact": (float(mp.log10(e)) if e > 0 else None),
                      "scipy": float(binom.sf(x - 1, n, r)).hex()})

    rs = (1e-12, 1e-9, 1.0535e-7, 3.3e-6, 1e-4, 2.5e-3, 0.04, 0.3, 0.77, 0.999)
    for n in (1000, 999, 400, 37, 5000, 10000, 100000):
        for r in rs:
            xs = {1, 2, 3, 5, 17, 41, 100, 300, n // 2, n - 1, n}
            xs |= set(int(v) for v in rng.integers(1, n + 1, 2))
            mean = n * r
            xs |= {max(1, int(mean)), int(mean) + 1, int(mean + 3 * (mean * (1 - r)) ** 0.5) + 1}
            # the underflow edge: the x where the tail crosses 1e-290 ... 1e-330 (denormals, then 0)
            lo, hi = max(1, int(mean) + 1), n
            if float(mp.log10(exact_tail(hi, n, set_size / (set_size / r)))) < -290:
                while hi - lo > 1:                       # smallest x with tail < 1e-290
                    mid = (lo + hi) // 2
                    if mp.log10(exact_tail(mid, n, set_size / (set_size / r))) < -290:
                        hi = mid
                    else:
                        lo = mid
                for d in range(-2, 12):
                    xs.add(hi + d)
            for x in sorted(xs):
                add(x, n, r)
    with open(os.path.join(HERE, "binom_exact.json"), "w") as f:
        json.dump(cases, f, indent=0)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
