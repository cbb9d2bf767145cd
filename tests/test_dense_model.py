"""CPU model of the DENSE GROUP engine (mash_amd/csrc/compare_dense.hip), step for step in numpy, against the
reference's merge loop (compareSketches, CommandDistance.cpp:347-385).

A group is any set of rows (for speed: rows that are near-copies of each other; for correctness: any).  Its UNIVERSE is
the sorted list of the values held by at least two of its rows; a row becomes a bit mask over the universe plus its
EXTRAS -- values no other row of the group holds, which can never be common -- each recorded only by its gap (the number
of universe values below it).  For a pair the loop of the reference counts the common values among the first s values of
the union; in universe order that is: common = popcount(A & B) over the universe positions e with f(e) < s, where
f(e) = union bits before e + extras of either row with gap <= e, and denom = min(s, |A u B|).  Words of 64 universe
positions are taken whole while the count at their end stays <= s; the word in which s is reached is resolved by a
bisection over its bit positions.  This file pins exactly that arithmetic (incl. the per-word cumulative extra counts
the kernel reads) before any kernel runs."""
import numpy as np
import pytest

from test_sparse_model import merge_codes


def encode_group(rows):
    """rows: list of ascending distinct uint64 arrays -> (universe, [(mask words, cx, gaps)])"""
    allv = np.concatenate(rows) if rows else np.zeros(0, np.uint64)
    vals, counts = np.unique(allv, return_counts=True)
    U = vals[counts >= 2]
    u = len(U)
    W = (u >> 6) + 1                                        # a gap index u falls into word u >> 6 <= W - 1
    enc = []
    for r in rows:
        idx = np.searchsorted(U, r, side="left")
        found = np.zeros(len(r), dtype=bool)
        if u:
            found = (idx < u) & (U[np.minimum(idx, u - 1)] == r)
        mask = np.zeros(W, dtype=np.uint64)
        for e in idx[found]:
            mask[e >> 6] |= np.uint64(1) << np.uint64(e & 63)
        gaps = idx[~found].astype(np.int64)                  # ascending, because the row ascends
        assert np.all(np.diff(gaps) >= 0)
        cx = np.array([int(np.sum(gaps < 64 * w)) for w in range(W + 1)], dtype=np.int64)
        assert cx[W] == len(gaps)
        enc.append((mask, cx, gaps))
    return U, enc


def popc(x):
    return bin(int(x)).count("1")


def dense_pair(ea, eb, s):
    (ma, cxa, ga), (mb, cxb, gb) = ea, eb
    W = len(ma)
    pu = common = 0
    for w in range(W):
        un, an = int(ma[w] | mb[w]), int(ma[w] & mb[w])
        F = pu + popc(un) + int(cxa[w + 1]) + int(cxb[w + 1])
        if F <= s:
            common += popc(an)
            pu += popc(un)
            if F == s:
                return common, s
            continue
        Fprev = pu + int(cxa[w]) + int(cxb[w])
        assert Fprev < s
        xa = ga[cxa[w]: cxa[w + 1]] - 64 * w                # offsets of the extras inside this word (0 .. 63)
        xb = gb[cxb[w]: cxb[w + 1]] - 64 * w
        assert np.all((xa >= 0) & (xa < 64)) and np.all((xb >= 0) & (xb < 64))

        def f(t):
            return Fprev + popc(un & ((1 << t) - 1)) + int(np.sum(xa <= t)) + int(np.sum(xb <= t))
        lo, hi = 0, 63                                       # smallest t with f(t) >= s (exists: f(63) >= F - 1 >= s)
        assert f(63) >= s
        while lo < hi:
            mid = (lo + hi) >> 1
            if f(mid) >= s:
                hi = mid
            else:
                lo = mid + 1
        common += popc(an & ((1 << lo) - 1))
        return common, s
    total = pu + len(ga) + len(gb)
    return common, min(s, total)


def _group(rng, m, s, pool_n, keep, private, clump=False, short_every=0):
    pool = np.unique(rng.integers(1, 1 << 40, size=pool_n * 2).astype(np.uint64))[:pool_n]
    pool_n = len(pool)
    rows = []
    for i in range(m):
        own = pool[rng.random(pool_n) < keep]
        if clump and i % 3 == 0:                             # many private values inside ONE gap of the universe
            lo, hi = int(pool[pool_n // 2]), int(pool[pool_n // 2 + 1])
            priv = rng.integers(lo + 1, max(lo + 2, hi), size=private).astype(np.uint64)
        else:
            priv = rng.integers(1, 1 << 40, size=private).astype(np.uint64)
        r = np.unique(np.concatenate([own, priv]))
        k = s if not (short_every and i % short_every == 0) else int(rng.integers(1, s))
        rows.append(r[:k])
    return rows


@pytest.mark.parametrize("m,s,pool_n,keep,private,clump,short_every", [
    (24, 100, 110, 0.97, 3, False, 0),          # near-copies: what the engine is for
    (20, 64, 64, 1.0, 0, False, 0),             # exact copies, universe of exactly one word (gap index == 64 * W boundary)
    (16, 128, 128, 0.95, 0, False, 0),          # universe a multiple of 64
    (18, 100, 300, 0.5, 20, False, 5),          # loosely related, some short rows
    (15, 90, 40, 0.9, 60, True, 4),             # private values clumped into one gap, short rows
    (12, 50, 0, 0.0, 50, False, 3),             # unrelated rows: the universe is empty
    (10, 200, 1000, 0.3, 0, False, 0),          # universe much larger than s
])
def test_dense_pair_equals_reference_merge(m, s, pool_n, keep, private, clump, short_every):
    rng = np.random.default_rng(m * 1000 + s)
    rows = _group(rng, m, s, pool_n, keep, private, clump, short_every) if pool_n else \
        [np.unique(rng.integers(1, 1 << 40, size=private).astype(np.uint64))[: (s if i % 3 else s // 2)] for i in range(m)]
    U, enc = encode_group(rows)
    for a in range(m):
        for b in range(a):
            want = merge_codes(rows[a], rows[b], s)
            got = dense_pair(enc[a], enc[b], s)
            assert got == want, (a, b, got, want)
