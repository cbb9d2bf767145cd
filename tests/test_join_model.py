"""CPU model of the join engine (mash_amd/csrc/compare_join.hip), step for step in numpy, against the oracle's merge loop
(compareSketches, CommandDistance.cpp:347-385).

The engine is made for collections in the MIDDLE of the similarity range (one species: every pair shares a tenth to a half
of its values, nothing is a near-copy): there every pair is a candidate of the inverted index, and a merge of ~2 s steps per
pair is what the reference pays.  The join engine pays per SHARED value instead.  Its claim:

    walk the values two rows have in common in ascending order, keep c = the common values counted so far; the value v at
    position p_i of row i and p_j of row j has rank p_i + p_j - c in the union of the two rows, and the reference's loop
    counts it iff that rank is below s (the loop's `denom` IS the rank of the value it looks at); once one value fails
    every later one fails.  numer = c at the end, denom = min(s, |A| + |B| - c).

The structure around it: rows in blocks of B; per block the list of its (value, row, position) entries in value order, as
groups of equal values; a tile (block I, block J) joins the two lists and updates c[i][j] for the holders of every matched
group -- value after value, so every pair sees its common values in ascending order."""
import numpy as np
import pytest

from oracle import pyoracle
from workloads import synth

from test_sparse_model import build_index, classes_of

PAD = np.uint64(0xFFFFFFFFFFFFFFFF)


def block_lists(ix, rep, cnt, n, B, only_shared=True):
    """per block: groups [(code >> 1, [(row, pos), ...])] ascending by code, rows ascending inside a group.  A copy of an
    earlier row has no entries in the index: it takes its representative's codes (jn_emit_kernel)."""
    off = ix["off"]
    lists = []
    for b0 in range(0, n, B):
        ent = []
        for r in range(b0, min(b0 + B, n)):
            src = int(rep[r])
            codes = ix["code_of"][off[src]: off[src + 1]]
            for p, c in enumerate(codes):
                if (c & 1) or not only_shared:
                    ent.append((int(c) >> 1, r, p))
        ent.sort()                                          # (code, row): what the stable sort of row-major entries gives
        groups = []
        for c, r, p in ent:
            if groups and groups[-1][0] == c:
                groups[-1][1].append((r, p))
            else:
                groups.append((c, [(r, p)]))
        lists.append(groups)
    return lists


def join_tile(LI, LJ, s, c):
    """merge-join of two block lists; c: dict (i, j) -> count, updated in value order"""
    a = b = 0
    while a < len(LI) and b < len(LJ):
        if LI[a][0] < LJ[b][0]:
            a += 1
        elif LJ[b][0] < LI[a][0]:
            b += 1
        else:
            for (i, pi) in LI[a][1]:
                for (j, pj) in LJ[b][1]:
                    if j < i and pi + pj - c.get((i, j), 0) < s:
                        c[(i, j)] = c.get((i, j), 0) + 1
            a += 1
            b += 1


def model_triangle(table, nhash, s, B):
    n = table.shape[0]
    cnt, rep, _ = classes_of(table, nhash, s)
    idx_nhash = np.where(rep == np.arange(n), cnt, 0).astype(nhash.dtype)
    ix = build_index(table, idx_nhash, s)
    # a value held by ONE row of the index and by that row's copies carries no "shared" bit (copies stay out of the index):
    # a table with copies lists every entry
    L = block_lists(ix, rep, cnt, n, B, only_shared=bool(np.all(rep == np.arange(n))))
    c = {}
    for bi in range(len(L)):
        for bj in range(bi + 1):
            join_tile(L[bi], L[bj], s, c)
    numer, denom = [], []
    for i in range(n):
        for j in range(i):
            k = c.get((i, j), 0)
            numer.append(k)
            denom.append(min(s, int(cnt[i]) + int(cnt[j]) - k))
    return np.array(numer, dtype=np.uint32), np.array(denom, dtype=np.uint32)


def _check(table, nhash, s, B):
    orc = pyoracle.Oracle()
    n = table.shape[0]
    t = np.full((n, s), PAD, dtype=np.uint64)
    w = min(s, table.shape[1])
    t[:, :w] = table[:, :w]
    nh = np.minimum(nhash, s).astype(np.uint32)
    for i in range(n):
        t[i, nh[i]:] = PAD
    numer, denom, _, _ = orc.triangle(t, nh, np.full(n, 1000, np.uint64), 0, n, 21, 4.0 ** 21)
    mn, md = model_triangle(t, nh, s, B)
    assert np.array_equal(mn, numer), np.flatnonzero(mn != numer)[:10]
    assert np.array_equal(md, denom), np.flatnonzero(md != denom)[:10]
    return int(numer.sum())


@pytest.mark.parametrize("B", [1, 3, 8])
def test_one_species(B):
    table, nh, _ = synth.species_sketches(40, 64, seed=3)
    assert _check(table, nh, 64, B) > 40 * 39 // 2 * 3          # pairs share values by the dozen


@pytest.mark.parametrize("s", [16, 50])
def test_clusters_with_short_empty_and_identical_rows(s):
    table, nh, _ = synth.clustered_sketches(36, 50, clusters=3, seed=5, pool=70, private=20)
    nh = nh.copy()
    nh[4] = 0
    nh[9] = 7
    nh[20] = 31
    table = table.copy()
    table[11] = table[2]; nh[11] = nh[2]                         # a copy, and a copy of a short row
    table[30] = table[9]; nh[30] = nh[9]
    _check(table, nh, s, 5)


def test_rows_of_many_lengths_sharing_their_smallest_values():
    """the rank test with positions that differ between the rows: short and long rows over one pool"""
    rng = np.random.default_rng(11)
    s = 40
    pool = np.sort(rng.choice(1 << 40, 120, replace=False).astype(np.uint64))
    n = 30
    table = np.full((n, s), PAD, dtype=np.uint64)
    nh = np.zeros(n, np.uint32)
    for i in range(n):
        k = int(rng.integers(0, s + 1))
        v = np.sort(rng.choice(pool, k, replace=False)) if k else np.zeros(0, np.uint64)
        table[i, :k] = v
        nh[i] = k
    _check(table, nh, s, 4)


def test_random_rows_share_nothing():
    table, nh, _ = synth.random_sketches(12, 32, seed=2)
    assert _check(table, nh, 32, 4) == 0
