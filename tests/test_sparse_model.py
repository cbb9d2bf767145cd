"""CPU model of the inverted-index compare engine (mash_amd/csrc/compare_sparse.hip), step for
step in numpy, against the oracle's merge loop (compareSketches, CommandDistance.cpp:347-385).

It pins the three claims the engine rests on, on tables with every edge the domain has (empty,
short and identical rows, values shared by many rows, values at the top of the hash range):
  1. a pair that shares no hash gives {0, min(s, |A| + |B|)} -- the fill;
  2. the pairs sharing a hash are exactly what the runs of the sorted (value, row) index yield
     -- the discovery;
  3. the reference's loop run on codes (2 * the first sorted position of the value's group in the
     table's index -- ordered and equal exactly as the values are; for rect queries the code scheme
     of sp_locate_kernel) takes the same branches as on the 64-bit values -- the merge.
The GPU tests then only have to show that the kernels do what this model does."""
import numpy as np
import pytest

from workloads import synth

PAD = np.uint64(0xFFFFFFFFFFFFFFFF)


def build_index(table, nhash, s):
    n = table.shape[0]
    cnt = np.minimum(np.minimum(nhash, table.shape[1]), s).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(cnt)])
    keys = np.concatenate([table[i, : cnt[i]] for i in range(n)]) if off[-1] else np.zeros(0, np.uint64)
    row_of = np.repeat(np.arange(n), cnt)
    order = np.argsort(keys, kind="stable")               # stable: rows ascend inside a value
    ks = keys[order]
    head = np.ones(len(ks), dtype=bool)
    head[1:] = ks[1:] != ks[:-1]
    grp = np.cumsum(head) - 1                              # group id per sorted position
    gstart = np.flatnonzero(head)
    gstart = np.concatenate([gstart, [len(ks)]])
    sorted_rows = row_of[order]
    pos_of = np.empty(len(ks), dtype=np.int64)
    pos_of[order] = np.arange(len(ks))                     # entry -> sorted position
    rank_of = grp[pos_of]                                  # entry -> dense rank of its value
    lo_of = gstart[rank_of]                                # entry -> first sorted position of its value (code / 2)
    gend = np.zeros(len(ks) + 1, dtype=np.int64)           # defined at group starts: one past the group's last position
    gend[gstart[:-1]] = gstart[1:]
    # the code of an entry: 2 x group start, + 1 if another row holds the value too (the low bit travels with the value)
    code_of = 2 * lo_of + (gend[lo_of] - lo_of >= 2)
    return dict(cnt=cnt, off=off, ks=ks, grp=grp, gstart=gstart, sorted_rows=sorted_rows, pos_of=pos_of,
                rank_of=rank_of, lo_of=lo_of, gend=gend, code_of=code_of)


def merge_codes(a, b, s):
    i = j = common = denom = 0
    na, nb = len(a), len(b)
    while denom < s and i < na and j < nb:
        if a[i] < b[j]:
            i += 1
        elif b[j] < a[i]:
            j += 1
        else:
            i += 1; j += 1; common += 1
        denom += 1
    if denom < s:
        denom = min(s, denom + (na - i) + (nb - j))
    return common, denom


def classes_of(table, nhash, s):
    """Identical rows: rep[row] = first row holding the same values (empty rows stay on their own)."""
    n = table.shape[0]
    cnt = np.minimum(np.minimum(nhash, table.shape[1]), s).astype(np.int64)
    rep = np.arange(n)
    first = {}
    for i in range(n):
        if cnt[i] == 0:
            continue
        key = table[i, : cnt[i]].tobytes()
        rep[i] = first.setdefault(key, i)
    members = {}
    for i in range(n):
        members.setdefault(int(rep[i]), []).append(i)
    return cnt, rep, members


def model_triangle(table, nhash, s, rb, re, dedup=True):
    """dedup: copies of an earlier row stay out of the index; discovery marks CLASSES (a copy walks the
    whole run of each of its representative's values, up to its own row) and expands them to rows; a
    pair inside a class is {n, n} without a merge -- as compare_sparse.hip does it."""
    cnt, rep, members = classes_of(table, nhash, s)
    if not dedup:
        rep = np.arange(table.shape[0])
        members = {i: [i] for i in range(table.shape[0])}
    idx_nhash = np.where(rep == np.arange(len(rep)), cnt, 0).astype(nhash.dtype)     # copies: no entries
    ix = build_index(table, idx_nhash, s)
    off = ix["off"]
    numer, denom = [], []
    ncand = 0
    for i in range(rb, re):
        # fill (true counts)
        row_n = np.zeros(i, dtype=np.int64)
        row_d = np.minimum(s, cnt[i] + cnt[:i])
        # discover: classes named by the runs of the entries of the row's representative
        er = int(rep[i])
        marked = set()
        for p in range(ix["cnt"][er]):
            e = off[er] + p
            lo = ix["lo_of"][e]
            hi = ix["pos_of"][e] if er == i else ix["gend"][lo]
            marked.update(int(r) for r in ix["sorted_rows"][lo:hi] if r < i and r != er)
        assert all(rep[c] == c for c in marked)
        cand = sorted(b for c in marked for b in members[c] if b < i)
        ncand += len(cand)
        # merge on codes, through the representatives
        ai = ix["code_of"][off[er]: off[er + 1]]
        for j in cand:
            rj = int(rep[j])
            assert rj != er
            row_n[j], row_d[j] = merge_codes(ai, ix["code_of"][off[rj]: off[rj + 1]], s)
        # pairs inside the row's class: {n, n} (sp_class_pairs_kernel)
        for j in members[er]:
            if j < i:
                row_n[j] = row_d[j] = cnt[i]
        numer.append(row_n); denom.append(row_d)
    return np.concatenate(numer) if numer else np.zeros(0), np.concatenate(denom) if denom else np.zeros(0), ncand


def model_rect(ref, ref_nh, qry, qry_nh, s):
    ix = build_index(ref, ref_nh, s)
    nq, nr = qry.shape[0], ref.shape[0]
    numer = np.zeros((nq, nr), dtype=np.int64)
    denom = np.zeros((nq, nr), dtype=np.int64)
    qcnt = np.minimum(np.minimum(qry_nh, qry.shape[1]), s).astype(np.int64)
    for q in range(nq):
        denom[q] = np.minimum(s, qcnt[q] + ix["cnt"])
        vals = qry[q, : qcnt[q]]
        lb = np.searchsorted(ix["ks"], vals, side="left")
        codes = np.zeros(len(vals), dtype=np.int64)
        cand = set()
        for t, v in enumerate(vals):
            # (the lower bound of a value that occurs IS its group's first position)
            found = lb[t] < len(ix["ks"]) and ix["ks"][lb[t]] == v
            codes[t] = 2 * lb[t] + 1 if found else 2 * lb[t]
            if found:
                cand.update(ix["sorted_rows"][lb[t]: ix["gend"][lb[t]]].tolist())
        # query values between the same two table values share a code: they are only ever compared with
        # table codes, so ascending (not strictly) is all the merge needs
        assert np.all(np.diff(codes) >= 0)
        for r in cand:
            bc = ix["code_of"][ix["off"][r]: ix["off"][r + 1]] | 1          # the merge sets the low bit of every table code
            numer[q, r], denom[q, r] = merge_codes(codes, bc, s)
    return numer, denom


def _edge_table(n, s, seed):
    table, nhash, lengths = synth.clustered_sketches(n, s, clusters=4, seed=seed, pool=int(1.5 * s) + 2,
                                                     private=max(1, int(0.4 * s)))
    nhash[2] = 0
    table[2, :] = PAD
    nhash[5] = min(1, s)
    table[5, nhash[5]:] = PAD
    nhash[9] = max(0, s - 1)
    table[9, nhash[9]:] = PAD
    if s >= 3:
        nhash[11] = s // 3
        table[11, nhash[11]:] = PAD
        nhash[12] = s // 3
        table[12] = table[11]
    table[17] = table[16]
    nhash[17] = nhash[16]
    # a value at the very top of the range (not the padding value) shared by two rows
    if s >= 2:
        for r in (20, 21):
            k = int(nhash[r])
            table[r, k - 1] = np.uint64(0xFFFFFFFFFFFFFFFE)
    return table, nhash, lengths


@pytest.mark.parametrize("dedup", [True, False])
@pytest.mark.parametrize("s", [1, 2, 7, 64, 100])
def test_model_triangle_equals_oracle(oracle, s, dedup):
    n = 40
    table, nhash, lengths = _edge_table(n, s, seed=s)
    # more copies: a class of four spread over the table, a copy of a copy, a copy of a short row
    for dst, src in ((30, 3), (35, 3), (39, 30), (33, 9), (25, 24)):
        table[dst] = table[src]
        nhash[dst] = nhash[src]
    numer, denom, _, _ = oracle.triangle(table, nhash, lengths, 0, n, 21, 4.0 ** 21)
    got_n, got_d, ncand = model_triangle(table, nhash, s, 0, n, dedup)
    assert np.array_equal(got_n, numer) and np.array_equal(got_d, denom)
    # the candidates are exactly the pairs sharing a hash: every other pair has numer 0
    assert ncand + sum(len(v) * (len(v) - 1) // 2 for v in classes_of(table, nhash, s)[2].values()) >= int(np.count_nonzero(numer))
    # a row range
    n2, d2, _, _ = oracle.triangle(table, nhash, lengths, 13, 38, 21, 4.0 ** 21)
    g2n, g2d, _ = model_triangle(table, nhash, s, 13, 38, dedup)
    assert np.array_equal(g2n, n2) and np.array_equal(g2d, d2)


@pytest.mark.parametrize("s", [1, 5, 64])
def test_model_rect_equals_oracle(oracle, s):
    """Queries against a reference table, including query values below, between and above the
    table's values, found and not found, and a smaller sketch size than the tables'."""
    n = 30
    table, nhash, lengths = _edge_table(n, s, seed=100 + s)
    rng = np.random.default_rng(s)
    qtab, qnh, _ = synth.clustered_sketches(6, s, clusters=4, seed=100 + s, pool=int(1.5 * s) + 2, private=max(1, int(0.4 * s)))
    # query rows: one copy of a table row, one all-new row with extreme values, an empty one
    qtab[1] = table[3]; qnh[1] = nhash[3]
    new = np.unique(np.concatenate([rng.integers(0, 1 << 54, s).astype(np.uint64), np.array([0, 2 ** 64 - 2], dtype=np.uint64)]))[:s]
    qtab[2, :] = PAD; qtab[2, : len(new)] = np.sort(new); qnh[2] = len(new)
    qnh[4] = 0; qtab[4, :] = PAD
    got_n, got_d = model_rect(table, nhash, qtab, qnh, s)
    for q in range(6):
        for r in range(n):
            c, d = merge_codes(qtab[q, : qnh[q]].tolist(), table[r, : nhash[r]].tolist(), s)
            assert (got_n[q, r], got_d[q, r]) == (c, d), (q, r)


@pytest.mark.parametrize("pack_min", [32, 43, 64])
def test_pack_mapping_of_candidates_to_lanes(pack_min):
    """The work distribution of sp_merge_pack_kernel (compare_sparse_x.hip) in numpy: candidates of all rows as one line
    of units (a row with candidates takes max(candidates, pack_min) units), item t = units [128 t, 128 t + 128).  Every
    lane finds its row from the rows of the item's probe units (0, pack_min, 2 pack_min, ..., 127) alone -- at most
    one row boundary lies between two probes -- and every candidate of every row is some lane's, exactly once."""
    rng = np.random.default_rng(pack_min)
    for trial in range(30):
        nrows = int(rng.integers(1, 400))
        cnt = rng.integers(0, 200, nrows)
        cnt[rng.random(nrows) < 0.4] = 0                                   # rows without candidates
        if trial % 5 == 0:
            cnt[:] = rng.integers(0, 3, nrows)                             # nearly all rows tiny
        if cnt.sum() == 0:
            cnt[0] = 1
        cost = np.where(cnt > 0, np.maximum(cnt, pack_min), 0)
        inc = np.cumsum(cost)                                              # chunk_inc
        total = int(inc[-1])
        nprobe = (128 + pack_min - 1) // pack_min + 1
        seen = [np.zeros(c, dtype=np.int32) for c in cnt]
        for item in range((total + 127) // 128):
            u0 = item * 128
            probes = [min(u0 + (127 if k == nprobe - 1 else k * pack_min), total - 1) for k in range(nprobe)]
            pslot = [int(np.searchsorted(inc, u, side="right")) for u in probes]   # first slot whose inclusive cost exceeds u
            assert len(set(pslot)) <= nprobe
            for tid in range(128):
                u = u0 + tid
                k = tid // pack_min
                sa, sb = pslot[k], pslot[k + 1]
                slot = sb if (sb != sa and u >= inc[sb - 1]) else sa
                start = inc[slot - 1] if slot else 0
                if u < total:
                    assert slot == int(np.searchsorted(inc, u, side="right")), (trial, item, tid)
                    q = u - start
                    assert q >= 0
                    if q < cnt[slot]:
                        seen[slot][q] += 1
                        # the row's index among the item's staged rows exists
                        distinct = [pslot[0]] + [pslot[j] for j in range(1, nprobe) if pslot[j] != pslot[j - 1]]
                        assert slot in distinct
        for r in range(nrows):
            assert np.all(seen[r] == 1), (trial, r)


# ---- the index's sort on the values' leading bits (sp_tie_find / sp_tie_segments / sp_tie_repair_kernel) ----

def sort_on_leading_bits(keys, idx, begin_bit, walk=1 << 16, max_breaks=1 << 16, max_values=64):
    """What sparse_build_index does when begin_bit > 0, step for step: a STABLE sort that looks at bits >= begin_bit only,
    the breaks between two different values of one prefix, the segment of every FIRST break (walked to both ends), and
    each segment rewritten by full value -- smallest first, entries of one value in the order the sort left them.
    Returns (keys, idx, overflow): overflow = the caller sorts again on every bit."""
    order = np.argsort(keys >> np.uint64(begin_bit), kind="stable")
    k, ix = keys[order].copy(), idx[order].copy()
    E = len(k)
    pre = k >> np.uint64(begin_bit)
    breaks = [p for p in range(1, E) if pre[p] == pre[p - 1] and k[p] != k[p - 1]]
    if len(breaks) > max_breaks:
        return k, ix, True
    segs = []
    for p in breaks:
        kp = k[p - 1]
        q, first = p - 1, True
        while q > 0 and pre[q - 1] == pre[p]:
            if k[q - 1] != kp:
                first = False                               # an earlier break of the same segment does the work
                break
            q -= 1
            if p - q > walk:
                return k, ix, True
        if not first:
            continue
        e = p
        while e < E and pre[e] == pre[p]:
            e += 1
            if e - p > walk:
                return k, ix, True
        segs.append((q, e))
    out_k, out_i = k.copy(), ix.copy()
    for s0, s1 in segs:
        vals = np.unique(k[s0:s1])
        if len(vals) > max_values:
            return k, ix, True
        w = s0
        for v in vals:                                      # ascending; entries of one value in their present order
            hit = np.flatnonzero(k[s0:s1] == v) + s0
            out_k[w:w + len(hit)] = v
            out_i[w:w + len(hit)] = ix[hit]
            w += len(hit)
        assert w == s1
    return out_k, out_i, False


@pytest.mark.parametrize("bits_kept", [4, 8, 12, 20, 40])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_sort_on_leading_bits_with_tie_repair_is_the_stable_full_sort(seed, bits_kept):
    """rows of ascending values, many values shared (runs of equal keys beside ties), some packed into one prefix"""
    rng = np.random.default_rng(seed)
    end_bit = 44
    pool = rng.integers(1, 1 << end_bit, size=300).astype(np.uint64)
    pool[:40] = (pool[0] & ~np.uint64(0xFFF)) + np.arange(40, dtype=np.uint64) * np.uint64(3)        # forty values, one prefix at >= 12 bits
    rows = [np.unique(rng.choice(pool, size=int(rng.integers(5, 60)))) for _ in range(50)]
    keys = np.concatenate(rows)
    idx = np.concatenate([np.arange(len(r), dtype=np.int64) + 64 * i for i, r in enumerate(rows)])   # row * stride + position: ascends
    full = np.argsort(keys, kind="stable")
    begin_bit = end_bit - bits_kept
    k, ix, overflow = sort_on_leading_bits(keys, idx, begin_bit)
    if overflow:
        assert bits_kept <= 8                               # too few bits for this table: every bit is sorted instead
        return
    assert np.array_equal(k, keys[full]) and np.array_equal(ix, idx[full])
    # (rows ascend inside a value: what discovery's "rows below mine" relies on)
    same = k[1:] == k[:-1]
    assert np.all(ix[1:][same] > ix[:-1][same])


def test_sort_on_leading_bits_gives_up_on_long_or_crowded_segments():
    keys = np.arange(1, 200, dtype=np.uint64)               # 199 values, one prefix at begin_bit 8: more than 64 in a segment
    idx = np.arange(len(keys), dtype=np.int64)
    assert sort_on_leading_bits(keys[::-1].copy(), idx, 8)[2] is True
    two = np.array([5] * 50 + [4] * 50, dtype=np.uint64)    # two values, the walk cut short
    assert sort_on_leading_bits(two, np.arange(100, dtype=np.int64), 8, walk=16)[2] is True
    k, ix, ov = sort_on_leading_bits(two, np.arange(100, dtype=np.int64), 8)
    assert not ov and np.array_equal(k, np.sort(two)) and np.array_equal(ix, np.concatenate([np.arange(50, 100), np.arange(50)]))
