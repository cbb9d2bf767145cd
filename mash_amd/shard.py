"""Row-block sharding of the all-vs-all triangle across the GPUs of one node
(SURVEY.md §8e): every rank holds the whole sketch table (one RCCL broadcast over
xGMI), owns a contiguous block of rows of equal AREA (row i has i pairs, so the
boundaries follow a square-root law), and writes its own slice of the
reference-ordered output.  No collective sits in the data path."""
import math


def equal_area_row_blocks(n, world):
    """Boundaries b[0..world] with b[0]=0, b[world]=n: rank g owns rows [b[g], b[g+1]).
    Pairs in a block = T(b[g+1]) - T(b[g]) with T(x) = x(x-1)/2, balanced to within one row."""
    total = n * (n - 1) // 2
    b = [0]
    for g in range(1, world):
        target = total * g / world
        # smallest r with r(r-1)/2 >= target
        r = int(math.ceil((1.0 + math.sqrt(1.0 + 8.0 * target)) / 2.0))
        while r > 0 and (r - 1) * (r - 2) // 2 >= target:
            r -= 1
        while r * (r - 1) // 2 < target:
            r += 1
        b.append(min(max(r, b[-1]), n))
    b.append(n)
    return b


def weighted_row_blocks(n, world, row_weight):
    """Boundaries when row i costs i + row_weight pair-units (mg_shard_tri_rows_weighted): the inverted-index engine
    fills per pair but discovers and merges per row.  row_weight 0 = equal_area_row_blocks."""
    if not row_weight > 0:
        return equal_area_row_blocks(n, world)
    cost = lambda r: r * (r - 1) // 2 + row_weight * r
    total = cost(n)
    b = [0]
    for g in range(1, world):
        want = total * g / world
        h = row_weight - 0.5
        r = int(max(0.0, min(float(n), -h + math.sqrt(h * h + 2.0 * want))))
        while r > 0 and cost(r) > want:
            r -= 1
        while r < n and cost(r + 1) <= want:
            r += 1
        b.append(min(max(r, b[-1]), n))
    b.append(n)
    return b


def costed_row_blocks(n, world, row_weight, prefix_weight):
    """Boundaries when the block [lo, hi) costs pairs + row_weight (hi - lo) + prefix_weight hi pair-units
    (mg_shard_tri_rows_costed): a rank indexes the rows below its block's end only.  prefix_weight 0 = weighted_row_blocks."""
    if not prefix_weight > 0:
        return weighted_row_blocks(n, world, row_weight)
    w = max(float(row_weight), 0.0)
    v = float(prefix_weight)
    cost = lambda lo, hi: tri_pairs(lo, hi) + w * (hi - lo) + v * hi

    def lay(T):
        b, lo = [0], 0
        for _ in range(world):
            a, e = lo, n
            if cost(lo, e) <= T:
                a = e
            else:
                while e - a > 1:
                    m = a + (e - a) // 2
                    if cost(lo, m) <= T:
                        a = m
                    else:
                        e = m
            if a > lo and cost(lo, a) > T:
                a = lo
            b.append(a)
            lo = a
        return b, lo >= n

    t_lo, t_hi = 0.0, float(cost(0, n))
    for _ in range(200):
        if t_hi - t_lo <= 0.5:
            break
        mid = (t_lo + t_hi) * 0.5
        if lay(mid)[1]:
            t_hi = mid
        else:
            t_lo = mid
    b = lay(t_hi)[0]
    b[-1] = n
    return b


def tri_pairs(row_begin, row_end):
    t = lambda x: x * (x - 1) // 2 if x else 0
    return t(row_end) - t(row_begin)


def even_blocks(n, world):
    """Contiguous near-equal split of n independent units (sketch inputs, query rows)."""
    base, extra = divmod(n, world)
    b = [0]
    for g in range(world):
        b.append(b[-1] + base + (1 if g < extra else 0))
    return b
