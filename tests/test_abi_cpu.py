"""CPU-only checks of the C-ABI library: it loads and exports every symbol that
include/mashgpu.h declares; host-side helpers behave like the reference."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from mash_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.exists(abi.LIB_PATH):
        g.build()
    return abi.load_library()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "mashgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mg_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(abi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_params_init_matches_reference_rules(lib):
    p = abi.make_params(lib, k=21, s=1000)
    assert p.use64 == 1 and p.alphabet_size == 4 and p.noncanonical == 0
    assert abi.make_params(lib, k=16).use64 == 0          # 4^16 == 2^32 is not > 2^32 (Sketch.cpp:1136)
    assert abi.make_params(lib, k=17).use64 == 1
    pp = abi.make_params(lib, k=9, alphabet="ACDEFGHIKLMNPQRSTVWY", noncanonical=True)
    assert pp.alphabet_size == 20 and pp.use64 == 1
    assert abi.make_params(lib, k=7, alphabet="ACDEFGHIKLMNPQRSTVWY", noncanonical=True).use64 == 0
    low = abi.make_params(lib, k=21, alphabet="acgt")
    assert low.alphabet[ord("A")] == 1 and low.alphabet[ord("a")] == 0
    keep = abi.make_params(lib, k=21, alphabet="acgt", preserve_case=True)
    assert keep.alphabet[ord("a")] == 1 and keep.alphabet[ord("A")] == 0
    bad = abi.MgParams()
    assert lib.mg_params_init(C.byref(bad), 33, 1000, 42, b"ACGT", 0, 0) != 0
    assert lib.mg_params_init(C.byref(bad), 0, 1000, 42, b"ACGT", 0, 0) != 0


def test_distance_and_pvalue_helpers(lib, oracle, golden_dir):
    import json
    assert lib.mg_distance(1000, 1000, 21) == 0.0
    assert lib.mg_distance(0, 1000, 21) == 1.0
    assert "%g" % lib.mg_distance(456, 1000, 21) == "0.0222766"      # tutorials.rst:24
    for c in json.load(open(os.path.join(golden_dir, "binom_sf.json")))[::7]:
        # mg_p_value(x, lenRef, lenQry, kmerSpace, n): pick lengths that reproduce r
        want = c["sf"]
        got = oracle.binomial_q(c["x"] - 1, c["r"], c["n"])
        if want > 1e-290:
            assert abs(got - want) / want < 1e-9
    # golden p-values of test/ref/genomes.dist through the product arithmetic
    from tests import helpers
    gh, glens, _ = helpers.load_golden_genomes()
    _, rlen, _ = helpers.load_golden_reads()
    want = ["4.48626e-214", "2.61074e-180", "4.45454e-214"]
    for i, x in enumerate((41, 35, 41)):
        assert "%g" % lib.mg_p_value(x, int(glens[i]), rlen, 4.0 ** 21, 1000) == want[i]


def test_finish_host_matches_oracle(lib, oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_compare_vectors.npz"))
    counts = np.zeros(len(z["numer"]), dtype=abi.COUNTS_DTYPE)
    counts["numer"] = z["numer"]
    counts["denom"] = z["denom"]
    lengths = np.ascontiguousarray(z["lengths"], dtype=np.uint64)
    out = np.zeros(len(counts), dtype=abi.PAIR_DTYPE)
    rc = lib.mg_finish_tri_host(counts.ctypes.data, lengths.ctypes.data, 0, 64, int(z["k"]),
                                float(z["kmer_space"]), -1.0, -1.0, out.ctypes.data)
    assert rc == 0
    assert np.array_equal(out["distance"], z["dist"])          # same libm => bit-exact
    ref_p = z["pval"]
    nz = ref_p > 1e-290
    assert np.all(np.abs(out["p_value"][nz] - ref_p[nz]) <= 1e-9 * ref_p[nz])
    assert np.all(out["pass"] == 1)
    # filters (CommandDistance.cpp:409-424)
    out2 = np.zeros(len(counts), dtype=abi.PAIR_DTYPE)
    lib.mg_finish_tri_host(counts.ctypes.data, lengths.ctypes.data, 0, 64, int(z["k"]),
                           float(z["kmer_space"]), 0.1, 1e-10, out2.ctypes.data)
    exp_pass = (z["dist"] <= 0.1) & (z["pval"] <= 1e-10)
    assert np.array_equal(out2["pass"].astype(bool), exp_pass)


def test_finish_host_threaded_matches_serial(lib):
    """Batches above 2^20 pairs are split over host threads; the bytes must not depend on it.
    Small calls (one row block each) stay on the calling thread and serve as the reference."""
    rng = np.random.default_rng(5)
    n, s = 1600, 1000
    lengths = rng.integers(10_000, 5_000_000, n).astype(np.uint64)
    npairs = n * (n - 1) // 2
    counts = np.zeros(npairs, dtype=abi.COUNTS_DTYPE)
    counts["numer"] = rng.integers(0, s + 1, npairs)
    counts["denom"] = s
    big = np.zeros(npairs, dtype=abi.PAIR_DTYPE)
    assert lib.mg_finish_tri_host(counts.ctypes.data, lengths.ctypes.data, 0, n, 21, 4.0 ** 21, 0.3, 1e-3,
                                  big.ctypes.data) == 0
    small = np.zeros(npairs, dtype=abi.PAIR_DTYPE)
    for r0 in range(0, n, 400):
        off = r0 * (r0 - 1) // 2 if r0 else 0
        assert lib.mg_finish_tri_host(counts[off:].ctypes.data, lengths.ctypes.data, r0, min(n, r0 + 400), 21,
                                      4.0 ** 21, 0.3, 1e-3, small[off:].ctypes.data) == 0
    assert big.tobytes() == small.tobytes()
    # rectangle: 1100 x 1000 in one call vs per-query rows
    nref, nq = 1100, 1000
    c2 = counts[: nref * nq]
    rect = np.zeros(nref * nq, dtype=abi.PAIR_DTYPE)
    assert lib.mg_finish_rect_host(c2.ctypes.data, lengths[:nref].ctypes.data, nref, lengths[-nq:].ctypes.data, nq,
                                   21, 4.0 ** 21, -1.0, -1.0, rect.ctypes.data) == 0
    rows = np.zeros(nref * nq, dtype=abi.PAIR_DTYPE)
    lq = np.ascontiguousarray(lengths[-nq:])
    for q in range(0, nq, 250):
        assert lib.mg_finish_rect_host(c2[q * nref:].ctypes.data, lengths[:nref].ctypes.data, nref,
                                       lq[q:].ctypes.data, 250, 21, 4.0 ** 21, -1.0, -1.0,
                                       rows[q * nref:].ctypes.data) == 0
    assert rect.tobytes() == rows.tobytes()


def test_no_gpu_fails_loudly(lib):
    """On a box without a GPU the context cannot be created — no silent fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib.mg_ctx_create(0, C.byref(h)) != 0
    assert lib.mg_last_error(None)
    with pytest.raises(abi.MashGpuError):
        abi.MashGpu(0)


def test_shard_rows_partition_the_work_evenly(lib):
    """mg_shard_tri_rows / mg_shard_rows (the row blocks of the multi-GPU paths): blocks
    tile the row range without gaps, and every block of the triangle holds the same number of
    pairs to within one row."""
    for n, rb in ((100_000, 0), (100_000, 1), (12_345, 777), (9, 0), (2, 1), (5, 5)):
        for G in (1, 2, 3, 4, 8):
            prev, areas = rb, []
            for g in range(G):
                b, e = abi.shard_tri_rows(lib, rb, n, G, g)
                assert b == prev and b <= e <= n
                areas.append(abi.tri_pairs(b, e))
                prev = e
            assert prev == n and sum(areas) == abi.tri_pairs(rb, n)
            if n - rb >= 64 * G:
                assert max(areas) - min(areas) <= 2 * n, (n, rb, G, areas)
            prev = rb
            for g in range(G):
                b, e = C.c_uint64(), C.c_uint64()
                lib.mg_shard_rows(rb, n, G, g, C.byref(b), C.byref(e))
                assert b.value == prev and e.value >= b.value
                prev = e.value
            assert prev == max(n, rb)


def test_comm_without_gpu_fails_loudly(lib):
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 1)
    assert lib.mg_comm_create_local(devs, 2, C.byref(h)) != 0
    assert b"HIP" in lib.mg_comm_last_error(None) or b"device" in lib.mg_comm_last_error(None)


def test_weighted_shard_balances_pairs_plus_rows(lib):
    """mg_shard_tri_rows_weighted: row i costs i + w pair-units (the inverted-index engine fills per pair and discovers
    and merges per row).  Blocks tile the range, every block's cost is the same to within a row's cost, weight 0 is the
    equal-area split, and the Python mirror bench.py's dry path uses gives the same boundaries."""
    from mash_amd import shard
    for n, rb in ((100_000, 0), (100_000, 1), (12_345, 777), (9, 0), (2, 1), (5, 5), (1, 0)):
        for G in (1, 2, 3, 8):
            for w in (0.0, 1.0, 750.5, 6e4, 1e7):
                prev, costs = rb, []
                for g in range(G):
                    b, e = abi.shard_tri_rows_weighted(lib, rb, n, G, g, w)
                    assert b == prev and b <= e <= max(n, rb), (n, rb, G, g, w, b, e)
                    costs.append(abi.tri_pairs(b, e) + w * (e - b))
                    prev = e
                assert prev == max(n, rb)
                if n - rb >= 64 * G:
                    assert max(costs) - min(costs) <= 2 * (n + w), (n, rb, G, w, costs)
                if w == 0.0:
                    assert [abi.shard_tri_rows_weighted(lib, rb, n, G, g, w) for g in range(G)] == [abi.shard_tri_rows(lib, rb, n, G, g) for g in range(G)]
                elif rb == 0:
                    assert [abi.shard_tri_rows_weighted(lib, 0, n, G, g, w)[0] for g in range(G)] + [n] == shard.weighted_row_blocks(n, G, w)
    # C3 at 8 ranks with a row worth 60 000 pairs: the first block shrinks from 35 355 rows to under 20 000
    assert abi.shard_tri_rows(lib, 0, 100_000, 8, 0)[1] == 35_355 and abi.shard_tri_rows_weighted(lib, 0, 100_000, 8, 0, 6e4)[1] < 20_000


def test_costed_shard_balances_pairs_rows_and_the_prefix_below_a_block(lib):
    """mg_shard_tri_rows_costed: the block [lo, hi) costs pairs + w (hi - lo) + v hi pair-units -- a rank indexes the rows
    below its block's end only (host_compare.cpp: tri_view), so late blocks pay for more of the table.  The blocks cover the
    rows without gaps, the costs of all blocks but the last agree to within one row's cost (the last takes what is left and is
    no dearer), v = 0 is the weighted cut, and mash_amd/shard.py (what bench.py's gloo dry run uses) cuts the same rows."""
    from mash_amd import shard
    for n in (1, 7, 1000, 100_000):
        for G in (1, 2, 3, 8):
            for w, v in ((0.0, 5.9e4), (6e4, 5.9e4), (3e3, 1.0), (0.0, 1e7)):
                b = [abi.shard_tri_rows_costed(lib, 0, n, G, g, w, v) for g in range(G)]
                assert b[0][0] == 0 and b[-1][1] == n and all(b[g][1] == b[g + 1][0] for g in range(G - 1)), (n, G, w, v, b)
                cost = lambda lo, hi: shard.tri_pairs(lo, hi) + w * (hi - lo) + v * hi
                one_row = n + w + v + 1
                full = [cost(lo, hi) for lo, hi in b if lo < hi < n]
                last = [cost(lo, hi) for lo, hi in b if lo < hi == n]
                if full:
                    assert max(full) - min(full) <= one_row, (n, G, w, v, b)
                    assert last and last[0] <= max(full) + one_row, (n, G, w, v, b)
                assert [x[0] for x in b] + [n] == shard.costed_row_blocks(n, G, w, v), (n, G, w, v)
            assert [abi.shard_tri_rows_costed(lib, 0, n, G, g, 6e4, 0.0) for g in range(G)] == [abi.shard_tri_rows_weighted(lib, 0, n, G, g, 6e4) for g in range(G)]
    # C3 on 8 ranks with the measured weight: the last rank, whose view is the whole table, gets a sliver of the pairs
    b = shard.costed_row_blocks(100_000, 8, 0.0, 5.9e4)
    assert shard.tri_pairs(b[7], b[8]) < shard.tri_pairs(b[0], b[1]) / 3
