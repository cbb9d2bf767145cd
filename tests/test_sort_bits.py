"""How many leading bits the inverted index's radix sort looks at (mash_amd/csrc/sort_bits.h, used by
host_index.cpp::SparseIndexBuild::lay_out): the rule is plain C++, compiled here on its own with g++ and checked on tables whose
answer can be worked out by hand and against a simulation of the thing it estimates -- the number of pairs of DIFFERENT
values that share a bucket of 2^bb."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "sort_bits.h"
extern "C" unsigned sort_bits_rule(const double *dens, unsigned end_bit, unsigned even_begin)
{
    return mg::sort_begin_bit_from_density(dens, end_bit, even_begin);
}
'''


@pytest.fixture(scope="module")
def rule(tmp_path_factory):
    d = tmp_path_factory.mktemp("sort_bits")
    src, so = d / "rule.cpp", d / "rule.so"
    src.write_text(SRC)
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "mash_amd", "csrc"), str(src), "-o", str(so)], check=True)
    lib = ctypes.CDLL(str(so))
    lib.sort_bits_rule.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
    lib.sort_bits_rule.restype = ctypes.c_uint

    def call(rows, end_bit, even_begin):
        """rows: (hashes of the row, its largest hash) pairs"""
        dens = np.zeros(65, dtype=np.float64)
        for c, last in rows:
            dens[max(1, int(last).bit_length())] += c / (last + 1.0)
        return int(lib.sort_bits_rule(dens.ctypes.data, end_bit, even_begin))
    return call


def test_evenly_spread_tables_keep_the_even_rule(rule):
    # C3: 10^5 rows x 1000 values below 2^54 -> 40 bits (begin 14); C5: 10^9 values -> 48 bits (begin 6)
    assert rule([(1000, (1 << 54) - 1)] * 1000, 54, 14) in (14,)          # (1000 rows at the same density per row: fewer ties, same cap)
    assert rule([(100000 * 1000, (1 << 54) - 1)], 54, 14) == 14
    assert rule([(100000 * 10000, (1 << 54) - 1)], 54, 6) == 6
    assert rule([(5, 100)], 7, 0) == 0                                      # the even rule said "every bit": nothing to add


def test_dense_low_end_takes_more_bits_or_every_bit(rule):
    # nine rows in ten below 2^50.7, one in ten up to 2^59: 56 of 60 bits (one pass of eight saved)
    mixed = [(90000 * 1000, int(2 ** 50.7)), (10000 * 1000, (1 << 59) - 1)]
    assert rule(mixed, 60, 20) == 4
    # everything below 2^40 but one row up to 2^62: the 40 dense bits cannot be cut, no pass is saved
    assert rule([(10 ** 8, (1 << 40) - 1), (1000, (1 << 62) - 1)], 63, 23) == 0
    # never above the even-spread rule
    assert rule([(1000, (1 << 54) - 1)], 54, 14) == 14


@pytest.mark.parametrize("seed", [0, 1])
def test_the_estimate_bounds_the_ties_of_a_sampled_table(rule, seed):
    """rows of two sizes; values drawn as a row's hashes are (uniform below its largest); the number of pairs of different
    values per bucket at the chosen begin bit stays near 2^13 or below (x 4 for the rounding to whole passes is the other way:
    whole passes only ever add bits)"""
    rng = np.random.default_rng(seed)
    rows = [(1000, (1 << 40) - 1)] * 3000 + [(1000, (1 << 52) - 1)] * 1000
    vals = np.concatenate([rng.integers(0, last + 1, size=c, dtype=np.uint64) for c, last in rows])
    end_bit = 52
    bb = rule(rows, end_bit, 24)
    assert 0 < bb <= 24
    b = np.sort(vals >> np.uint64(bb))
    _, counts = np.unique(b, return_counts=True)
    pairs = int((counts.astype(np.int64) * (counts - 1) // 2).sum())
    assert pairs <= 4 * 8192
    # with the even-spread rule's begin bit the same table would have had far more
    b24 = vals >> np.uint64(24)
    _, c24 = np.unique(b24, return_counts=True)
    assert int((c24.astype(np.int64) * (c24 - 1) // 2).sum()) > 20 * max(pairs, 1)
