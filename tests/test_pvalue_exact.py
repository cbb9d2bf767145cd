"""p-value parity (CPU): the product's binomial tail (mash_amd/csrc/pvalue.h through the C ABI,
mg_p_value_within / mg_p_value) against EXACT values (tests/golden/binom_exact.json: 80-digit sums
rounded once, made by tests/golden/make_binom_exact.py), reported in ulp.

The reference's pValue (CommandDistance.cpp:427-448) calls gsl_cdf_binomial_Q or Boost, neither in
the reference tree nor pinned; "1 ulp of the reference" is therefore not defined.  What IS defined is
the exact tail, and the bar here is <= 1 ulp of it (0 observed) across 1658 cases: n up to 1e5,
r from 1e-12 to 0.999, tails down through the denormals to exact 0.  The same file holds scipy's
(Boost's) value for each case: the test prints its ulp distribution -- hundreds of ulp in the deep
tails -- which is the spread a GSL or Boost build of the reference itself has."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

from mash_amd import abi


def _ordint(x):
    i = struct.unpack("<q", struct.pack("<d", x))[0]
    return i if i >= 0 else -(i & 0x7FFFFFFFFFFFFFFF)


def ulps(a, b):
    return abs(_ordint(a) - _ordint(b))


@pytest.fixture(scope="module")
def lib():
    return abi.load_library()


@pytest.fixture(scope="module")
def cases(golden_dir):
    out = []
    for c in json.load(open(os.path.join(golden_dir, "binom_exact.json"))):
        out.append(dict(x=c["x"], n=c["n"], set_size=c["set_size"], kmer_space=float.fromhex(c["kmer_space"]),
                        r=float.fromhex(c["r"]), exact=float.fromhex(c["exact"]), scipy=float.fromhex(c["scipy"]),
                        log10=c["log10_exact"]))
    return out


def _dist(v):
    v = np.asarray(v)
    return {"n": int(v.size), "max": int(v.max()), "p99": float(np.percentile(v, 99)), "median": float(np.median(v)),
            "zero": int((v == 0).sum())}


def test_binomial_tail_within_one_ulp_of_exact(lib, cases):
    ours, boost, worst = [], [], None
    for c in cases:
        got = lib.mg_p_value_within(c["x"], c["set_size"], c["kmer_space"], c["n"])
        u = ulps(got, c["exact"])
        ours.append(u)
        boost.append(ulps(c["scipy"], c["exact"]))
        if worst is None or u > worst[0]:
            worst = (u, c, got)
        assert u <= 1, (c, got)
        assert "%g" % got == "%g" % c["exact"], (c, got)          # what the CLI prints
    d_ours, d_boost = _dist(ours), _dist(boost)
    print("\np-value ulp error vs the exact tail, %d cases" % len(cases))
    print("  this library : %s" % d_ours)
    print("  scipy (Boost): %s" % d_boost)
    assert d_ours["max"] <= 1
    assert d_boost["max"] > 16           # the fixture does show a production library's spread


def test_underflow_edge_denormals_and_zero(lib, cases):
    """Below 1e-290 nothing is masked: values in the denormal range agree to one denormal step and
    exact zeros are zeros -- this decides whether the CLI prints `0`."""
    edge = [c for c in cases if c["exact"] < 1e-290]
    assert sum(1 for c in edge if c["exact"] == 0.0) >= 50
    assert sum(1 for c in edge if 0.0 < c["exact"] < 2.3e-308) >= 10          # denormals are covered
    for c in edge:
        got = lib.mg_p_value_within(c["x"], c["set_size"], c["kmer_space"], c["n"])
        if c["exact"] == 0.0:
            # exactly representable zero only when the true tail is below half the smallest denormal
            assert got == 0.0, (c, got)
        else:
            assert ulps(got, c["exact"]) <= 1, (c, got)
        assert ("%g" % got == "0") == ("%g" % c["exact"] == "0")


def test_p_value_goldens_and_r_arithmetic(lib, golden_dir):
    """mg_p_value computes r from the two lengths exactly as pValue does (CommandDistance.cpp:436-441)
    and reproduces the printed p-values of test/ref/genomes.dist."""
    from tests import helpers
    _, glens, _ = helpers.load_golden_genomes()
    _, rlen, _ = helpers.load_golden_reads()
    want = ["4.48626e-214", "2.61074e-180", "4.45454e-214"]
    ks = 4.0 ** 21
    for i, x in enumerate((41, 35, 41)):
        assert "%g" % lib.mg_p_value(x, int(glens[i]), rlen, ks, 1000) == want[i]
        pX = 1.0 / (1.0 + ks / float(glens[i]))
        pY = 1.0 / (1.0 + ks / float(rlen))
        r = pX * pY / (pX + pY - pX * pY)
        # same tail through the other entry point, r handed over as a quotient that reproduces it
        assert lib.mg_p_value(0, 1, 1, ks, 1000) == 1.0
        got = lib.mg_p_value(x, int(glens[i]), rlen, ks, 1000)
        import mpmath as mp
        mp.mp.dps = 60
        rr, q = mp.mpf(r), 1 - mp.mpf(r)
        tot = mp.mpf(0)
        for j in range(x, 1001):
            tot += mp.binomial(1000, j) * rr ** j * q ** (1000 - j)
        assert ulps(got, float(tot)) <= 1


def test_oracle_binomial_is_an_independent_algorithm(oracle, cases):
    """The oracle keeps the regularized incomplete beta by Lentz's continued fraction in log space
    (oracle/mash_oracle.c) -- a different algorithm from the product's exact sum -- and agrees with
    the exact values to 1e-9 relative where a log-space evaluation can (above 1e-290)."""
    worst = 0.0
    for c in cases[::3]:
        if c["n"] > 10000:
            continue                                                  # its own accuracy falls with n (1e-10 at 1e5)
        got = oracle.binomial_q(c["x"] - 1, c["r"], c["n"])
        if c["exact"] > 1e-290:
            rel = abs(got - c["exact"]) / c["exact"]
            worst = max(worst, rel)
            assert rel < 1e-9, (c, got)
        else:
            assert got < 1e-280
    print("\noracle (Lentz) worst relative error vs exact: %.2e" % worst)


@pytest.fixture(scope="module")
def pair_cases(golden_dir):
    out = []
    for c in json.load(open(os.path.join(golden_dir, "binom_exact_pairs.json"))):
        out.append(dict(x=c["x"], n=c["n"], len_ref=c["len_ref"], len_qry=c["len_qry"], kmer_space=float.fromhex(c["kmer_space"]),
                        r=float.fromhex(c["r"]), exact=float.fromhex(c["exact"])))
    return out


def test_pair_p_value_within_one_ulp_of_exact(lib, pair_cases):
    """pValue(x, lenRef, lenQry, kmerSpace, sketchSize) as compareSketches calls it (CommandDistance.cpp:427-448):
    r from the two genome lengths, then the tail -- 4812 cases against exact values
    (tests/golden/make_binom_exact_pairs.py), the form in which the tail also runs on the device
    (tests/test_gpu_parity.py::test_device_finish_on_exact_p_values feeds the same cases to finish.hip)."""
    lib.mg_p_value.restype = C.c_double
    lib.mg_p_value.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_uint64]
    worst = 0
    for c in pair_cases:
        pX = 1.0 / (1.0 + c["kmer_space"] / float(c["len_ref"]))
        pY = 1.0 / (1.0 + c["kmer_space"] / float(c["len_qry"]))
        assert pX * pY / (pX + pY - pX * pY) == c["r"]                 # the fixture's r is this machine's r
        got = lib.mg_p_value(c["x"], c["len_ref"], c["len_qry"], c["kmer_space"], c["n"])
        u = ulps(got, c["exact"])
        worst = max(worst, u)
        assert u <= 1, (c, got)
        assert "%g" % got == "%g" % c["exact"], (c, got)
    assert sum(1 for c in pair_cases if c["exact"] == 0.0) >= 100 and sum(1 for c in pair_cases if 0.0 < c["exact"] < 2.3e-308) >= 5
    print("\npair p-values: %d cases, worst %d ulp" % (len(pair_cases), worst))
