"""CPU check of the sketch kernel's per-lane k-mer state (kmer_hash.h, host build):
the hash emitted at every byte position must equal the oracle's getHash of the
canonical k-mer ending there, and validity must match the reference's skip rule."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from workloads import synth

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def roller():
    out = os.path.join(HERE, "host", "libroller_host.so")
    src = os.path.join(HERE, "host", "roller_host.cpp")
    hdr = os.path.join(HERE, "..", "mash_amd", "csrc", "kmer_hash.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out, src], check=True)
    lib = C.CDLL(out)
    lib.roller_run.restype = C.c_int
    return lib


def _comp(b):
    return bytes({65: 84, 67: 71, 71: 67, 84: 65}[x] for x in reversed(b))


def _expected(oracle, seq, k, mode, alphabet, fold, seed, use64):
    n = len(seq)
    valid = np.zeros(n, dtype=np.uint8)
    hashes = np.zeros(n, dtype=np.uint64)
    if fold:
        up = bytes((c - 32) if 97 <= c <= 122 else c for c in seq)
    else:
        up = seq
    ok = [alphabet[c] for c in up]
    for e in range(k - 1, n):
        st = e - k + 1
        if all(ok[st:e + 1]):
            valid[e] = 1
            km = up[st:e + 1]
            if mode == 0:
                rc = _comp(km)
                if rc < km:
                    km = rc
            hashes[e] = oracle.get_hash(km, seed, use64)
    return valid, hashes


@pytest.mark.parametrize("k", [1, 2, 7, 8, 9, 15, 16, 17, 21, 23, 24, 25, 31, 32])
@pytest.mark.parametrize("mode", [0, 1])
def test_roller_dna(roller, oracle, k, mode):
    rng = np.random.default_rng(k * 10 + mode)
    seq = b"".join(synth.adversarial_dna_records(rng, 1))[:1500] + b"\n" + synth.adversarial_dna_records(rng, 2)[0][:600]
    seq += b"\n" + synth.adversarial_dna_records(rng, 4)[0][:600]
    alphabet = np.zeros(256, dtype=np.uint8)
    for ch in b"ACGT":
        alphabet[ch] = 1
    for fold in (1, 0):
        use64 = 4.0 ** k > 2.0 ** 32
        arr = np.frombuffer(seq, dtype=np.uint8)
        v = np.zeros(len(arr), dtype=np.uint8)
        h = np.zeros(len(arr), dtype=np.uint64)
        rc = roller.roller_run(k, arr.ctypes.data_as(C.c_void_p), C.c_uint64(len(arr)), mode,
                               alphabet.ctypes.data_as(C.c_void_p), fold, 42, int(use64),
                               v.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p))
        assert rc == 0
        ev, eh = _expected(oracle, seq, k, mode, alphabet, fold, 42, use64)
        assert np.array_equal(v, ev)
        assert np.array_equal(h[ev == 1], eh[ev == 1])


@pytest.mark.parametrize("k", [1, 3, 8, 9, 16, 17, 32])
def test_roller_protein(roller, oracle, k):
    rng = np.random.default_rng(100 + k)
    seq = b"\n".join(synth.random_protein_records(rng, 1) + synth.random_protein_records(rng, 2))
    alphabet = np.zeros(256, dtype=np.uint8)
    for ch in b"ACDEFGHIKLMNPQRSTVWY":
        alphabet[ch] = 1
    use64 = 20.0 ** k > 2.0 ** 32
    arr = np.frombuffer(seq, dtype=np.uint8)
    v = np.zeros(len(arr), dtype=np.uint8)
    h = np.zeros(len(arr), dtype=np.uint64)
    rc = roller.roller_run(k, arr.ctypes.data_as(C.c_void_p), C.c_uint64(len(arr)), 2,
                           alphabet.ctypes.data_as(C.c_void_p), 1, 7, int(use64),
                           v.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p))
    assert rc == 0
    ev, eh = _expected(oracle, seq, k, 2, alphabet, 1, 7, use64)
    assert np.array_equal(v, ev)
    assert np.array_equal(h[ev == 1], eh[ev == 1])
