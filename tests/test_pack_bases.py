"""mg_pack_bases -- the host side of the packed nucleotide input (include/mashgpu.h, mash_amd/csrc/pack_bases.cpp) -- against a
numpy statement of the format: two bits per base, code = (ASCII >> 1) & 3 after the case fold of addMinHashes
(Sketch.cpp:512-523: 'a'..'z' -> upper case unless preserveCase), one invalid bit per base that is none of ACGT
(Sketch.cpp:550: a k-mer over a character outside the alphabet is skipped).  Host code only: runs without a GPU."""
import numpy as np
import pytest

from mash_amd import abi

ALPHA = np.frombuffer(b"ACGTacgtNnRYKMryUu\n-*.\x00\x01\x7f\x80\xc1\xe1\xffZzBbDdHhVv", dtype=np.uint8)


def model(bases, preserve_case):
    u = bases.copy()
    if not preserve_case:
        lower = (u > 96) & (u < 123)
        u[lower] -= 32
    valid = np.isin(u, np.frombuffer(b"ACGT", dtype=np.uint8))
    codes = ((u >> 1) & 3).astype(np.uint8)
    n = len(bases)
    pad = (-n) % 8
    inv = np.concatenate([~valid, np.zeros(pad, dtype=bool)])
    mask = np.packbits(inv, bitorder="little")
    return codes, valid, mask


@pytest.mark.parametrize("preserve_case", [False, True])
@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 1000, 4097, 65539])
def test_pack_bases_matches_the_format(n, preserve_case):
    rng = np.random.default_rng(n * 2 + int(preserve_case))
    bases = rng.choice(ALPHA, size=n).astype(np.uint8)
    packed, mask, ninv = abi.pack_bases(bases, preserve_case)
    codes, valid, want_mask = model(bases, preserve_case)
    assert len(packed) == (n + 3) // 4 and len(mask) == (n + 7) // 8
    assert np.array_equal(mask, want_mask)
    assert ninv == int((~valid).sum())
    got = (packed[np.arange(n) // 4] >> (2 * (np.arange(n) % 4)).astype(np.uint8)) & 3
    assert np.array_equal(got[valid], codes[valid])              # (the code bits of an invalid base mean nothing)
    assert {0: b"A", 1: b"C", 2: b"T", 3: b"G"} == {int((c >> 1) & 3): bytes([c]) for c in b"ACTG"}


def test_pack_bases_every_byte_value():
    bases = np.arange(256, dtype=np.uint8).repeat(3)
    for pc in (False, True):
        _, mask, ninv = abi.pack_bases(bases, pc)
        _, valid, want = model(bases, pc)
        assert np.array_equal(mask, want) and ninv == 3 * (256 - (4 if pc else 8))


def test_pack_bases_ranges_side_by_side():
    """disjoint ranges that start at multiples of 8 bases, packed by different threads into the same arrays"""
    rng = np.random.default_rng(8)
    bases = rng.choice(ALPHA, size=(1 << 21) + 13).astype(np.uint8)
    one = abi.pack_bases(bases, False)
    many = abi.pack_bases(bases, False, threads=5)
    assert np.array_equal(one[0], many[0]) and np.array_equal(one[1], many[1]) and one[2] == many[2]


def test_pack_bases_portable_and_wide_steps_agree():
    """hosts with AVX2 + BMI2 pack 32 bases per step; MASHGPU_PACK_PORTABLE=1 (read once per process) keeps the 64-bit
    steps: both processes must produce the same bytes"""
    import hashlib, os, subprocess, sys
    code = ("import numpy as np, hashlib, sys; sys.path.insert(0, %r); from mash_amd import abi;"
            "rng = np.random.default_rng(5); a = np.frombuffer(%r, dtype=np.uint8);"
            "b = rng.choice(a, size=100003).astype(np.uint8);"
            "h = hashlib.sha256();"
            "[h.update(x.tobytes()) for pc in (False, True) for x in abi.pack_bases(b, pc)[:2]];"
            "print(h.hexdigest(), abi.pack_bases(b)[2])") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ALPHA.tobytes())
    outs = []
    for portable in ("", "1"):
        env = dict(os.environ)
        env.pop("MASHGPU_PACK_PORTABLE", None)
        if portable:
            env["MASHGPU_PACK_PORTABLE"] = "1"
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip())
    assert outs[0] == outs[1] and len(outs[0].split()) == 2
