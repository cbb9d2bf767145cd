"""Synthetic workloads for tests and bench.py (SURVEY.md §8d definitions).

* `synthetic_genome(g, L)`      — C2 genomes: i.i.d. uniform ACGT from splitmix64,
                                   state0 = 0x9E3779B97F4A7C15*(g+1), 32 bases per
                                   64-bit output, 2 bits each from the LSB up.
* `clustered_sketches(N, s, …)` — C3 sketch tables: clusters of related sketches so
                                   Jaccard values span 0 … ~0.7 (plus extremes).
* adversarial record generators — N runs, lowercase, short records, non-ASCII bytes,
                                   repeats, palindromes: the edge cases of
                                   Sketch.cpp:512-583 / SURVEY Appendix B.
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def splitmix64_stream(state0, n):
    """First n outputs of splitmix64 started at state0 (numpy uint64 array)."""
    with np.errstate(over="ignore"):
        z = np.uint64(state0) + GOLDEN * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def synthetic_genome_codes(g, length):
    """2-bit base codes (uint8 0..3) of synthetic genome g."""
    with np.errstate(over="ignore"):
        state0 = GOLDEN * np.uint64(g + 1)
    nw = (length + 31) // 32
    w = splitmix64_stream(state0, nw)
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    codes = ((w[:, None] >> shifts) & np.uint64(3)).astype(np.uint8).reshape(-1)
    return codes[:length]


def synthetic_genome(g, length):
    """ASCII bytes (numpy uint8) of synthetic genome g: uppercase ACGT, no N."""
    return _ACGT[synthetic_genome_codes(g, length)]


def robust_variant(seq, g):
    """SURVEY §8d robustness variant: 10 N-runs (1..100) + 5 lowercase 1 kb runs."""
    rng = np.random.default_rng(1000003 * (g + 1))
    out = seq.copy()
    n = len(out)
    for _ in range(10):
        ln = int(rng.integers(1, 101))
        st = int(rng.integers(0, max(1, n - ln)))
        out[st:st + ln] = ord("N")
    for _ in range(5):
        st = int(rng.integers(0, max(1, n - 1000)))
        seg = out[st:st + 1000]
        up = (seg >= 65) & (seg <= 90)
        seg[up] += 32
    return out


def clustered_sketches(n, s=1000, clusters=None, seed=0, pool=1500, private=400, keep_p=0.8,
                       length=1_000_000):
    """(table u64[n,s] ascending, padded UINT64_MAX; nhash u32[n];
    lengths u64[n]).  Cluster c owns a pool of `pool` hashes (< 2^54); sketch m takes each
    pool element with probability keep_p plus `private` hashes of its own; bottom-s kept."""
    if clusters is None:
        clusters = max(1, n // 100)
    rng = np.random.default_rng(seed)
    table = np.full((n, s), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    nhash = np.zeros(n, dtype=np.uint32)
    with np.errstate(over="ignore"):
        for m in range(n):
            c = m % clusters
            pool_h = splitmix64_stream(np.uint64(0xD1B54A32D192ED03) * np.uint64(c + 1), pool) >> np.uint64(10)
            take = rng.random(pool) < keep_p
            priv = splitmix64_stream(np.uint64(0xA0761D6478BD642F) * np.uint64(m + 1) + np.uint64(seed),
                                     private) >> np.uint64(10)
            h = np.unique(np.concatenate([pool_h[take], priv]))[:s]
            table[m, : len(h)] = h
            nhash[m] = len(h)
    lengths = np.full(n, length, dtype=np.uint64)
    return table, nhash, lengths


# --------------------------------------------------------------------------
# one species: a tree of descent (the middle of the similarity range)

_SP_C1, _SP_C2, _SP_C3 = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xC2B2AE3D27D4EB4F), np.uint64(0x165667B19E3779F9)


def _sp_mix(z):
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def species_sketches(n, s=1000, seed=0, q_inner=0.04, q_leaf=0.18, length=1_000_000):
    """A single-species collection at Mash distances of roughly 0.02 - 0.1: n leaves of a binary tree of descent; on every
    edge a share of the sketch's values is replaced by new ones (q_inner on inner edges, q_leaf on the last), so two rows
    share what neither lineage replaced since their common ancestor: ~50 % for siblings down to ~10 % across the root -- no
    near-copies, no small common pool (the values held by two rows and more number in the hundreds of thousands).  Slot k
    of a sketch holds a value of the k-th stratum of the hash range, so rows are ascending by construction.  Rows in a
    fixed pseudo-random order.  Same bits as workloads/synth_torch.species_sketch_table."""
    L = max(1, int(np.ceil(np.log2(max(n, 2)))))
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint64)[:, None]
        k = np.arange(s, dtype=np.uint64)[None, :]
        origin = np.zeros((n, s), dtype=np.uint64)
        sd = np.uint64(seed) * GOLDEN
        for lev in range(1, L + 1):
            a = i >> np.uint64(L - lev)
            u = _sp_mix((k * _SP_C1) ^ (np.uint64(lev) * _SP_C2) ^ (a * _SP_C3) ^ sd)
            thr = np.uint64(int((q_leaf if lev == L else q_inner) * 4294967296.0))
            rep = (u >> np.uint64(32)) < thr
            origin = np.where(rep, (np.uint64(lev) << np.uint64(40)) | a, origin)
        val = (k << np.uint64(44)) + (_sp_mix((k * _SP_C2) ^ (origin * _SP_C1) ^ sd) & np.uint64((1 << 44) - 1))
        order = np.argsort(_sp_mix(np.arange(n, dtype=np.uint64) * _SP_C3 ^ sd), kind="stable")
    table = np.ascontiguousarray(val[order])
    return table, np.full(n, s, dtype=np.uint32), np.full(n, length, dtype=np.uint64)


def random_sketches(n, s=1000, seed=0, bits=54):
    """All-random table (common ~ 0): the cheap extreme of the merge."""
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 1 << bits, size=(n, s + 8), dtype=np.uint64)
    t.sort(axis=1)
    table = np.full((n, s), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    nhash = np.zeros(n, dtype=np.uint32)
    for i in range(n):
        u = np.unique(t[i])[:s]
        table[i, : len(u)] = u
        nhash[i] = len(u)
    return table, nhash, np.full(n, 1_000_000, dtype=np.uint64)


# --------------------------------------------------------------------------
# adversarial inputs (edge cases the reference's hot loop handles)

def _rand_dna(rng, n):
    return _ACGT[rng.integers(0, 4, n)].tobytes()


def adversarial_dna_records(rng, variant):
    """A list of records (bytes) for one sketch."""
    if variant == 0:      # plain multi-record, one shorter than any k, one empty-ish
        return [_rand_dna(rng, 5000), b"ACG", _rand_dna(rng, 2500), _rand_dna(rng, 20), _rand_dna(rng, 33)]
    if variant == 1:      # N runs, lowercase, IUPAC codes, non-alphabet bytes
        s = bytearray(_rand_dna(rng, 8000))
        for _ in range(12):
            st = int(rng.integers(0, 7900)); ln = int(rng.integers(1, 60))
            s[st:st + ln] = b"N" * ln
        for _ in range(6):
            st = int(rng.integers(0, 7000)); ln = int(rng.integers(1, 400))
            s[st:st + ln] = bytes(s[st:st + ln]).lower()
        for pos, ch in ((100, b"R"), (101, b"y"), (900, b"-"), (901, b"*"), (2000, b"."), (4000, b"U"), (4100, b"u")):
            s[pos:pos + 1] = ch
        s[5001] = 0x01
        return [bytes(s), bytes(_rand_dna(rng, 700)).lower()]
    if variant == 2:      # heavy repeats + palindromes + homopolymers
        unit = _rand_dna(rng, 37)
        pal = b"ACGTTGCATGCAACGT" * 4
        return [unit * 120 + b"A" * 500 + pal * 10 + _rand_dna(rng, 300), b"T" * 300, (b"AT" * 200)]
    if variant == 4:
        # bytes >= 0x80: the reference indexes alphabet[] with a negative char here
        # (Sketch.cpp:550, undefined behaviour); this engine DEFINES them as invalid
        # bases.  Never part of reference-run fixtures, only oracle-vs-device tests.
        s = bytearray(_rand_dna(rng, 3000))
        for pos in (17, 500, 501, 1999, 2999):
            s[pos] = int(rng.integers(0x80, 0x100))
        return [bytes(s)]
    # variant 3: long single record
    return [_rand_dna(rng, 60000)]


_PROT = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)


def random_protein_records(rng, variant):
    n = [3000, 800, 12000, 64][variant]
    s = bytearray(_PROT[rng.integers(0, 20, n)].tobytes())
    if variant == 1:
        for pos in (10, 11, 300, 500):
            s[pos:pos + 1] = b"X"
        s[600:640] = bytes(s[600:640]).lower()
    if variant == 2:
        s[100:101] = b"*"
        return [bytes(s[:5000]), bytes(s[5000:]), b"AC"]
    return [bytes(s)]
