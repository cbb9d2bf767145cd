"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU
oracle on the same inputs, against the committed golden fixtures, and — at
BASELINE.json sizes — through size-independent properties.
Bar: bit-exact for hashes, hash counts, numer/denom and distances (same libm);
p-values within 1e-9 relative (the binomial tail lives in an unpinned third-party
library in the reference; golden values pin 6 digits — checked exactly as text)."""
import os

import numpy as np
import pytest

from mash_amd import abi

from workloads import synth
from tests import helpers

pytestmark = pytest.mark.gpu

KSPACE21 = 4.0 ** 21


@pytest.fixture(scope="module")
def eng():
    # torch ships its own HIP runtime: when both live in one process torch must initialise first
    # (the loader then shares one runtime); the library itself never needs torch.
    import torch
    torch.cuda.init()
    e = abi.MashGpu(0)
    # the dispatch's prices stay at their defaults: this one context serves every test of the module, and which engine takes a
    # job of borderline size must not depend on the tests that ran before (test_dispatch_costs_are_learned_per_context
    # switches the learning on for itself)
    e.set_option("MASHGPU_COSTS_FIXED", "1")
    yield e
    e.close()


def _check_sketches(eng, oracle, sketches, **kw):
    p = eng.params(**kw)
    op = oracle.params(**kw)
    hashes, nhash, counts = eng.sketch_host(sketches, p, counts=True)
    hashes2, nhash2 = eng.sketch_host(sketches, p)                 # the no-counts path gives the same sketch
    assert np.array_equal(hashes, hashes2) and np.array_equal(nhash, nhash2)
    s = kw.get("s", 1000)
    for i, recs in enumerate(sketches):
        h, c, _, _, _ = oracle.sketch_records(list(recs), op)
        assert nhash[i] == len(h), (i, kw, int(nhash[i]), len(h))
        assert np.array_equal(hashes[i, : len(h)], h), (i, kw)
        assert np.all(hashes[i, len(h):] == np.uint64(abi.HASH_PAD))
        assert hashes.shape[1] == s
        # multiplicities incl. the reference's order-dependent count of the largest kept hash
        assert np.array_equal(counts[i, : len(h)], c), (i, kw)
        assert np.all(counts[i, len(h):] == 0)


# ---------------------------------------------------------------- sketching

@pytest.mark.parametrize("k,s", [(21, 1000), (21, 50), (31, 400), (32, 128), (16, 300), (11, 64),
                                 (5, 1000), (1, 10), (17, 1000), (24, 2500), (21, 5000)])
def test_sketch_dna_canonical_vs_oracle(eng, oracle, k, s):
    rng = np.random.default_rng(1000 * k + s)
    sketches = [synth.adversarial_dna_records(rng, v) for v in (0, 1, 2, 3, 4)]
    sketches.append([b"ACGT"])                    # shorter than k (for k > 4): empty sketch
    sketches.append([b""])
    _check_sketches(eng, oracle, sketches, k=k, s=s)


@pytest.mark.parametrize("kw", [
    dict(k=21, s=200, noncanonical=True),
    dict(k=21, s=200, preserve_case=True),
    dict(k=9, s=300, alphabet="ACDEFGHIKLMNPQRSTVWY", noncanonical=True),
    dict(k=3, s=100, alphabet="ACDEFGHIKLMNPQRSTVWY", noncanonical=True),
    dict(k=7, s=1000, alphabet="ACGTN", noncanonical=True),
    dict(k=21, s=1000, seed=7),
])
def test_sketch_modes_vs_oracle(eng, oracle, kw):
    rng = np.random.default_rng(5)
    if len(kw.get("alphabet", "ACGT")) > 5:
        sketches = [synth.random_protein_records(rng, v) for v in range(4)]
    else:
        sketches = [synth.adversarial_dna_records(rng, v) for v in range(5)]
    _check_sketches(eng, oracle, sketches, **kw)


@pytest.mark.parametrize("k", range(1, 33))
def test_sketch_every_kmer_size(eng, oracle, k):
    """Every k-mer size is its own kernel instantiation (window dwords, tail bytes of the hash,
    32- vs 64-bit hashes): canonical DNA, forward-only DNA, protein and a min-copies run for each,
    on adversarial records, sketch sizes around the number of distinct k-mers."""
    rng = np.random.default_rng(500 + k)
    dna = [synth.adversarial_dna_records(rng, v) for v in (0, 2, 3)]
    s = int(rng.choice([16, 64, 300]))
    _check_sketches(eng, oracle, dna, k=k, s=s)
    _check_sketches(eng, oracle, dna[:2], k=k, s=s, noncanonical=True, preserve_case=bool(k % 2))
    prot = [synth.random_protein_records(rng, v) for v in (0, 1)]
    _check_sketches(eng, oracle, prot, k=k, s=s, alphabet="ACDEFGHIKLMNPQRSTVWY", noncanonical=True)
    reads = [r for recs in dna for r in recs] * 2                  # every k-mer at least twice
    p = eng.params(k=k, s=s, min_copies=2)
    hashes, nhash, counts = eng.sketch_host([reads], p, counts=True)
    oh, oc, _, _, _ = oracle.sketch_records(reads, oracle.params(k=k, s=s, min_copies=2))
    assert nhash[0] == len(oh) and np.array_equal(hashes[0, : len(oh)], oh) and np.array_equal(counts[0, : len(oh)], oc)


def test_sketch_reference_run_vectors(eng):
    """Device vs outputs of the reference's own objects (tests/golden/ref_sketch_vectors.npz)."""
    for cfg, recs, gh, gc in helpers.load_ref_sketch_vectors():
        p = eng.params(k=cfg["k"], s=cfg["s"], alphabet=cfg["alphabet"],
                       noncanonical=cfg["noncanonical"], preserve_case=cfg["preserve_case"])
        hashes, nhash, counts = eng.sketch_host([recs], p, counts=True)
        assert nhash[0] == len(gh), cfg
        assert np.array_equal(hashes[0, : len(gh)], gh), cfg
        assert np.array_equal(counts[0, : len(gc)], gc), cfg       # counts produced by the reference objects


def test_sketch_min_copies_reference_run_vectors(eng):
    """`mash sketch -r -m <m>` on the device == the reference's MinHashHeap with
    multiplicityMinimum = m (tests/golden/ref_sketch_vectors_m.npz): hashes and counts."""
    for cfg, recs, gh, gc in helpers.load_ref_sketch_vectors("ref_sketch_vectors_m.npz"):
        p = eng.params(k=cfg["k"], s=cfg["s"], min_copies=cfg["min_copies"])
        hashes, nhash, counts = eng.sketch_host([recs], p, counts=True)
        assert nhash[0] == len(gh), cfg
        assert np.array_equal(hashes[0, : len(gh)], gh), cfg
        assert np.array_equal(counts[0, : len(gc)], gc), cfg
        h2, n2 = eng.sketch_host([recs], p)                          # without the multiplicity pass
        assert n2[0] == len(gh) and np.array_equal(h2[0, : len(gh)], gh)


def test_sketch_min_copies_rounds_and_batches(eng, oracle, monkeypatch):
    """Several sketches in one call, many counting rounds (tiny ranges forced), table overflow
    retry, 32-bit hashes, and a read set in which no k-mer repeats."""
    rng = np.random.default_rng(21)

    def reads_of(g, cov, lo=60, hi=200):
        out, tot = [], 0
        while tot < cov * len(g):
            l = int(rng.integers(lo, hi))
            st = int(rng.integers(0, len(g) - l))
            r = g[st:st + l]
            out.append(r if rng.random() < 0.5 else _revcomp(r))
            tot += l
        return out

    sets = [reads_of(synth._rand_dna(rng, 30000), 5.0), reads_of(synth._rand_dna(rng, 8000), 2.0),
            [synth._rand_dna(rng, 50000)],                            # unique k-mers only: empty sketch
            reads_of(synth._rand_dna(rng, 12000), 9.0), [b"ACGT"]]
    for k, s, m in [(21, 400, 2), (15, 250, 3)]:
        for expect in (None, "2048"):
            if expect:
                monkeypatch.setenv("MASHGPU_MINCOPIES_EXPECT", expect)
            else:
                monkeypatch.delenv("MASHGPU_MINCOPIES_EXPECT", raising=False)
            p = eng.params(k=k, s=s, min_copies=m)
            hashes, nhash, counts = eng.sketch_host(sets, p, counts=True)
            for i, recs in enumerate(sets):
                oh, oc, _, _, _ = oracle.sketch_records(recs, oracle.params(k=k, s=s, min_copies=m))
                assert nhash[i] == len(oh), (k, s, m, i, expect)
                assert np.array_equal(hashes[i, : len(oh)], oh), (k, s, m, i, expect)
                assert np.array_equal(counts[i, : len(oh)], oc), (k, s, m, i, expect)
            assert nhash[2] == 0 and nhash[4] == 0


@pytest.mark.parametrize("k,s", [(21, 20000), (31, 60000), (14, 13000)])
def test_sketch_beyond_lds_selector(eng, oracle, k, s):
    """s > 12288: bottom-s by exact range counting in HBM instead of the LDS selector (same
    result by definition: the s smallest distinct hashes); full and short sketches, two at once."""
    rng = np.random.default_rng(s)
    seqs = [[synth._rand_dna(rng, 150000)], [synth._rand_dna(rng, 9000), b"ACGTNNNN" * 3, synth._rand_dna(rng, 4000)]]
    p = eng.params(k=k, s=s)
    hashes, nhash = eng.sketch_host(seqs, p)
    for i, recs in enumerate(seqs):
        oh = oracle.sketch_records(recs, oracle.params(k=k, s=s))[0]
        assert nhash[i] == len(oh), (k, s, i)
        assert np.array_equal(hashes[i, : len(oh)], oh), (k, s, i)
    assert nhash[0] == s and nhash[1] < s
    # the compare path takes such sketches through the generic kernel
    t = eng.table_upload(hashes, nhash, np.array([150000, 13000], np.uint64))
    got = eng.compare_tri_host(t)
    o = oracle.compare(hashes[1, : nhash[1]], hashes[0, : nhash[0]], 13000, 150000, s, k, 4.0 ** k)
    assert (int(got["numer"][0]), int(got["denom"][0])) == (o.numer, o.denom)
    t.free()


def test_sketch_target_coverage_reference_run_vectors(eng):
    """-c on the device + host replay == the reference's record loop with its own MinHashHeap
    (tests/golden/ref_sketch_vectors_c.npz): reads used, hashes, counts."""
    for cfg, recs, gh, gc in helpers.load_ref_sketch_vectors("ref_sketch_vectors_c.npz"):
        p = eng.params(k=cfg["k"], s=cfg["s"], min_copies=cfg["min_copies"], target_cov=cfg["target_cov"])
        h, c, used = eng.sketch_reads(recs, p)
        assert used == cfg["used"], cfg
        assert np.array_equal(h, gh) and np.array_equal(c, gc), cfg


@pytest.mark.parametrize("k,s,m", [(21, 200, 1), (21, 1000, 1), (16, 100, 2), (11, 64, 3), (31, 300, 1)])
def test_sketch_target_coverage_early_stop(eng, oracle, k, s, m):
    """`mash sketch -r -c <cov>`: the sequential heap decides after which read the input ends
    (Sketch.cpp:1258).  Device event stream + host replay against the oracle: same hashes, same
    counts, same number of reads used; thresholds below, inside and beyond what the input reaches,
    with and without -m."""
    rng = np.random.default_rng(k * 100 + s)
    g = synth._rand_dna(rng, 6000)
    reads = []
    for _ in range(3000):
        l = int(rng.integers(30, 150))
        st = int(rng.integers(0, 6000 - l))
        r = g[st:st + l]
        if rng.random() < 0.1:
            r = r[: l // 2] + b"N" + r[l // 2 + 1:]
        reads.append(r if rng.random() < 0.5 else _revcomp(r))
    reads.insert(5, b"ACGT")                                          # shorter than k: not counted
    for cov in (1.01, 1.7, 4.0, 11.5, 1000.0):
        p = eng.params(k=k, s=s, min_copies=m, target_cov=cov)
        gh, gc, used = eng.sketch_reads(reads, p)
        oh, oc, _, oused, omult = oracle.sketch_reads(reads, oracle.params(k=k, s=s, min_copies=m, target_cov=cov))
        assert used == oused, (k, s, m, cov, used, oused)
        assert np.array_equal(gh, oh) and np.array_equal(gc, oc), (k, s, m, cov)
        # the same through a session, a few records at a time: identical sketch and "reads used", and
        # the caller is told to stop reading with the chunk in which the coverage is reached
        for per in (1, 37, 5000):
            ch, cc, cused, fed = eng.sketch_reads_chunked(reads, p, per)
            assert cused == used and np.array_equal(ch, gh) and np.array_equal(cc, gc), (k, s, m, cov, per)
            nchunks = (len(reads) + per - 1) // per
            if used < sum(1 for r in reads if len(r) >= k) and per == 37:
                assert fed < nchunks
    long_enough = sum(1 for r in reads if len(r) >= k)
    assert used == long_enough                                        # 1000x is never reached: everything is read
    h0, c0, u0 = eng.sketch_reads(reads, eng.params(k=k, s=s, min_copies=m))       # target_cov 0: plain reads mode
    ph, pn, pc = eng.sketch_host([reads], eng.params(k=k, s=s, min_copies=m), counts=True)
    assert u0 == long_enough and np.array_equal(h0, ph[0, : pn[0]]) and np.array_equal(c0, pc[0, : pn[0]])
    # plain reads mode in CONSTANT memory (Sketch.cpp:1196-1270): the read set cut into 7 chunks
    # (and into many) through a reads session gives the one-shot call's hashes AND counts -- incl. the
    # order-dependent multiplicity of the largest kept hash under -m (MinHashHeap.cpp:96-144) -- and the oracle's
    oh, oc, _, _, _ = oracle.sketch_reads(reads, oracle.params(k=k, s=s, min_copies=m))
    assert np.array_equal(h0, oh) and np.array_equal(c0, oc)
    for per in ((len(reads) + 6) // 7, 211):
        # (MASHGPU_TEST_READS_RESET=1: the second time through a session that has seen another read set and was emptied by
        #  mg_reads_reset -- no trace of it; opt-in until that call has run on the GPU)
        first = reads[::-1][:500] if per == 211 and os.environ.get("MASHGPU_TEST_READS_RESET") else None
        ch, cc, cu, fed = eng.sketch_reads_chunked(reads, eng.params(k=k, s=s, min_copies=m), per, first=first)
        assert fed == (len(reads) + per - 1) // per and cu == long_enough, (k, s, m, per)
        assert np.array_equal(ch, h0) and np.array_equal(cc, c0), (k, s, m, per)


def test_reads_session_record_longer_than_the_event_buffer(eng, oracle):
    """`mash sketch -r` on a chromosome-level record.  While the heap is not full every k-mer is an
    event, so a record of more than 2^23 k-mers overflowed the event buffer and the call failed; records are now cut into
    pieces of k-mer positions, the stop test of -c still follows whole records.  One 9.5 Mbp record between short reads,
    with -m 2 (the kept set stays below s for long: every k-mer an event), plain -r and -c; the oracle has no limit."""
    from workloads import synth
    rng = np.random.default_rng(12)
    k, s = 21, 1000
    big = bytes(synth.synthetic_genome(3, 9_500_000))
    small = [bytes(synth.synthetic_genome(4, 30_000))[i * 150:(i + 1) * 150] for i in range(150)]
    reads = small[:50] + [big] + small[50:] + [big[4_000_000:4_600_000]]
    for kw in (dict(min_copies=1), dict(min_copies=2), dict(min_copies=1, target_cov=1.5)):
        p = eng.params(k=k, s=s, **kw)
        oh, oc, _, oused, _ = oracle.sketch_reads(reads, oracle.params(k=k, s=s, **kw))
        for per in (7, 1000):
            ch, cc, cused, _ = eng.sketch_reads_chunked(reads, p, per)
            assert np.array_equal(ch, oh) and np.array_equal(cc, oc), (kw, per)
            if "target_cov" in kw:
                assert cused == oused, (kw, per, cused, oused)


def _packed_batch(sketches, preserve_case=False):
    """records of every sketch joined as mg_sketch_host takes them, then packed (mg_pack_bases)"""
    blobs = [abi.join_records(r) for r in sketches]
    bases = np.frombuffer(b"".join(blobs), dtype=np.uint8)
    off = np.zeros(len(blobs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(b) for b in blobs], dtype=np.uint64)
    packed, mask, ninv = abi.pack_bases(bases, preserve_case)
    return bases, off, packed, mask, ninv


@pytest.mark.parametrize("kw", [
    dict(k=21, s=1000),
    dict(k=21, s=200, noncanonical=True),
    dict(k=21, s=200, preserve_case=True),
    dict(k=16, s=300),                      # 32-bit hashes
    dict(k=32, s=64, seed=11),
    dict(k=5, s=50),
])
def test_sketch_packed_input_matches_the_ascii_path(eng, oracle, kw, monkeypatch):
    """mg_sketch_host_packed (two bits per base + one invalid bit, ingest.hip) == mg_sketch_host on the same bytes == the
    oracle, hashes and multiplicities: adversarial records (N runs, IUPAC codes, lower case, records shorter than k, empty
    sketches), several records per sketch, sketch boundaries that are no multiples of four bases, the batch taken as
    one piece and in pieces of a few thousand bases (each piece crosses PCIe while the previous one is sketched)."""
    rng = np.random.default_rng(77 + kw["k"])
    sketches = [synth.adversarial_dna_records(rng, v) for v in range(5)] + [[b""], [b"ACGTN"]]
    sketches += [[bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.integers(3000, 9000))))] for _ in range(6)]
    bases, off, packed, mask, ninv = _packed_batch(sketches, kw.get("preserve_case", False))
    assert ninv > 0 and len(set(int(o) % 4 for o in off)) > 1
    p = eng.params(**kw)
    want_h, want_n, want_c = eng.sketch_host_raw(bases, off, p, counts=True)
    for piece in (None, "4000", "1"):
        if piece:
            monkeypatch.setenv("MASHGPU_PACKED_PIECE", piece)
        h, n, c = eng.sketch_host_packed_raw(packed, mask, len(bases), off, p, counts=True)
        assert np.array_equal(n, want_n), (kw, piece, n, want_n)
        assert np.array_equal(h, want_h), (kw, piece, np.argwhere(h != want_h)[:5])
        assert np.array_equal(c, want_c), (kw, piece, np.argwhere(c != want_c)[:5], c[c != want_c][:5], want_c[c != want_c][:5])
    monkeypatch.delenv("MASHGPU_PACKED_PIECE")
    for i in (0, 3, len(sketches) - 1):
        oh, oc, _, _, _ = oracle.sketch_records(sketches[i], oracle.params(**kw))
        assert want_n[i] == len(oh) and np.array_equal(h[i, : len(oh)], oh) and np.array_equal(c[i, : len(oh)], oc), (kw, i)


def test_sketch_batch_of_nothing_but_empty_sketches(eng):
    """No sketch of the batch holds a k-mer: hashes padded, counts ZERO (they used to be left as the pool handed them out --
    found by the packed path, which sketches piece by piece)."""
    import torch
    dev = torch.device("cuda", 0)
    p = eng.params(k=21, s=64)
    junk = torch.full((1 << 16,), 0x5A5A5A5A, dtype=torch.int32, device=dev)      # (what a recycled block may hold)
    del junk
    h, n, c = eng.sketch_host([[b""], [b"ACGT"], [b"NNNNNNNNNNNNNNNNNNNNNNNNNNNNNN"]], p, counts=True)
    assert not n.any() and not c.any() and np.all(h == np.uint64(abi.HASH_PAD))


@pytest.mark.parametrize("unpack", [False, True])
def test_sketch_packed_input_without_a_mask_and_from_device_memory(eng, unpack, monkeypatch):
    """A clean input (nothing but ACGT, one record per sketch) needs no mask: NULL.  mg_sketch_dev_packed takes the packed
    arrays from device memory (whole batch, offsets anywhere) and leaves the sketches there.  Both forms of the packed path:
    the sketch kernel expanding the codes itself while it stages its tiles (round 5, the default for plain sketches) and
    round 4's unpack_bases_kernel in front of the ASCII kernel (MASHGPU_PACKED_UNPACK=1; what multiplicities still take)."""
    import torch
    if unpack:
        monkeypatch.setenv("MASHGPU_PACKED_UNPACK", "1")
    rng = np.random.default_rng(3)
    lens = [int(x) for x in rng.integers(2000, 30000, size=9)]
    bases = rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), size=sum(lens)).astype(np.uint8)
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens, dtype=np.uint64)
    p = eng.params(k=21, s=400)
    want_h, want_n = eng.sketch_host_raw(bases, off, p)
    packed, mask, ninv = abi.pack_bases(bases)
    assert ninv == 0 and not mask.any()
    h, n = eng.sketch_host_packed_raw(packed, None, len(bases), off, p)
    assert np.array_equal(n, want_n) and np.array_equal(h, want_h)
    dev = torch.device("cuda", 0)
    pad = np.zeros(16, dtype=np.uint8)
    d_pk = torch.from_numpy(np.concatenate([packed, pad])).to(dev)
    dirty = bases.copy()
    dirty[::977] = ord("N")                               # now with a mask
    want_h2, want_n2 = eng.sketch_host_raw(dirty, off, p)
    packed2, mask2, ninv2 = abi.pack_bases(dirty)
    assert ninv2 == len(dirty[::977])
    d_pk2 = torch.from_numpy(np.concatenate([packed2, pad])).to(dev)
    d_mk2 = torch.from_numpy(np.concatenate([mask2, pad])).to(dev)
    d_h = torch.zeros((len(lens), 400), dtype=torch.int64, device=dev)
    d_n = torch.zeros(len(lens), dtype=torch.int32, device=dev)
    eng.sketch_dev_packed(d_pk.data_ptr(), None, len(bases), off, p, d_h.data_ptr(), d_n.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_n.cpu().numpy().astype(np.uint32), want_n) and np.array_equal(d_h.cpu().numpy().view(np.uint64), want_h)
    eng.sketch_dev_packed(d_pk2.data_ptr(), d_mk2.data_ptr(), len(bases), off, p, d_h.data_ptr(), d_n.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(d_n.cpu().numpy().astype(np.uint32), want_n2) and np.array_equal(d_h.cpu().numpy().view(np.uint64), want_h2)
    # other alphabets have no packed form
    with pytest.raises(abi.MashGpuError, match="ACGT"):
        eng.sketch_host_packed_raw(packed, None, len(bases), off, eng.params(k=9, s=100, alphabet="ACDEFGHIKLMNPQRSTVWY", noncanonical=True))


def test_sketch_fuzz_regressions(eng, oracle):
    """Inputs on which tests/fuzz_sketch.py found the engine wrong (tests/golden/sketch_fuzz_regressions.npz):
    long records over a handful of distinct k-mers (protein k = 3, k = 1 over an 8-letter alphabet).
    Nearly every k-mer keeps passing the threshold there; a segmented tile that started with a
    nearly full candidate buffer wrote its first segment past the buffer, into the staged bases."""
    import json
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sketch_fuzz_regressions.npz"))
    for cfg in json.loads(str(z["cfgs"])):
        i = cfg["idx"]
        bases, lens = z[f"bases_{i}"].tobytes(), z[f"lens_{i}"]
        recs, o = [], 0
        for l in lens:
            recs.append(bases[o:o + int(l)])
            o += int(l)
        kw = dict(k=cfg["k"], s=cfg["s"], seed=cfg["seed"], alphabet=cfg["alphabet"], noncanonical=bool(cfg["noncanonical"]),
                  preserve_case=bool(cfg["preserve_case"]))
        h, n, c = eng.sketch_host([recs, recs], eng.params(**kw), counts=True)
        oh, oc, _, _, _ = oracle.sketch_records(recs, oracle.params(**kw))
        assert np.array_equal(oh, z[f"oracle_{i}"]), cfg["tag"]
        for r in (0, 1):
            assert np.array_equal(h[r, : n[r]], oh) and np.array_equal(c[r, : n[r]], oc), cfg["tag"]


@pytest.mark.parametrize("alphabet,k,s,length", [("ACDEFGHIKLMNPQRSTVWY", 3, 400, 37000), ("ACDEFGHIKLMNPQRSTVWY", 2, 300, 61000),
                                                 ("ACGTacgt", 1, 2, 39000), ("AC", 7, 100, 45000), ("ACGT", 3, 40, 70000),
                                                 ("ACGT", 21, 1000, 50000)])
def test_sketch_few_distinct_kmers(eng, oracle, alphabet, k, s, length):
    """Long inputs with few distinct k-mers (tiny k-mer spaces; for k = 21 a short unit repeated):
    duplicates of kept hashes pass the threshold all the time, the candidate buffer fills within a
    fraction of a tile, every tile goes through overflow handling.  Several sketches per batch,
    forward and canonical, against the oracle, with multiplicities."""
    rng = np.random.default_rng(k * 1000 + s)
    letters = np.frombuffer(alphabet.upper().encode(), dtype=np.uint8)
    sketches = []
    for v in range(4):
        if k == 21:
            unit = letters[rng.integers(0, len(letters), int(rng.integers(25, 400)))].tobytes()
            seq = (unit * (length // len(unit) + 1))[:length]
        else:
            seq = letters[rng.integers(0, len(letters), length + 1000 * v)].tobytes()
        sketches.append([seq])
    for noncanon in ([True] if alphabet != "ACGT" else [False, True]):
        kw = dict(k=k, s=s, alphabet=alphabet, noncanonical=noncanon)
        h, n, c = eng.sketch_host(sketches, eng.params(**kw), counts=True)
        for i, recs in enumerate(sketches):
            oh, oc, _, _, _ = oracle.sketch_records(recs, oracle.params(**kw))
            assert np.array_equal(h[i, : n[i]], oh), (alphabet, k, s, noncanon, i)
            assert np.array_equal(c[i, : n[i]], oc), (alphabet, k, s, noncanon, i)


def test_sketch_bloom_reference_run_vectors(eng):
    """-b on the device + host replay == the reference's record loop with its own MinHashHeap and
    the vendored Bloom filter (tests/golden/ref_sketch_vectors_b.npz): hashes, counts, reads used;
    in one piece and through a session a few records at a time."""
    for cfg, recs, gh, gc in helpers.load_ref_sketch_vectors("ref_sketch_vectors_b.npz"):
        p = eng.params(k=cfg["k"], s=cfg["s"], target_cov=cfg["target_cov"], bloom_bytes=cfg["bloom_bytes"])
        h, c, used = eng.sketch_reads(recs, p)
        assert used == cfg["used"], cfg
        assert np.array_equal(h, gh) and np.array_equal(c, gc), cfg
        for per in (1, 53, 100000):
            ch, cc, cused, _ = eng.sketch_reads_chunked(recs, p, per)
            assert cused == used and np.array_equal(ch, gh) and np.array_equal(cc, gc), (cfg, per)


@pytest.mark.parametrize("k,s,bloom", [(21, 200, 64), (21, 1000, 20000), (16, 100, 700), (11, 64, 5), (31, 300, 1 << 22), (21, 50, 1)])
def test_sketch_bloom_filter(eng, oracle, k, s, bloom):
    """`mash sketch -b <bytes>` (MinHashHeap.cpp:78-94): a hash is kept, with count 2, when the bit it
    maps to is already set -- by itself or by an alias -- so the sketch depends on the ORDER of the
    k-mers.  Device event stream + host replay against the oracle, filters from 8 bits (everything
    aliases) to 32 Mbit (nothing does), with noisy reads (singletons) so that aliases matter; with
    and without -c; shuffling the reads changes the result the same way on both sides."""
    rng = np.random.default_rng(k * 1000 + s + bloom)
    g = synth._rand_dna(rng, 8000)
    reads = []
    for _ in range(2500):
        l = int(rng.integers(30, 150))
        st = int(rng.integers(0, 8000 - l))
        r = bytearray(g[st:st + l])
        for _ in range(int(rng.integers(0, 3))):                      # substitution errors: k-mers seen once
            r[int(rng.integers(0, l))] = b"ACGT"[int(rng.integers(0, 4))]
        r = bytes(r)
        reads.append(r if rng.random() < 0.5 else _revcomp(r))
    reads.insert(7, b"ACG")
    outcomes = []
    for order in (reads, reads[::-1]):
        for cov in (0.0, 2.2):
            p = eng.params(k=k, s=s, target_cov=cov, bloom_bytes=bloom)
            gh, gc, used = eng.sketch_reads(order, p)
            oh, oc, _, oused, _ = oracle.sketch_reads(order, oracle.params(k=k, s=s, target_cov=cov, bloom_bytes=bloom))
            assert used == oused, (k, s, bloom, cov)
            assert np.array_equal(gh, oh) and np.array_equal(gc, oc), (k, s, bloom, cov)
            assert len(gc) == 0 or gc.min() >= 2
            ch, cc, cused, _ = eng.sketch_reads_chunked(order, p, 41)
            assert cused == used and np.array_equal(ch, gh) and np.array_equal(cc, gc), (k, s, bloom, cov)
            if cov == 0.0:
                outcomes.append(gh)
    if bloom in (64, 700):
        assert not np.array_equal(outcomes[0], outcomes[1])           # aliases: the order matters


def test_sketch_bloom_refusals(eng):
    """-b is a property of the sequential heap: the batch entry points refuse it, and it excludes
    -m as in the reference (sketchParameterSetup.cpp:44-48)."""
    p = eng.params(k=21, s=100, bloom_bytes=1000)
    with pytest.raises(abi.MashGpuError, match="bloom_bytes needs mg_sketch_reads_host"):
        eng.sketch_host([[b"ACGT" * 20]], p)
    with pytest.raises(abi.MashGpuError, match="min_copies cannot be used with bloom_bytes"):
        eng.sketch_reads([b"ACGT" * 20], eng.params(k=21, s=100, bloom_bytes=1000, min_copies=2))


@pytest.mark.parametrize("stage", [None, "4096", "100"])
def test_streamed_ingest_equals_one_batch(eng, oracle, stage, monkeypatch):
    """mg_sketch_begin / add / end_sketch / finish (pinned staging ring, copies on a copy stream) ==
    mg_sketch_host on the concatenation: pieces of any size, sketches spanning many staging buffers,
    empty sketches, two batches through one session, multiplicities."""
    if stage:
        monkeypatch.setenv("MASHGPU_STAGE_BYTES", stage)
    rng = np.random.default_rng(12)
    sketches = [synth.adversarial_dna_records(rng, v) for v in (0, 1, 2, 3, 4)]
    sketches.insert(2, [b""])
    sketches.append([b"ACGT"])
    sketches.append([bytes(synth.synthetic_genome(7, 300_000))])
    p = eng.params(k=21, s=400)
    want = eng.sketch_host(sketches, p, counts=True)
    for piece in (None, 7, 1000):
        for windows in (False, True):       # mg_sketch_add copies / the caller fills windows lent by mg_sketch_stage
            got = eng.sketch_stream(sketches, p, counts=True, piece=piece, windows=windows)
            for a, b in zip(got, want):
                assert np.array_equal(a, b), (stage, piece, windows)
    with pytest.raises(abi.MashGpuError, match="larger than the staging buffer"):
        win, ss = abi.C.c_void_p(), abi.C.c_void_p()
        eng._check(eng.lib.mg_sketch_begin(eng.ctx, abi.C.byref(p), abi.C.byref(ss)))
        try:
            eng._check(eng.lib.mg_sketch_stage(ss, eng.lib.mg_sketch_stage_capacity(ss) + 1, abi.C.byref(win)))
        finally:
            eng.lib.mg_sketch_session_free(ss)
    h, n = eng.sketch_stream([[b""], [b"AC"]], p)                       # nothing but empty sketches
    assert h.shape == (2, 400) and not n.any() and np.all(h == np.uint64(abi.HASH_PAD))


def test_sketch_reads_json_golden(eng, golden_dir):
    """mash sketch -r reads1.fastq reads2.fastq == test/ref/reads.json (hashes)."""
    r1 = helpers.read_fastx(os.path.join(golden_dir, "reads1.fastq.gz"))
    r2 = helpers.read_fastx(os.path.join(golden_dir, "reads2.fastq.gz"))
    recs = [r[2] for r in helpers.round_robin([r1, r2])]
    hashes, nhash = eng.sketch_host([recs], eng.params(k=21, s=1000))
    gh, glen, _ = helpers.load_golden_reads()
    assert nhash[0] == 1000 and np.array_equal(hashes[0], gh)
    # reads-mode length = estimateSetSize = 2^64 * n / max hash (MinHashHeap.h:45)
    assert int(2.0 ** 64 * 1000 / float(hashes[0, 999])) == glen


def test_sketch_multichunk_and_overflow(eng, oracle, monkeypatch):
    """Force many chunks per sketch (shared threshold + merge kernel) and feed
    sequences that flood the candidate buffer (homopolymers, tandem repeats)."""
    monkeypatch.setenv("MASHGPU_SKETCH_MIN_CHUNK", "15360")
    monkeypatch.setenv("MASHGPU_SKETCH_ITEMS", "100000")
    rng = np.random.default_rng(11)
    big = synth._rand_dna(rng, 300_000)
    unit = synth._rand_dna(rng, 23)
    flood = b"A" * 70_000 + unit * 4000 + synth._rand_dna(rng, 50_000) + b"C" * 30_000
    sketches = [[big], [flood], [big[:100_000], flood[:90_000], big[100_000:180_000]], [b"ACGTACGTAC" * 9000]]
    _check_sketches(eng, oracle, sketches, k=21, s=1000)
    _check_sketches(eng, oracle, sketches, k=21, s=3000)
    _check_sketches(eng, oracle, sketches[:2], k=16, s=100)


@pytest.mark.parametrize("chunks", ["one", "many"])
def test_sketch_seeded_threshold_is_never_trusted(eng, oracle, chunks, monkeypatch):
    """Sketches start from a threshold guessed from their length (3 s/L of the hash range,
    sketch_dev_impl); inputs with far fewer distinct k-mers than positions (tandem repeats, mostly
    invalid bytes, 32-bit hashes of a small k) end up short below the guess and must be re-run
    unseeded -- next to ordinary genomes in the same call, single- and multi-chunk, with counts."""
    if chunks == "many":
        monkeypatch.setenv("MASHGPU_SKETCH_MIN_CHUNK", "15360")
        monkeypatch.setenv("MASHGPU_SKETCH_ITEMS", "100000")
    rng = np.random.default_rng(77)
    unit_small = synth._rand_dna(rng, 700)           # < s distinct k-mers
    unit_mid = synth._rand_dna(rng, 2500)            # > s distinct, but almost none below the seed
    normal = synth._rand_dna(rng, 400_000)
    mostly_n = (b"N" * 900 + synth._rand_dna(rng, 100)) * 400
    sketches = [[normal], [unit_small * 400], [unit_mid * 120], [mostly_n], [normal[:200_000], unit_mid * 60],
                [unit_mid * 120 + normal[:50_000]]]
    _check_sketches(eng, oracle, sketches, k=21, s=1000)
    _check_sketches(eng, oracle, sketches, k=16, s=400)
    _check_sketches(eng, oracle, sketches, k=9, s=1000)          # k <= 16: 32-bit hashes, 4^9 = 262144 possible k-mers
    # the unseeded run is the same sketch
    monkeypatch.setenv("MASHGPU_SKETCH_NO_SEED", "1")
    p = eng.params(k=21, s=1000)
    plain = eng.sketch_host(sketches, p)
    monkeypatch.delenv("MASHGPU_SKETCH_NO_SEED")
    seeded = eng.sketch_host(sketches, p)
    assert np.array_equal(plain[0], seeded[0]) and np.array_equal(plain[1], seeded[1])


def test_sketch_c2_genomes_full_size(eng, oracle):
    """BASELINE config 2 shape: 1 Mbp synthetic genomes, k=21 s=1000 (a batch of 12,
    incl. the robustness variant) vs the oracle; plus concatenation property:
    sketch(records of A and B) == bottom-s of union(sketch(A), sketch(B))."""
    genomes = [synth.synthetic_genome(g, 1_000_000) for g in range(12)]
    genomes[3] = synth.robust_variant(genomes[3], 3)
    genomes[7] = synth.robust_variant(genomes[7], 7)
    sketches = [[bytes(g)] for g in genomes]
    _check_sketches(eng, oracle, sketches, k=21, s=1000)
    p = eng.params(k=21, s=1000)
    hs, nh = eng.sketch_host(sketches[:2] + [[bytes(genomes[0]), bytes(genomes[1])]], p)
    union = np.unique(np.concatenate([hs[0], hs[1]]))[:1000]
    assert np.array_equal(hs[2], union)


def test_config5_k31_s10000_sketch_and_triangle(eng, oracle):
    """BASELINE.json configs[4] exactly: k = 31 (64-bit hashes, 15-byte murmur tail), s = 10 000 --
    the NT = 1024 selector of sketch_chunks_kernel<31, 0, 1024, *> with its multi-pass bitonic
    merge, then the value-window compare path at s = 10 000 -- on three adversarial inputs, two
    synthetic 1 Mbp genomes and three relatives of them (so that pairs share thousands of hashes),
    against the oracle: hashes, multiplicities, numer/denom, distances bit for bit."""
    rng = np.random.default_rng(31_10000)
    g0 = synth.synthetic_genome(0, 1_000_000)
    g1 = synth.synthetic_genome(1, 1_000_000)

    def mutated(g, rate):
        a = g.copy()
        idx = np.flatnonzero(rng.random(len(a)) < rate)
        a[idx] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, len(idx))]
        return a

    sketches = [synth.adversarial_dna_records(rng, v) for v in (0, 2, 3)]
    sketches += [[bytes(g0)], [bytes(g1)], [bytes(mutated(g0, 0.01))], [bytes(mutated(g0, 0.08))],
                 [bytes(synth.robust_variant(g1, 1))], [bytes(g1[:400_000]), bytes(g0[:300_000])]]
    kw = dict(k=31, s=10000)
    p, op = eng.params(**kw), oracle.params(**kw)
    assert p.use64 == 1
    hashes, nhash, counts = eng.sketch_host(sketches, p, counts=True)
    lengths = np.zeros(len(sketches), dtype=np.uint64)
    for i, recs in enumerate(sketches):
        h, c, _, _, _ = oracle.sketch_records(list(recs), op)
        assert nhash[i] == len(h) and np.array_equal(hashes[i, : len(h)], h), i
        assert np.array_equal(counts[i, : len(h)], c), i
        lengths[i] = sum(len(r) for r in recs)
    assert int(nhash[3]) == 10000 and int(nhash[4]) == 10000
    t = eng.table_upload(hashes, nhash, lengths)
    got = eng.compare_tri_host(t)
    n = len(sketches)
    numer, denom, dist, pval = oracle.triangle(hashes, nhash, lengths, 0, n, 31, 4.0 ** 31, stats=True)
    assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom)
    assert int(got["numer"].max()) > 3000                              # the relatives really share hashes
    fin = eng.finish_tri(got, lengths, 0, n, 31, 4.0 ** 31)
    assert np.array_equal(fin["distance"], dist)
    big = pval > 1e-290
    assert np.all(np.abs(fin["p_value"][big] - pval[big]) <= 1e-9 * pval[big]) and np.all(fin["p_value"][~big] <= 1e-280)
    t.free()


# ---------------------------------------------------------------- comparing

def _set_kernel(monkeypatch, kernel):
    """Select the compare engine.  "windows": the merged kernel in value-window mode (the
    large-sketch path) forced on whatever the sketch size, with a small window target so that even
    short sketches are cut into many windows and pairs are carried from launch to launch."""
    if kernel == "plain":                      # merged kernel, whole rows in the tile table (no value windows)
        monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "merged")
        monkeypatch.setenv("MASHGPU_COMPARE_WINDOWS", "0")
    elif kernel.startswith("windows"):
        monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "merged")
        monkeypatch.setenv("MASHGPU_COMPARE_WINDOWS", "1")
        if kernel != "windows":
            monkeypatch.setenv("MASHGPU_COMPARE_WIN_TARGET", kernel[len("windows"):])
    else:
        monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", kernel)


def _oracle_tri(oracle, table, nhash, lengths, rb, re, k=21, kspace=KSPACE21):
    numer, denom, _, _ = oracle.triangle(table, nhash, lengths, rb, re, k, kspace)
    return numer, denom


@pytest.mark.parametrize("kernel", ["merged", "sparse", "join", "plain", "generic", "windows150"])
def test_compare_reference_run_vectors(eng, golden_dir, kernel, monkeypatch):
    _set_kernel(monkeypatch, kernel)
    z = np.load(os.path.join(golden_dir, "ref_compare_vectors.npz"))
    t = eng.table_upload(z["table"], z["nhash"], z["lengths"])
    got = eng.compare_tri_host(t)
    assert np.array_equal(got["numer"], z["numer"])
    assert np.array_equal(got["denom"], z["denom"])
    fin = eng.finish_tri(got, z["lengths"], 0, 64, int(z["k"]), float(z["kmer_space"]))
    assert np.array_equal(fin["distance"], z["dist"])
    # (vectors made through the oracle's log-space tail: 1e-9 relative is ITS accuracy; the product's
    # own bar -- 1 ulp of the exact tail, denormals and zeros included -- is tests/test_pvalue_exact.py)
    nz = z["pval"] > 1e-290
    assert np.all(np.abs(fin["p_value"][nz] - z["pval"][nz]) <= 1e-9 * z["pval"][nz])
    assert np.all(fin["p_value"][~nz] <= 1e-280)
    t.free()


@pytest.mark.parametrize("kernel", ["merged", "sparse", "join", "plain", "generic", "windows", "windows150", "windows7"])
def test_compare_large_reference_run_vectors(eng, golden_dir, kernel, monkeypatch):
    """Counts produced by the reference's own objects at s = 3000: the default path there is the
    value-window mode; forced window sizes, the merge-path and the generic kernel must agree."""
    _set_kernel(monkeypatch, kernel)
    z = np.load(os.path.join(golden_dir, "ref_compare_vectors_large.npz"))
    t = eng.table_upload(z["table"], z["nhash"], z["lengths"])
    got = eng.compare_tri_host(t)
    assert np.array_equal(got["numer"], z["numer"])
    assert np.array_equal(got["denom"], z["denom"])
    fin = eng.finish_tri(got, z["lengths"], 0, 16, int(z["k"]), float(z["kmer_space"]))
    assert np.array_equal(fin["distance"], z["dist"])
    # (vectors made through the oracle's log-space tail: 1e-9 relative is ITS accuracy; the product's
    # own bar -- 1 ulp of the exact tail, denormals and zeros included -- is tests/test_pvalue_exact.py)
    nz = z["pval"] > 1e-290
    assert np.all(np.abs(fin["p_value"][nz] - z["pval"][nz]) <= 1e-9 * z["pval"][nz])
    assert np.all(fin["p_value"][~nz] <= 1e-280)
    t.free()


@pytest.mark.parametrize("kernel", ["merged", "sparse", "join", "plain", "windows29"])
@pytest.mark.parametrize("s", [1, 7, 64, 65, 100, 400, 1000, 1024])
def test_compare_tiled_vs_oracle_sizes(eng, oracle, s, kernel, monkeypatch):
    _set_kernel(monkeypatch, kernel)
    n = 150
    table, nhash, lengths = synth.clustered_sketches(n, s, clusters=5, seed=s, pool=int(1.5 * s) + 2,
                                                     private=max(1, int(0.4 * s)))
    # ragged / degenerate rows
    nhash[2] = 0
    nhash[5] = min(1, s)
    nhash[9] = max(0, s - 1)
    table[17] = table[16]
    nhash[17] = nhash[16]
    t = eng.table_upload(table, nhash, lengths)
    got = eng.compare_tri_host(t)
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n)
    assert np.array_equal(got["numer"], numer)
    assert np.array_equal(got["denom"], denom)
    # sub-range of rows
    got2 = eng.compare_tri_host(t, 37, 101)
    n2, d2 = _oracle_tri(oracle, table, nhash, lengths, 37, 101)
    assert np.array_equal(got2["numer"], n2) and np.array_equal(got2["denom"], d2)
    t.free()


@pytest.mark.parametrize("kernel", ["merged", "sparse", "join", "plain", "generic", "windows", "windows150"])
@pytest.mark.parametrize("s", [1500, 4096, 10000])
def test_compare_large_sketch(eng, oracle, s, kernel, monkeypatch):
    """Config-5 sized sketches (s = 10000): merged-rows kernel with few rows per tile, and the
    generic binary-search kernel as an independent cross-check."""
    _set_kernel(monkeypatch, kernel)
    n = 24
    table, nhash, lengths = synth.clustered_sketches(n, s, clusters=3, seed=3, pool=int(1.5 * s),
                                                     private=int(0.4 * s))
    nhash[4] = s // 3
    t = eng.table_upload(table, nhash, lengths)
    windows = kernel.startswith("windows")
    if windows:
        eng.prof_enable(True)
        eng.prof_reset()
    got = eng.compare_tri_host(t)
    if windows:                                   # the window path really ran: one launch per window
        launches = eng.prof_avg_ms("compare")[1]
        eng.prof_enable(False)
        assert launches >= (2 if kernel == "windows" else s // 150 - 2), launches
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n, k=31, kspace=4.0 ** 31)
    assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom)
    # a few queries against the table (rect), and a row range
    tq = eng.table_upload(table[5:8], nhash[5:8], lengths[5:8])
    rect = eng.compare_rect_host(t, tq)
    for q in range(3):
        for r in range(n):
            i, j = max(q + 5, r), min(q + 5, r)
            if i != j:
                idx = i * (i - 1) // 2 + j
                assert (rect["numer"][q, r], rect["denom"][q, r]) == (numer[idx], denom[idx]), (q, r)
    got2 = eng.compare_tri_host(t, 9, 20)
    n2, d2 = _oracle_tri(oracle, table, nhash, lengths, 9, 20, k=31, kspace=4.0 ** 31)
    assert np.array_equal(got2["numer"], n2) and np.array_equal(got2["denom"], d2)
    t.free(); tq.free()


@pytest.mark.parametrize("s", [20000, 50000])
def test_compare_sketches_beyond_plain_tiles(eng, oracle, s, monkeypatch):
    """s > 16 384 does not fit the plain tile kernel at all; the value-window mode still applies
    (its tiles hold one window's hashes whatever s is), including classes that need a single
    window (a tiny sketch, an empty one).  Checked against the oracle and the generic kernel."""
    n = 14
    table, nhash, lengths = synth.clustered_sketches(n, s, clusters=2, seed=s, pool=int(1.5 * s), private=int(0.4 * s))
    nhash[3] = s // 3
    nhash[5] = 50                                   # a class of its own: one window
    table[5, 50:] = np.uint64(abi.HASH_PAD)
    nhash[9] = 0                                    # empty sketch
    table[9, :] = np.uint64(abi.HASH_PAD)
    table[11] = table[2]; nhash[11] = nhash[2]      # identical pair
    t = eng.table_upload(table, nhash, lengths)
    eng.prof_enable(True)
    eng.prof_reset()
    got = eng.compare_tri_host(t)
    launches = eng.prof_avg_ms("compare")[1]
    eng.prof_enable(False)
    assert launches >= s // 1000, launches          # one launch per window: the window path ran
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n, k=31, kspace=4.0 ** 31)
    assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom)
    monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "generic")
    gen = eng.compare_tri_host(t)
    assert np.array_equal(gen["numer"], numer) and np.array_equal(gen["denom"], denom)
    monkeypatch.delenv("MASHGPU_COMPARE_KERNEL")
    tq = eng.table_upload(table[4:7], nhash[4:7], lengths[4:7])
    rect = eng.compare_rect_host(t, tq)
    for q in range(3):
        for r in range(n):
            i, j = max(q + 4, r), min(q + 4, r)
            if i != j:
                idx = i * (i - 1) // 2 + j
                assert (rect["numer"][q, r], rect["denom"][q, r]) == (numer[idx], denom[idx]), (q, r)
    t.free(); tq.free()


def test_compare_extremes_and_random(eng, oracle):
    """All-random (common ~ 0) and all-identical (common = s) bracket the merge."""
    table, nhash, lengths = synth.random_sketches(200, 1000, seed=9)
    table[150:] = table[0]
    nhash[150:] = nhash[0]
    t = eng.table_upload(table, nhash, lengths)
    got = eng.compare_tri_host(t)
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, 200)
    assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom)
    assert got["numer"].max() == 1000
    t.free()


def test_compare_runs_that_name_the_same_rows(eng, oracle, monkeypatch):
    """Clades: hundreds of values held by exactly the same sketches, i.e. runs of the index that are copies of each
    other.  Every pair that is linked by ANY value must be found -- here rows of two clades with a common core each
    (copied runs), values held by all rows but one (runs that differ from the core's in a single row), and bridge
    values between the clades that are the ONLY link of their pairs.  Inverted-index engine == oracle."""
    rng = np.random.default_rng(77)
    n, s = 90, 96
    vals = np.sort(rng.choice(np.arange(1, 10 ** 6, dtype=np.uint64), 4000, replace=False))
    core = [vals[:40], vals[40:80]]
    table = np.full((n, s), np.uint64(abi.HASH_PAD), dtype=np.uint64)
    nhash = np.zeros(n, dtype=np.uint32)
    rows = []
    nxt = 80
    for i in range(n):
        c = i % 2
        own = set(int(x) for x in core[c])
        if i % 7 == 0:                                   # a member that lacks one core value: that value's run differs in one row
            own.discard(int(core[c][i % 40]))
        for _ in range(int(rng.integers(5, 30))):        # private values
            own.add(int(vals[nxt])); nxt += 1
        rows.append(own)
    for b in range(25):                                  # bridges: one value, two rows of different clades (or any two rows)
        x, y = int(rng.integers(0, n)), int(rng.integers(0, n))
        v = int(vals[nxt]); nxt += 1
        rows[x].add(v); rows[y].add(v)
    for i, own in enumerate(rows):
        r = np.array(sorted(own), dtype=np.uint64)[:s]
        table[i, : len(r)] = r
        nhash[i] = len(r)
    lengths = np.full(n, 10 ** 6, dtype=np.uint64)
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n)
    monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "sparse")
    t = eng.table_upload(table, nhash, lengths)
    got = eng.compare_tri_host(t)
    assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom)
    t.free()


def _clade_table(rng, sizes, s, keep=0.96, private=0.03, clump=False, short_every=0, gap_rows=3):
    """consecutive clades of the given sizes (near-copies of a pool of 1.06 s values), separated by `gap_rows` unrelated
    rows; optionally rows with many private values clumped between two pool values, and short rows"""
    rows = []
    for ci, m in enumerate(sizes):
        pool = np.unique(rng.integers(1, 1 << 60, size=int(1.06 * s) + 8).astype(np.uint64))
        for i in range(m):
            own = pool[rng.random(len(pool)) < keep]
            npriv = max(1, int(private * s))
            if clump and i % 4 == 0:
                lo, hi = int(pool[len(pool) // 3]), int(pool[len(pool) // 3 + 1])
                priv = rng.integers(lo + 1, max(lo + 2, hi), size=5 * npriv).astype(np.uint64)
            else:
                priv = rng.integers(1, 1 << 60, size=npriv).astype(np.uint64)
            r = np.unique(np.concatenate([own, priv]))
            k = s if not (short_every and i % short_every == 1) else int(rng.integers(s // 3, s))
            rows.append(r[:k])
        for _ in range(gap_rows):
            rows.append(np.unique(rng.integers(1, 1 << 60, size=s + 8).astype(np.uint64))[:s])
    n = len(rows)
    table = np.full((n, s), np.uint64(abi.HASH_PAD), dtype=np.uint64)
    nhash = np.zeros(n, dtype=np.uint32)
    for i, r in enumerate(rows):
        table[i, : len(r)] = r
        nhash[i] = len(r)
    return table, nhash


@pytest.mark.parametrize("s,sizes,kw", [
    (1000, (40, 9, 130), dict()),                                   # what the engine is for: clades of near-copies
    (1000, (300,), dict(keep=1.0, private=0.0)),                    # exact pool copies next to each other (the copy classes take them: no groups)
    (128, (70, 70), dict(keep=0.95, private=0.0)),                  # universe about two words, no extras
    (64, (200,), dict(keep=0.9, private=0.1, short_every=3)),       # one word, short rows
    (100, (33, 150, 8), dict(clump=True, short_every=5)),           # many extras inside one gap of the universe
    (1000, (260,), dict(keep=0.6, private=0.2)),                    # loosely related: a universe larger than s
    (3000, (140,), dict()),                                         # a universe of 50 words: the column block is streamed, not staged
    (1000, (60, 45, 90, 30), dict(shuffle=True)),                   # the same clades in RANDOM row order: the index is built on a clustered copy
    (100, (33, 150, 8), dict(clump=True, short_every=5, shuffle=True)),
    (4000, (70, 40), dict(shuffle=True, keep=0.8, private=0.2)),    # interleaved AND a universe of ~80 words
])
def test_compare_dense_groups(eng, oracle, s, sizes, kw, monkeypatch):
    """Near-identical rows (compare_dense.hip): their inner pairs as bit-mask arithmetic, everything else through the
    inverted index with clipped runs == oracle, whole triangle (the index then works on a copy of the table with related
    rows next to each other, whatever their order) and row ranges that cut through a group (consecutive rows only); the
    same bytes with the dense groups switched off."""
    rng = np.random.default_rng(s + sum(sizes))
    kw = dict(kw)
    shuffle = kw.pop("shuffle", False)
    table, nhash = _clade_table(rng, sizes, s, **kw)
    if shuffle:
        perm = rng.permutation(len(nhash))
        table, nhash = table[perm].copy(), nhash[perm].copy()
    n = len(nhash)
    lengths = np.full(n, 10 ** 6, dtype=np.uint64)
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n)
    monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "sparse")
    t = eng.table_upload(table, nhash, lengths)
    got = eng.compare_tri_host(t)
    assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom)
    for rb, re in ((n // 3, n - 5), (1, 2), (sizes[0] // 2, sizes[0] // 2 + 40)):
        part = eng.compare_tri_host(t, rb, re)
        lo, hi = rb * (rb - 1) // 2, re * (re - 1) // 2
        assert np.array_equal(part["numer"], numer[lo:hi]) and np.array_equal(part["denom"], denom[lo:hi]), (rb, re)
    t.free()
    monkeypatch.setenv("MASHGPU_COMPARE_DENSE", "0")
    t = eng.table_upload(table, nhash, lengths)
    plain = eng.compare_tri_host(t)
    assert plain.tobytes() == got.tobytes()
    t.free()


@pytest.mark.parametrize("bits", [9, 20, 26, 34])
@pytest.mark.parametrize("s,sizes,kw", [
    (1000, (40, 9, 130), dict()),
    (100, (33, 150, 8), dict(clump=True, short_every=5)),           # values packed into one gap: long segments of one prefix
    (200, (60, 45), dict(shuffle=True, gap_rows=400)),              # mostly unrelated rows
])
def test_compare_sparse_index_sorted_on_leading_bits(eng, oracle, bits, s, sizes, kw, monkeypatch):
    """The index sorts the values on their leading bits only and repairs the segments in which different values agree
    in all of them (sp_tie_find_kernel / sp_tie_repair_kernel).  Forced down to a few bits: many ties, long segments,
    segments of more than 64 values (the build is then repeated on every bit) -- the same bytes as the oracle and as the
    index sorted on every bit."""
    rng = np.random.default_rng(bits * 1000 + s)
    kw = dict(kw)
    shuffle = kw.pop("shuffle", False)
    table, nhash = _clade_table(rng, sizes, s, **kw)
    if shuffle:
        perm = rng.permutation(len(nhash))
        table, nhash = table[perm].copy(), nhash[perm].copy()
    n = len(nhash)
    lengths = np.full(n, 10 ** 6, dtype=np.uint64)
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n)
    monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "sparse")
    monkeypatch.setenv("MASHGPU_SPARSE_INDEX", "sort")            # (round 5 builds the index by tiles: this test is about the sort)
    monkeypatch.setenv("MASHGPU_SPARSE_SORT_BITS", str(bits))
    t = eng.table_upload(table, nhash, lengths)
    got = eng.compare_tri_host(t)
    assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom)
    rb, re = n // 4, n - 3
    part = eng.compare_tri_host(t, rb, re)
    lo, hi = rb * (rb - 1) // 2, re * (re - 1) // 2
    assert np.array_equal(part["numer"], numer[lo:hi]) and np.array_equal(part["denom"], denom[lo:hi])
    t.free()
    monkeypatch.delenv("MASHGPU_SPARSE_SORT_BITS")
    monkeypatch.setenv("MASHGPU_SPARSE_SORT_ALL_BITS", "1")
    t = eng.table_upload(table, nhash, lengths)
    assert eng.compare_tri_host(t).tobytes() == got.tobytes()
    t.free()


def test_ctx_options_override_the_environment(eng, oracle, monkeypatch):
    """mg_ctx_set_option: a knob set on the context wins over the environment, NULL hands it back; names outside
    MASHGPU_* are refused.  Shown on MASHGPU_COMPARE_KERNEL: "sparse" makes the index engine take a job it would leave to
    the tile engine (too few pairs) -- and fail loudly on a table it cannot take (a hash equal to the padding value)."""
    rng = np.random.default_rng(4)
    n, s = 60, 32
    table = np.sort(rng.integers(1, 1 << 40, size=(n, s)).astype(np.uint64), axis=1)
    nhash = np.full(n, s, dtype=np.uint32)
    lengths = np.full(n, 10 ** 6, dtype=np.uint64)
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n)
    bad = table.copy()
    bad[7, s - 1] = np.uint64(abi.HASH_PAD)                       # a real hash with the padding's value: outside the index engine
    monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "generic")
    try:
        eng.set_option("MASHGPU_COMPARE_KERNEL", "sparse")        # the context's setting wins
        t = eng.table_upload(table, nhash, lengths)
        got = eng.compare_tri_host(t)
        assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom)
        t.free()
        tb = eng.table_upload(bad, nhash, lengths)
        with pytest.raises(abi.MashGpuError, match="sparse engine cannot take"):
            eng.compare_tri_host(tb)
        eng.set_option("MASHGPU_COMPARE_KERNEL", None)            # back to the environment: the generic kernel takes it
        got = eng.compare_tri_host(tb)
        assert len(got) == n * (n - 1) // 2
        tb.free()
        with pytest.raises(abi.MashGpuError, match="MASHGPU_"):
            eng.set_option("PATH", "x")
    finally:
        eng.set_option("MASHGPU_COMPARE_KERNEL", None)


def test_compare_sparse_index_of_a_collection_of_many_genome_sizes(eng, oracle, monkeypatch):
    """A table large enough for the index's sort on leading bits (>= 2^22 entries) whose values are NOT spread evenly:
    nine rows in ten keep their hashes below 2^44 (large genomes), the tenth reaches 2^58 -- the low end of the range is
    a thousand times denser than the even-spread rule assumes; the number of bits comes from the rows' largest hashes
    (host_index.cpp).  Every pair against the ORACLE (9.7e6 pairs), through the index built by tiles
    (the default: its buckets are sized from the same density), by the sort with the default bits, with far too few bits
    (thousands of ties, then the fallback) and with every bit sorted."""
    rng = np.random.default_rng(12)
    n, s = 4400, 1000
    pools = [np.sort(rng.choice(np.arange(1, 1 << 22, dtype=np.uint64), 1500, replace=False)) << np.uint64(22) for _ in range(40)]
    table = np.zeros((n, s), dtype=np.uint64)
    for i in range(n):
        if i % 10 == 9:
            row = np.unique(rng.integers(1, 1 << 58, size=s + 16).astype(np.uint64))[:s]
        else:
            own = rng.choice(pools[i % 40], size=800, replace=False)
            priv = rng.integers(1, 1 << 44, size=260).astype(np.uint64)
            row = np.unique(np.concatenate([own, priv]))[:s]
        assert len(row) == s
        table[i] = row
    nhash = np.full(n, s, dtype=np.uint32)
    lengths = np.full(n, 10 ** 6, dtype=np.uint64)
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n)
    assert int(numer.max()) > 100                           # (rows of one pool share hundreds of values)
    monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "sparse")
    for knobs in ((), (("MASHGPU_SPARSE_INDEX", "sort"),), (("MASHGPU_SPARSE_INDEX", "sort"), ("MASHGPU_SPARSE_SORT_BITS", "24")),
                  (("MASHGPU_SPARSE_INDEX", "sort"), ("MASHGPU_SPARSE_SORT_ALL_BITS", "1"))):
        for knob, value in knobs:
            eng.set_option(knob, value)
        try:
            t = eng.table_upload(table, nhash, lengths)
            got = eng.compare_tri_host(t)
            t.free()
        finally:
            for knob, _ in knobs:
                eng.set_option(knob, None)
        assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom), knobs


@pytest.mark.parametrize("kind", ["clusters", "clades_in_order", "ragged", "random"])
def test_sparse_matrix_is_the_whole_matrix(eng, oracle, kind, monkeypatch):
    """mg_compare_tri_sparse_host: the triangle as its exceptions -- every pair with numer >= 1 -- from which the rule
    {0, min(s, |A| + |B|)} and the tables' hash counts give back EVERY {numer, denom} (mg_expand_tri_sparse), byte for byte
    what mg_compare_tri_host returns (CommandTriangle.cpp:159-198 prints mostly the constant).  Tables with clusters
    (candidates), clades listed in order (the index in table order has dense groups: their inner pairs are appended to the
    rows' lists by the dense kernel -- round 4 sent such list jobs to the blocked matrix path), short / empty / copied rows
    (the matrix path behind the same call) and unrelated rows (no exception at all); also a row range, and the same
    through the thresholded results call."""
    rng = np.random.default_rng(5)
    if kind == "clusters":
        table, nhash, _ = synth.clustered_sketches(2200, 1000, clusters=22, seed=9)
    elif kind == "clades_in_order":
        table, nhash = _clade_table(rng, (120, 35, 260, 9), 1000, short_every=0)
    elif kind == "ragged":
        table, nhash, _ = _index_tables("ragged", rng)
    else:
        table, nhash, _ = synth.random_sketches(1500, 1000, seed=3)
    n, s = table.shape
    lengths = np.full(n, 10 ** 6, dtype=np.uint64)
    t = eng.table_upload(table, nhash, lengths)
    full = eng.compare_tri_host(t)
    for i in sorted(set([1, n // 2, n - 1])):
        numer, denom = _oracle_tri(oracle, table, nhash, lengths, i, i + 1)
        lo = i * (i - 1) // 2
        assert np.array_equal(full["numer"][lo:lo + i], numer) and np.array_equal(full["denom"][lo:lo + i], denom), i
    monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "sparse")
    monkeypatch.setenv("MASHGPU_SPARSE_DBG", "1")
    edges = eng.compare_tri_sparse(t)
    assert len(edges) == int(np.count_nonzero(full["numer"]))
    assert np.all(edges["numer"] >= 1)
    key = edges["row"].astype(np.uint64) << np.uint64(32) | edges["col"].astype(np.uint64)
    assert np.all(key[1:] > key[:-1])                       # reference order
    assert eng.expand_tri_sparse(edges, nhash, s, 0, n).tobytes() == full.tobytes()
    rb, re = n // 3, n - 7
    part = eng.compare_tri_sparse(t, rb, re)
    lo, hi = rb * (rb - 1) // 2, re * (re - 1) // 2
    assert eng.expand_tri_sparse(part, nhash, s, rb, re).tobytes() == full[lo:hi].tobytes()
    # the thresholded results of the same table: the list engine with the groups' pairs in it == the matrix path
    res = eng.compare_tri_results(t, 21, KSPACE21, max_d=0.2)
    monkeypatch.setenv("MASHGPU_RESULTS_MATRIX", "1")
    res_m = eng.compare_tri_results(t, 21, KSPACE21, max_d=0.2)
    assert res.tobytes() == res_m.tobytes()
    t.free()


@pytest.mark.parametrize("kernel", ["default", "sparse", "merged", "join"])
def test_one_species_collection(eng, oracle, kernel, monkeypatch):
    """The middle of the similarity range: one species as a tree of descent (workloads/synth.species_sketches)
    -- every pair shares 10 - 50 % of its hashes, no near-copies, no small pool the rows draw from, rows in random order: nothing
    for the dense groups, everything a candidate.  Every pair against the oracle (compareSketches, CommandDistance.cpp:347-385),
    through the engine the dispatch picks, the index engine and the tile engine."""
    n, s = 900, 300
    table, nhash, lengths = synth.species_sketches(n, s, seed=4)
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n)
    assert 0.08 * s < np.percentile(numer, 1) and np.percentile(numer, 99) < 0.6 * s       # the regime the bracket is about
    if kernel != "default":
        monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", kernel)
    t = eng.table_upload(table, nhash, lengths)
    got = eng.compare_tri_host(t)
    assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom)
    t.free()


def test_one_species_takes_the_join_engine(eng, oracle, monkeypatch):
    """A collection of one species large enough for the dispatch to look at (4.5e6 pairs): the shared hashes counted from the
    images say the join engine (compare_join.hip) -- it must be the one that runs (its launches are recorded), on the first
    call and on a cached plan, for the whole triangle (the index built on the clustered copy: results mapped back through
    inv) and for a row range that cuts through blocks (the plain index); every pair against the oracle (compareSketches,
    CommandDistance.cpp:347-385).  Then the same bytes with the engine switched off, and a rect job of the table's own rows
    and strangers against it."""
    n, s = 3000, 256
    table, nhash, lengths = synth.species_sketches(n, s, seed=9)
    nhash = nhash.copy()
    nhash[17] = 0                                           # an empty row, short rows, a copy
    nhash[1200] = 90
    nhash[2999] = 201
    table = table.copy()
    table[777] = table[76]
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n)
    t = eng.table_upload(table, nhash, lengths)
    eng.prof_enable(True)
    for rnd in range(2):                                    # (the second call: the plan cached with the table)
        eng.prof_reset()
        got = eng.compare_tri_host(t)
        assert eng.prof_avg_ms("compare_join")[1] == 1 and eng.prof_avg_ms("compare_fill")[1] == 0, rnd
        assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom), rnd
    rb, re = 1501, 2990
    eng.prof_reset()
    monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "join")   # (3.3e6 pairs: below what the dispatch sends to an index at all)
    t.invalidate()
    got2 = eng.compare_tri_host(t, rb, re)
    # ... and the table's LAST rows (a rank's job on its view): the lists in family order, those rows in a segment of their own
    eng.prof_reset()
    got4 = eng.compare_tri_host(t, 1200, n)
    assert eng.prof_avg_ms("compare_join")[1] == 1
    assert np.array_equal(got4["numer"], numer[1200 * 1199 // 2:]) and np.array_equal(got4["denom"], denom[1200 * 1199 // 2:])
    monkeypatch.delenv("MASHGPU_COMPARE_KERNEL")
    base = rb * (rb - 1) // 2
    cnt = re * (re - 1) // 2 - base
    assert np.array_equal(got2["numer"], numer[base: base + cnt]) and np.array_equal(got2["denom"], denom[base: base + cnt])
    # a LIST job on such a table (thresholded results): the list engine would merge every pair, so the job goes down the matrix
    # path in blocks -- through this engine -- and is filtered on the device; same records as with the engine switched off
    t.invalidate()
    eng.prof_reset()
    res = eng.compare_tri_results(t, 21, KSPACE21, max_d=0.25, capacity=1 << 22)
    assert eng.prof_avg_ms("compare_join")[1] >= 1 and len(res) > 10000
    monkeypatch.setenv("MASHGPU_COMPARE_JOIN", "0")
    t.invalidate()
    res0 = eng.compare_tri_results(t, 21, KSPACE21, max_d=0.25, capacity=1 << 22)
    assert res.tobytes() == res0.tobytes()
    monkeypatch.delenv("MASHGPU_COMPARE_JOIN")
    # (the table above has an empty row and a copy, which the list engine declines by itself; a clean one takes the hand-over)
    clean, cn, cl = synth.species_sketches(n, s, seed=12)
    tc = eng.table_upload(clean, cn, cl)
    eng.prof_reset()
    res = eng.compare_tri_results(tc, 21, KSPACE21, max_d=0.25, capacity=1 << 22)
    assert eng.prof_avg_ms("compare_join")[1] >= 1 and eng.prof_avg_ms("compare_merge")[1] == 0 and len(res) > 10000
    monkeypatch.setenv("MASHGPU_COMPARE_JOIN", "0")
    tc.invalidate()
    assert eng.compare_tri_results(tc, 21, KSPACE21, max_d=0.25, capacity=1 << 22).tobytes() == res.tobytes()
    monkeypatch.delenv("MASHGPU_COMPARE_JOIN")
    tc.free()
    # the other engines on the same table: the same bytes
    monkeypatch.setenv("MASHGPU_COMPARE_JOIN", "0")
    t.invalidate()
    eng.prof_reset()
    got3 = eng.compare_tri_host(t)
    assert eng.prof_avg_ms("compare_join")[1] == 0
    assert got3.tobytes() == got.tobytes()
    monkeypatch.delenv("MASHGPU_COMPARE_JOIN")
    eng.prof_enable(False)
    # rect: 300 rows of the table and 40 strangers as queries
    monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "join")
    other, onh, olen = synth.species_sketches(40, s, seed=10)
    q = np.concatenate([table[1000:1300], other])
    qn = np.concatenate([nhash[1000:1300], onh])
    ql = np.concatenate([lengths[1000:1300], olen])
    tq = eng.table_upload(q, qn, ql)
    rect = eng.compare_rect_host(t, tq)
    monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "merged")
    rect_m = eng.compare_rect_host(t, tq)
    assert rect.tobytes() == rect_m.tobytes()
    for qi in (0, 150, 299, 300, 339):
        a_n = int(qn[qi])
        for r in (0, 17, 76, 777, 1000 + min(qi, 299), 2999):
            o = oracle.compare(q[qi, :a_n], table[r, : int(nhash[r])], 1000, 1000, s, 21, KSPACE21)
            assert (int(rect[qi, r]["numer"]), int(rect[qi, r]["denom"])) == (o.numer, o.denom), (qi, r)
    tq.free()
    t.free()


def test_triangle_rows_are_served_from_a_view_of_the_first_rows(eng, oracle, monkeypatch):
    """A triangle call over rows [rb, re) looks at rows below re only (CommandTriangle.cpp:200-214), so the library derives
    what it needs -- the inverted index -- from a VIEW of the table's first re rows (host_compare.cpp: tri_view; what one rank
    of several does).  Same bytes as with the view switched off, for a range inside the table and for a range that starts at
    row 0 (then the job is the view's WHOLE triangle: clustered index, dense groups); rows against the oracle; the view goes
    with the table's derived data (mg_table_invalidate) and the table stays usable."""
    n, s = 7000, 200
    table, nhash, lengths = synth.clustered_sketches(n, s, clusters=70, seed=21, pool=300, private=80)
    nhash = nhash.copy()
    nhash[100] = 0
    nhash[2500] = 50
    t = eng.table_upload(table, nhash, lengths)
    monkeypatch.setenv("MASHGPU_SPARSE_DBG", "1")
    for rb, re in ((1500, 4500), (0, 3200)):
        t.invalidate()                                      # (a table whose whole index exists is served from that)
        got = eng.compare_tri_host(t, rb, re)
        monkeypatch.setenv("MASHGPU_TRI_PREFIX", "0")
        t.invalidate()
        ref = eng.compare_tri_host(t, rb, re)
        monkeypatch.delenv("MASHGPU_TRI_PREFIX")
        assert got.tobytes() == ref.tobytes(), (rb, re)
        for i in (rb, rb + 777, re - 1):
            nn, dd = _oracle_tri(oracle, table, nhash, lengths, i, i + 1)
            base = i * (i - 1) // 2 - (rb * (rb - 1) // 2 if rb else 0)
            assert np.array_equal(got["numer"][base: base + i], nn) and np.array_equal(got["denom"][base: base + i], dd), (rb, re, i)
    # thresholded results of a range: the list engine on the view
    t.invalidate()
    res = eng.compare_tri_results(t, 21, KSPACE21, max_d=0.2, row_begin=1500, row_end=4500)
    monkeypatch.setenv("MASHGPU_TRI_PREFIX", "0")
    t.invalidate()
    res0 = eng.compare_tri_results(t, 21, KSPACE21, max_d=0.2, row_begin=1500, row_end=4500)
    assert res.tobytes() == res0.tobytes() and len(res) > 1000
    t.free()


def _index_tables(kind, rng):
    """tables for the tile-built index: (table, nhash, may_refuse)"""
    if kind == "clusters":                                  # C3 in small: clusters interleaved over the rows
        table, nhash, _ = synth.clustered_sketches(3000, 1000, clusters=30, seed=5)
        return table, nhash, False
    if kind == "random":                                    # nothing shared: every group is one entry
        table, nhash, _ = synth.random_sketches(2500, 1000, seed=6)
        return table, nhash, False
    if kind == "ragged":                                    # short and empty rows, copies of rows (they stay out of the index)
        table, nhash, _ = synth.clustered_sketches(1800, 600, clusters=12, seed=7)
        for i in range(0, 1800, 7):
            nhash[i] = int(rng.integers(0, 600))
            table[i, nhash[i]:] = np.uint64(abi.HASH_PAD)
        for i in range(5, 1800, 90):
            table[i] = table[i - 3]
            nhash[i] = nhash[i - 3]
        return table, nhash, False
    if kind == "large_sketches":                            # config 5's sketch size: tiles of many windows per row
        table, nhash, _ = synth.clustered_sketches(260, 10000, clusters=4, seed=8, pool=14000, private=2500)
        return table, nhash, False
    if kind == "sizes":                                     # genomes of many sizes: dense low end, sparse high end
        n, s = 2600, 500
        table = np.zeros((n, s), dtype=np.uint64)
        for i in range(n):
            top = 1 << int(rng.integers(46, 60))
            table[i] = np.unique(rng.integers(1, top, size=s + 40).astype(np.uint64))[:s]
        return table, np.full(n, s, dtype=np.uint32), True  # (may be refused for too many buckets: then the sort builds it)
    if kind == "top_bit":                                   # values up to 2^64 - 2
        n, s = 2200, 400
        t = rng.integers(0, (1 << 64) - 1, size=(n, s + 8), dtype=np.uint64)
        t.sort(axis=1)
        table = np.zeros((n, s), dtype=np.uint64)
        for i in range(n):
            table[i] = np.unique(t[i])[:s]
        return table, np.full(n, s, dtype=np.uint32), False
    if kind == "clade":                                     # every value of a pool held by hundreds of rows: buckets beyond the LDS sort,
        table, nhash = _clade_table(rng, (700, 500), 400)    # split once more through global memory (ix_big_bucket_kernel)
        return table, nhash, False
    if kind == "big_clade":                                 # values held by 6 700 rows: more than the LDS takes, streamed out in row order
        table, nhash = _clade_table(rng, (7000, 900), 64)
        return table, nhash, False
    if kind == "twins":                                     # pairs of NEIGHBOURING values with 6 700 holders each: no leading bits tell
        n, s = 7000, 32                                      # them apart -- the tiles must refuse and the sort build the index
        pool = np.unique(rng.integers(1, 1 << 60, size=40).astype(np.uint64))[:34]
        pool[1::2] = pool[0::2] + np.uint64(1)
        table = np.full((n, s), np.uint64(abi.HASH_PAD), dtype=np.uint64)
        nhash = np.zeros(n, dtype=np.uint32)
        for i in range(n):
            r = np.unique(np.concatenate([pool[rng.random(len(pool)) < 0.96], rng.integers(1, 1 << 60, size=2).astype(np.uint64)]))[:s]
            table[i, :len(r)] = r
            nhash[i] = len(r)
        return table, nhash, True
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["clusters", "random", "ragged", "large_sketches", "sizes", "top_bit", "clade", "big_clade", "twins"])
def test_index_built_by_tiles_equals_the_sorted_index(eng, oracle, kind, monkeypatch):
    """Round 5 builds the inverted index without a general sort (index_build.hip: one partition pass over tiles of
    512 rows x a window of buckets, an LDS sort per bucket that finds the groups, the images written back row segment by
    row segment).  MASHGPU_SPARSE_INDEX=verify builds it BOTH ways and compares every array on the device word by word
    (values, rows, group starts and ends, code and position images, the statistics); the triangle it serves is compared
    with the oracle (sampled rows) and with the sort-built index (every byte), for the whole table (the clustered copy)
    and for a row range (the index in table order).  A value held by hundreds or thousands of rows makes a bucket beyond the
    LDS: split once more through global memory, and streamed out if one value alone is beyond it.  Tables the tiles refuse
    (two neighbouring values with thousands of holders each; too many buckets) must say so and come out right through the sort."""
    rng = np.random.default_rng(77)
    table, nhash, may_refuse = _index_tables(kind, rng)
    n, s = table.shape
    lengths = np.full(n, 10 ** 6, dtype=np.uint64)
    monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "sparse")
    monkeypatch.setenv("MASHGPU_SPARSE_INDEX", "sort")
    t = eng.table_upload(table, nhash, lengths)
    want = eng.compare_tri_host(t)
    rb, re = n // 3, n - 5
    want_part = eng.compare_tri_host(t, rb, re)
    t.free()
    rows = sorted(set([1, 2, n // 2, n - 1] + [int(x) for x in rng.integers(1, n, size=6)]))
    for i in rows:                                          # whole rows against the oracle
        numer, denom = _oracle_tri(oracle, table, nhash, lengths, i, i + 1)
        lo = i * (i - 1) // 2
        assert np.array_equal(want["numer"][lo:lo + i], numer) and np.array_equal(want["denom"][lo:lo + i], denom), i
    monkeypatch.setenv("MASHGPU_SPARSE_INDEX", "verify")
    if may_refuse:
        monkeypatch.setenv("MASHGPU_SPARSE_INDEX_MAY_REFUSE", "1")
    t = eng.table_upload(table, nhash, lengths)
    got = eng.compare_tri_host(t)                           # (raises if any array of the two builds differs)
    assert got.tobytes() == want.tobytes()
    assert eng.compare_tri_host(t, rb, re).tobytes() == want_part.tobytes()
    t.free()
    monkeypatch.setenv("MASHGPU_SPARSE_INDEX", "tiles")
    t = eng.table_upload(table, nhash, lengths)
    assert eng.compare_tri_host(t).tobytes() == want.tobytes()
    t.free()


def test_bucket_sorts_refuse_an_order_they_did_not_make(eng, oracle, monkeypatch, capfd):
    """The bucket sorts of index_build.hip (K4) rely on the LDS serving the lanes of one atomic instruction in lane order for
    the ORDER of equal values -- checked in the kernel, not promised by the ISA.  MASHGPU_IX_DEBUG_SWAP swaps
    two entries of one value behind the partition, which leaves exactly what a ticket out of order would leave: the check
    must fire (the build says so), the table must be indexed by the general sort instead, and the results must be the same
    bytes as without the knob and equal to the oracle's rows."""
    table, nhash, lengths = synth.clustered_sketches(3200, 400, clusters=32, seed=41, pool=600, private=160)
    n = table.shape[0]
    monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "sparse")
    t = eng.table_upload(table, nhash, lengths)
    want = eng.compare_tri_host(t)
    monkeypatch.setenv("MASHGPU_IX_DEBUG_SWAP", "1")
    monkeypatch.setenv("MASHGPU_SPARSE_DBG", "1")
    t.invalidate()
    capfd.readouterr()
    got = eng.compare_tri_host(t)
    err = capfd.readouterr().err
    assert "clumped values: sorted instead" in err, err[-2000:]
    assert got.tobytes() == want.tobytes()
    for i in (5, n // 2, n - 1):
        numer, denom = _oracle_tri(oracle, table, nhash, lengths, i, i + 1)
        lo = i * (i - 1) // 2
        assert np.array_equal(got["numer"][lo:lo + i], numer) and np.array_equal(got["denom"][lo:lo + i], denom), i
    t.free()


def test_row_ranges_keep_the_clustered_index_and_its_dense_groups(eng, oracle, monkeypatch):
    """a triangle job over a RANGE of rows -- what one rank of several is given -- used to fall back to the index
    in table order, where the near-copies of an interleaved collection are not neighbours: no dense groups, every pair inside a
    cluster a merge.  Now the job's table is the view of the rows below the range's end (tri_view) and its index keeps the
    rows from the range's start on in a segment of their own (clustered order with a split): the job's rows are a range of
    index rows, the rows of a cluster are neighbours inside each segment, and the dense pairs kernel runs.  Same bytes as the
    slices of the whole triangle; rows against the oracle."""
    n, s = 6000, 256
    table, nhash, lengths = synth.clustered_sketches(n, s, clusters=5, seed=51, pool=280, private=12, keep_p=0.95)       # families of 1 200 rows
    nhash = nhash.copy()
    nhash[4100] = 0
    nhash[2222] = 100
    t = eng.table_upload(table, nhash, lengths)
    whole = eng.compare_tri_host(t)
    # (the pairs ACROSS the cut share hundreds of hashes each: left to itself the dispatch hands such a range to the join
    #  engine -- the same bytes, asserted at the end; this test is about the index engine's groups)
    monkeypatch.setenv("MASHGPU_COMPARE_JOIN", "0")
    eng.prof_enable(True)
    for rb, re in ((3000, 6000), (2000, 4500), (4242, 6000)):
        t.invalidate()
        eng.prof_reset()
        got = eng.compare_tri_host(t, rb, re)
        assert eng.prof_avg_ms("compare_dense")[1] >= 1, (rb, re)       # the groups' pairs went through dn_pairs_kernel
        lo, hi = rb * (rb - 1) // 2, re * (re - 1) // 2
        assert got.tobytes() == whole[lo:hi].tobytes(), (rb, re)
    eng.prof_enable(False)
    for i in (3000, 4100, 5999):
        numer, denom = _oracle_tri(oracle, table, nhash, lengths, i, i + 1)
        lo = i * (i - 1) // 2
        assert np.array_equal(whole["numer"][lo:lo + i], numer) and np.array_equal(whole["denom"][lo:lo + i], denom), i
    monkeypatch.delenv("MASHGPU_COMPARE_JOIN")
    t.invalidate()
    assert eng.compare_tri_host(t, 3000, 6000).tobytes() == whole[3000 * 2999 // 2:].tobytes()       # (whatever engine the dispatch picks)
    # the same ranges with the clustering switched off: the same bytes (the plain index)
    monkeypatch.setenv("MASHGPU_COMPARE_CLUSTER", "0")
    t.invalidate()
    got = eng.compare_tri_host(t, 3000, 6000)
    assert got.tobytes() == whole[3000 * 2999 // 2:].tobytes()
    monkeypatch.delenv("MASHGPU_COMPARE_CLUSTER")
    t.free()
    # families of a hundred rows: a range job keeps the table's own order (the split order would cost more than it saves) --
    # and gives the same bytes when it is forced
    small, snh, sln = synth.clustered_sketches(5000, 256, clusters=50, seed=52, pool=280, private=12, keep_p=0.95)
    ts = eng.table_upload(small, snh, sln)
    a = eng.compare_tri_host(ts, 2000, 5000)
    monkeypatch.setenv("MASHGPU_SPLIT_ALWAYS", "1")
    ts.invalidate()
    b = eng.compare_tri_host(ts, 2000, 5000)
    assert a.tobytes() == b.tobytes()
    numer, denom = _oracle_tri(oracle, small, snh, sln, 4999, 5000)
    assert np.array_equal(a["numer"][-4999:], numer) and np.array_equal(a["denom"][-4999:], denom)
    ts.free()


def test_dispatch_costs_are_learned_per_context(eng, monkeypatch, capfd):
    """the cost table of the compare dispatch (SparseCosts) is a per-context copy of the defaults that
    moves half way towards what the context's own launches measure -- the phases of every job seen for the first time are
    timed, those of a millisecond and more count -- and never further than a factor of four from the defaults.  Shown on a job
    whose fill runs for more than a millisecond (1.15e9 pairs, on a table that has its index: beside an index build the fill is not timed):
    the context reports a fill rate that is no longer the default's and lies
    inside the clamp; a small job teaches nothing (its phases are their launches); MASHGPU_COSTS_FIXED keeps the defaults;
    the results do not depend on any of it."""
    import re
    import torch
    from workloads import synth_torch
    dev = torch.device("cuda", 0)
    default_fill = 4.5e12

    def fill_rate(err):
        lines = [l for l in err.splitlines() if l.startswith("compare costs (context")]
        return [float(re.search(r"fill ([0-9.e+-]+) B/s", l).group(1)) for l in lines]

    n, s = 48000, 32                                        # (9.2 GB of fill: 1.4 ms and more, the floor is 1 ms)
    h, nh, ln = synth_torch.random_sketch_table(n, s, device=dev)
    out = torch.empty((n * (n - 1) // 2, 2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t = eng.table_wrap(h.data_ptr(), nh.data_ptr(), ln.data_ptr(), n, s, keep=(h, nh, ln))
    small, snh, sln = synth.clustered_sketches(4000, 300, clusters=40, seed=61, pool=450, private=120)
    ts = eng.table_upload(small, snh, sln)
    monkeypatch.setenv("MASHGPU_SPARSE_DBG", "1")
    eng.set_option("MASHGPU_COSTS_FIXED", "0")             # (the module's context keeps the defaults otherwise)
    capfd.readouterr()
    a = eng.compare_tri_host(ts)                           # phases of microseconds: nothing learned
    rates = fill_rate(capfd.readouterr().err)
    assert rates and all(r == default_fill for r in rates), rates
    eng.compare_tri_dev(t, 0, n, out.data_ptr())           # (per table the fill runs beside the index build: not a price)
    eng.compare_tri_dev(t, 1000, n, out.data_ptr())        # other rows of the indexed table: a fill of milliseconds by itself
    eng.compare_tri_dev(t, 0, n, out.data_ptr())
    torch.cuda.synchronize()
    rates = fill_rate(capfd.readouterr().err)
    assert rates and rates[-1] != default_fill and default_fill / 4.0001 <= rates[-1] <= default_fill * 4.0001, rates
    sums = [int(out[:, 0].sum(dtype=torch.int64).item()), int(out[:, 1].sum(dtype=torch.int64).item())]
    eng.set_option("MASHGPU_COSTS_FIXED", "1")
    t.invalidate()
    ts.invalidate()
    capfd.readouterr()
    eng.compare_tri_dev(t, 0, n, out.data_ptr())
    torch.cuda.synchronize()
    b = eng.compare_tri_host(ts)
    assert not fill_rate(capfd.readouterr().err)
    assert sums == [int(out[:, 0].sum(dtype=torch.int64).item()), int(out[:, 1].sum(dtype=torch.int64).item())] and a.tobytes() == b.tobytes()
    t.free()
    ts.free()


def test_dense_groups_survive_leader_lists_that_overflow(eng, oracle, monkeypatch, capfd):
    """The build by tiles appends the dense groups' leaders to a thousand lists of fixed room, by bucket; a list that overflows
    used to cost the table its dense groups.  With the room forced to 8 entries (MASHGPU_DENSE_LEAD_CAP) the
    leaders must be found a second time with the room they asked for: the dense pairs kernel still runs, and the bytes are
    those of the run with room to spare."""
    table, nhash, lengths = synth.clustered_sketches(4000, 400, clusters=40, seed=43, pool=440, private=20, keep_p=0.95)
    monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "sparse")
    t = eng.table_upload(table, nhash, lengths)
    eng.prof_enable(True)
    eng.prof_reset()
    want = eng.compare_tri_host(t)
    assert eng.prof_avg_ms("compare_dense")[1] >= 1
    monkeypatch.setenv("MASHGPU_DENSE_LEAD_CAP", "8")
    monkeypatch.setenv("MASHGPU_SPARSE_DBG", "1")
    t.invalidate()
    eng.prof_reset()
    capfd.readouterr()
    got = eng.compare_tri_host(t)
    err = capfd.readouterr().err
    assert "found again with that room" in err, err[-2000:]
    assert eng.prof_avg_ms("compare_dense")[1] >= 1
    eng.prof_enable(False)
    assert got.tobytes() == want.tobytes()
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 3999, 4000)
    lo = 3999 * 3998 // 2
    assert np.array_equal(got["numer"][lo:], numer) and np.array_equal(got["denom"][lo:], denom)
    t.free()


def test_table_invalidate_after_the_buffers_changed(eng, oracle, monkeypatch):
    """mg_table_invalidate: a wrapped table whose buffers were refilled is answered from the NEW contents -- index, plans,
    classes of copies, short rows all rebuilt (first table: clusters; second: other values, some rows short, some
    copies) -- and the blocks of the dropped index are reused (the context's pool), also by a table of another shape."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(91)
    n, s = 400, 120
    def make(seed, short=False, copies=False):
        r = np.random.default_rng(seed)
        pool = np.sort(r.choice(np.arange(1, 10 ** 7, dtype=np.uint64), 40 * 200, replace=False)).reshape(40, 200)
        table = np.full((n, s), np.uint64(abi.HASH_PAD), dtype=np.uint64)
        nhash = np.zeros(n, dtype=np.uint32)
        for i in range(n):
            c = pool[i % 40]
            k = int(r.integers(20, s)) if short and i % 5 == 0 else s
            row = np.sort(r.choice(c, k, replace=False))
            table[i, :k] = row
            nhash[i] = k
        if copies:
            for i in range(10, n, 37):
                table[i] = table[i - 7]; nhash[i] = nhash[i - 7]
        return table, nhash
    lengths = np.full(n, 10 ** 6, dtype=np.uint64)
    t1, n1 = make(1)
    t2, n2 = make(2, short=True, copies=True)
    monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "sparse")
    dh = torch.from_numpy(t1.view(np.int64)).to(dev)
    dn = torch.from_numpy(n1.view(np.int32)).to(dev)
    dl = torch.from_numpy(lengths.view(np.int64)).to(dev)
    out = torch.zeros((n * (n - 1) // 2, 2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t = eng.table_wrap(dh.data_ptr(), dn.data_ptr(), dl.data_ptr(), n, s, keep=(dh, dn, dl))
    for table, nhash in ((t1, n1), (t2, n2), (t1, n1)):
        dh.copy_(torch.from_numpy(table.view(np.int64)))
        dn.copy_(torch.from_numpy(nhash.view(np.int32)))
        torch.cuda.synchronize()
        t.invalidate()
        for _ in range(2):                                  # the first pass on the new contents and a pass over its cached plan
            out.zero_()
            eng.compare_tri_dev(t, 0, n, out.data_ptr())
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n)
            assert np.array_equal(got[:, 0], numer) and np.array_equal(got[:, 1], denom)
    # a row range (its own plan and slice of the visiting order), after another invalidate
    t.invalidate()
    rb, re = 150, 390
    eng.compare_tri_dev(t, rb, re, out.data_ptr())
    torch.cuda.synchronize()
    numer, denom = _oracle_tri(oracle, t1, n1, lengths, rb, re)
    got = out.cpu().numpy()[: len(numer)]
    assert np.array_equal(got[:, 0], numer) and np.array_equal(got[:, 1], denom)
    t.free()
    eng.trim()


@pytest.mark.parametrize("pace", [None, "2,0", "48,300", "4096,0,64", "at the sorts"])
def test_the_fill_beside_the_index_build_changes_nothing(eng, oracle, pace, monkeypatch):
    """A matrix job on a table without an index writes its constant -- {0, s} for every pair -- on a stream of its own WHILE
    the index is built (SparseJobRun::prefill: chunks of the output taken from a counter, what is left ended at full speed on
    the context's stream).  Switched on for small tables here (by default jobs of 1e8 pairs and more), at the default pace, with
    two workgroups, with a pace so slow that the build ends first, and at full speed: a collection (the inverted index), one
    species (the join engine takes the job from under the fill), a table of nothing but copies (the constant is another one),
    a table the index refuses (a hash equal to the padding value: the tile engine), a range of the last rows (the view), a
    rect job, and an output that starts 8 bytes off a 16-byte boundary -- every pair against the oracle (compareSketches,
    CommandDistance.cpp:347-385); a warm pass launches nothing aside.  "at the sorts": the fill of a LONG build (3e8 entries and
    more by default, any here) starts when the bucket sorts are queued, not with the build -- a table whose build never gets
    there fills behind it."""
    import torch
    monkeypatch.setenv("MASHGPU_FILL_ASIDE_MIN_PAIRS", "1")
    late = pace == "at the sorts"
    if late:
        monkeypatch.setenv("MASHGPU_FILL_ASIDE_AT_SORT", "1")
    elif pace:
        monkeypatch.setenv("MASHGPU_FILL_ASIDE", pace)
    eng.prof_enable(True)
    n, s = 3000, 256
    tables = {"collection": synth.clustered_sketches(n, s, clusters=60, seed=21, pool=400, private=100),
              "species": synth.species_sketches(n, s, seed=22)}
    one, _, _ = synth.random_sketches(1, s, seed=4)
    tables["copies"] = (np.repeat(one[:1], n, axis=0), np.full(n, s, dtype=np.uint32), np.full(n, 10 ** 6, dtype=np.uint64))
    ref, rn, rl = (x.copy() for x in tables["collection"])
    ref[5, int(rn[5]) - 1] = np.uint64(abi.HASH_PAD)         # (the largest value of a row: the padding value itself)
    tables["refused"] = (ref, rn, rl)
    for name, (table, nhash, lengths) in tables.items():
        numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n)
        t = eng.table_upload(table, nhash, lengths)
        eng.prof_reset()
        got = eng.compare_tri_host(t)
        # (nothing but copies: the build says so, the launch ends at its next chunk and another one writes {c, c})
        aside = eng.prof_avg_ms("compare_fill_aside")[1]
        if late:                                            # (copies: known before the sorts; refused: no sorts)
            assert aside == (1 if name in ("collection", "species") else aside) and aside <= 1, (name, aside)
        else:
            assert aside == (2 if name == "copies" else 1), name
        assert eng.prof_avg_ms("compare_fill")[1] == (0 if name in ("species", "refused") else 1), name
        assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom), name
        if name == "species":
            assert eng.prof_avg_ms("compare_join")[1] == 1
        eng.prof_reset()
        got = eng.compare_tri_host(t)                       # warm: the index (or its refusal) is there
        assert eng.prof_avg_ms("compare_fill_aside")[1] == 0, name
        assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom), name
        if name == "collection":
            # the last rows on a fresh table: the index of the view, the fill beside it
            t.invalidate()
            rb = 900                                        # (4.09e6 pairs: still the index's)
            eng.prof_reset()
            part = eng.compare_tri_host(t, rb, n)
            base = rb * (rb - 1) // 2
            assert eng.prof_avg_ms("compare_fill_aside")[1] == 1
            assert np.array_equal(part["numer"], numer[base:]) and np.array_equal(part["denom"], denom[base:])
            # ... and into device memory 8 bytes off a 16-byte boundary
            t.invalidate()
            pairs = n * (n - 1) // 2
            buf = torch.zeros((pairs + 3, 2), dtype=torch.int32, device="cuda")
            eng.compare_tri_dev(t, 0, n, buf.data_ptr() + 8)
            torch.cuda.synchronize()
            dev = buf.cpu().numpy()
            assert np.array_equal(dev[1:pairs + 1, 0].astype(np.uint32), numer) and np.array_equal(dev[1:pairs + 1, 1].astype(np.uint32), denom)
            assert not dev[0].any() and not dev[pairs + 1:].any()
            # rect: a table of queries against the collection, the collection's index not built yet
            t.invalidate()
            q = eng.table_upload(table[100:1700], nhash[100:1700], lengths[100:1700])
            eng.prof_reset()
            rect = eng.compare_rect_host(t, q)
            assert eng.prof_avg_ms("compare_fill_aside")[1] == 1
            for qi in (0, 7, 800, 1599):                    # (pairs of the triangle, read across)
                a = qi + 100
                for b in (0, 99, 101, 1500, 2999):
                    hi, lo = max(a, b), min(a, b)
                    k = hi * (hi - 1) // 2 + lo
                    assert (int(rect["numer"][qi, b]), int(rect["denom"][qi, b])) == (int(numer[k]), int(denom[k])), (qi, b)
            q.free()
        t.free()
    eng.prof_enable(False)


@pytest.mark.parametrize("kernel", ["merged", "sparse"])
@pytest.mark.parametrize("count", [1, 24, 1000])
def test_compare_table_of_copies(eng, oracle, kernel, count, monkeypatch):
    """Nothing but copies of ONE sketch -- full, short (24 of s = 1000 hashes: two short sketches that share
    nothing would be {0, 48}, copies are {24, 24}) and of a single hash: every pair {c, c}; whole triangle and a row
    range.  (The inverted-index engine answers a table of nothing but copies with its fill alone;
    tools/compare_fuzz.py found the short-pairs pass overwriting that fill before it shipped -- by default the pairs
    inside a class of copies are written by their own kernel after the fill.)"""
    _set_kernel(monkeypatch, kernel)
    one, _, _ = synth.random_sketches(1, 1000, seed=4)
    n = 300
    table = np.full((n, 1000), np.uint64(abi.HASH_PAD), dtype=np.uint64)
    table[:, :count] = one[0, :count]
    nhash = np.full(n, count, dtype=np.uint32)
    lengths = np.full(n, 10 ** 6, dtype=np.uint64)
    t = eng.table_upload(table, nhash, lengths)
    got = eng.compare_tri_host(t)
    assert np.all(got["numer"] == count) and np.all(got["denom"] == count)
    part = eng.compare_tri_host(t, 100, 250)
    assert len(part) == abi.tri_pairs(100, 250) and np.all(part["numer"] == count) and np.all(part["denom"] == count)
    numer, denom = _oracle_tri(oracle, table[:40], nhash[:40], lengths[:40], 0, 40)
    assert np.all(numer == count) and np.all(denom == count)
    t.free()


@pytest.mark.parametrize("kernel", ["merged", "sparse", "plain", "windows61"])
@pytest.mark.parametrize("top", [0xFFFFFFFF, 0xFFFFFFFFFFFFFFFE, 0xFFFFFFFE00000000])
def test_compare_values_at_the_top_of_the_hash_range(eng, oracle, kernel, top, monkeypatch):
    """32-bit sketches reaching 0xFFFFFFFF / 64-bit sketches reaching 2^64-2: the prefix image
    reserves 0xFFFFFFFE (sentinel) and 0xFFFFFFFF (padding), so the shift must keep real
    prefixes below them; short rows, shared top values and a row holding only the top value."""
    _set_kernel(monkeypatch, kernel)
    rng = np.random.default_rng(top % 1000)
    n, s = 40, 64
    table = np.full((n, s), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    nhash = np.zeros(n, dtype=np.uint32)
    pool = np.unique(np.concatenate([
        (rng.integers(0, 2 ** 62, 150).astype(np.uint64) % np.uint64(top)),
        np.array([top, top - 1, top - 2, top - 3, top >> 1, (top >> 1) + 1, 0, 1], dtype=np.uint64)]))
    for i in range(n):
        k = int(rng.integers(1, s + 1))
        row = np.sort(rng.choice(pool, size=min(k, len(pool)), replace=False))
        if i % 3 == 0:
            row = np.unique(np.concatenate([row[: s - 2], np.array([top - 1, top], dtype=np.uint64)]))
        table[i, : len(row)] = row
        nhash[i] = len(row)
    table[5, :] = np.uint64(0xFFFFFFFFFFFFFFFF); table[5, 0] = np.uint64(top); nhash[5] = 1
    lengths = np.full(n, 1000, dtype=np.uint64)
    t = eng.table_upload(table, nhash, lengths)
    got = eng.compare_tri_host(t)
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n)
    assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom)
    t.free()


@pytest.mark.parametrize("kernel", ["merged", "sparse", "plain", "windows90"])
def test_compare_mixed_hash_densities(eng, oracle, kernel, monkeypatch):
    """Sketches of very different genome sizes in one table (hash ranges from 2^44 to 2^64):
    the merged kernel tiles rows by density class and compares every class through its own
    prefix image; triangle and rect, incl. related sketches across classes and short rows."""
    _set_kernel(monkeypatch, kernel)
    rng = np.random.default_rng(77)
    n, s = 210, 256
    table = np.full((n, s), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    nhash = np.zeros(n, dtype=np.uint32)
    shared = np.sort(rng.integers(1, 2 ** 44, 400).astype(np.uint64))          # small values: in every range
    for i in range(n):
        bits = [44, 50, 54, 58, 62, 64][int(rng.integers(0, 6))] if i % 7 else 64
        k = s if i % 5 else int(rng.integers(1, s))
        top = (1 << bits) - 2
        own = rng.integers(1, 2 ** 63, 2 * s).astype(np.uint64) % np.uint64(top)
        take = np.concatenate([own, rng.choice(shared, size=int(rng.integers(0, 60)), replace=False)])
        row = np.unique(take)
        if len(row) > k:
            row = np.sort(rng.choice(row, size=k, replace=False)) if i % 3 else row[:k]
        table[i, : len(row)] = row
        nhash[i] = len(row)
    table[50] = table[49]; nhash[50] = nhash[49]                               # identical pair
    lengths = np.full(n, 5000, dtype=np.uint64)
    t = eng.table_upload(table, nhash, lengths)
    got = eng.compare_tri_host(t)
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n)
    assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom)
    assert numer.max() >= min(nhash[49], s) > 0
    got2 = eng.compare_tri_host(t, 33, 170)
    n2, d2 = _oracle_tri(oracle, table, nhash, lengths, 33, 170)
    assert np.array_equal(got2["numer"], n2) and np.array_equal(got2["denom"], d2)
    tq = eng.table_upload(table[100:160], nhash[100:160], lengths[100:160])
    rect = eng.compare_rect_host(t, tq)
    for q in range(60):
        for r in range(0, n, 3):
            i, j = max(q + 100, r), min(q + 100, r)
            if i == j:
                continue
            idx = i * (i - 1) // 2 + j
            assert (rect["numer"][q, r], rect["denom"][q, r]) == (numer[idx], denom[idx]), (q, r)
    t.free(); tq.free()


@pytest.mark.parametrize("kernel", ["merged", "sparse", "plain", "windows13", "windows200"])
@pytest.mark.parametrize("seed", range(24))
def test_compare_random_tables_vs_oracle(eng, oracle, seed, kernel, monkeypatch):
    """Randomised tables: any sketch size, ragged / empty / identical rows, values shared between
    rows (several rows of a tile holding the same value), hash ranges from 2^20 to 2^64, random
    row ranges, triangle and rect -- the merged kernel against the oracle, bit for bit."""
    _set_kernel(monkeypatch, kernel)
    rng = np.random.default_rng(1000 + seed)
    s = int(rng.choice([1, 2, 3, 5, 17, 64, 100, 257, 600, 1000, 1024, 1500]))
    n = int(rng.integers(2, 70))
    table = np.full((n, s), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    nhash = np.zeros(n, dtype=np.uint32)
    pool = rng.integers(1, 2 ** 63, 3 * s + 8).astype(np.uint64)
    for i in range(n):
        bits = int(rng.choice([20, 33, 44, 54, 60, 64]))
        u = rng.random()
        if u < 0.08:
            continue                                                  # empty row
        if u < 0.2 and i > 0:
            table[i] = table[int(rng.integers(0, i))]                 # identical to an earlier row
            nhash[i] = nhash[np.where((table[:i] == table[i]).all(axis=1))[0][0]]
            continue
        k = int(rng.integers(1, s + 1)) if rng.random() < 0.4 else s
        top = np.uint64((1 << bits) - 2)
        own = rng.integers(1, 2 ** 63, 2 * s + 4).astype(np.uint64) % top
        shared = rng.choice(pool, size=int(rng.integers(0, min(len(pool), s) + 1)), replace=False) % top
        row = np.unique(np.concatenate([own, shared]))
        row = row[:k] if rng.random() < 0.5 else np.sort(rng.choice(row, size=min(k, len(row)), replace=False))
        table[i, : len(row)] = row
        nhash[i] = len(row)
    lengths = rng.integers(1000, 10 ** 7, n).astype(np.uint64)
    t = eng.table_upload(table, nhash, lengths)
    kmer = 21 if s < 1000 else 31
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n, k=kmer, kspace=4.0 ** kmer)
    got = eng.compare_tri_host(t)
    assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom), (seed, s, n)
    lo = int(rng.integers(0, n)); hi = int(rng.integers(lo, n + 1))
    g2 = eng.compare_tri_host(t, lo, hi)
    n2, d2 = _oracle_tri(oracle, table, nhash, lengths, lo, hi, k=kmer, kspace=4.0 ** kmer)
    assert np.array_equal(g2["numer"], n2) and np.array_equal(g2["denom"], d2), (seed, lo, hi)
    q0 = int(rng.integers(0, n)); q1 = int(rng.integers(q0, n + 1))
    if q1 > q0:
        tq = eng.table_upload(table[q0:q1], nhash[q0:q1], lengths[q0:q1])
        rect = eng.compare_rect_host(t, tq)
        for q in range(q1 - q0):
            for r in range(n):
                i, j = max(q + q0, r), min(q + q0, r)
                if i == j:
                    exp = (nhash[i], nhash[i]) if nhash[i] <= s else (s, s)
                    assert (rect["numer"][q, r], rect["denom"][q, r]) == (min(nhash[i], s), min(nhash[i], s)), (seed, q, r)
                else:
                    idx = i * (i - 1) // 2 + j
                    assert (rect["numer"][q, r], rect["denom"][q, r]) == (numer[idx], denom[idx]), (seed, q, r)
        tq.free()
    t.free()


@pytest.mark.parametrize("kernel", ["merged", "sparse", "plain", "windows100", "windows333"])
@pytest.mark.parametrize("s,n,seed", [(1000, 150, 0), (300, 260, 1), (2000, 90, 2), (1000, 40, 3)])
def test_compare_wide_window_tiles(eng, oracle, s, n, seed, kernel, monkeypatch):
    """Window tiles list up to 32 rows of ONE density class (the default engine for s >= 200: two
    windows or so, pairs decided exactly at the end of a window's part).  Rows of equal density,
    with everything that stresses the carried state: clusters of related rows (consecutive AND
    interleaved), identical rows, short rows, a value shared by every row, rows that are subsets of
    others -- triangle and rect against the oracle, bit for bit."""
    _set_kernel(monkeypatch, kernel)
    rng = np.random.default_rng(77 + seed)
    top = np.uint64(1) << np.uint64(54)
    table = np.full((n, s), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    nhash = np.zeros(n, dtype=np.uint32)
    nclu = 5
    pools = [rng.integers(1, 2 ** 63, 2 * s).astype(np.uint64) % top for _ in range(nclu)]
    everywhere = np.uint64(int(top) // 3)
    for i in range(n):
        u = rng.random()
        c = (i // 12) % nclu if i < n // 2 else i % nclu             # consecutive runs, then interleaved
        own = rng.integers(1, 2 ** 63, 2 * s).astype(np.uint64) % top
        if u < 0.05 and i > 0:
            j = int(rng.integers(0, i))
            table[i], nhash[i] = table[j], nhash[j]                   # identical to an earlier row
            continue
        if u < 0.45:
            keep = pools[c][rng.random(2 * s) < rng.choice([0.3, 0.7, 0.95])]
            vals = np.concatenate([keep, own[: s // 2]])
        else:
            vals = own
        vals = np.unique(np.concatenate([vals, [everywhere]]))
        # equal density: keep the values below a common bound (about s of them), not the s smallest
        bound = np.uint64(int(top) * min(1.0, s / len(vals)))
        vals = vals[vals < bound][:s]
        if u > 0.93:
            vals = vals[: int(rng.integers(1, len(vals) + 1))]        # short row (a prefix: same density)
        if 0.88 < u <= 0.93 and i > 0:
            j = int(rng.integers(0, i))
            vals = table[j, : nhash[j]][rng.random(int(nhash[j])) < 0.5]   # subset of an earlier row
        table[i, : len(vals)] = vals
        nhash[i] = len(vals)
    lengths = rng.integers(10 ** 5, 10 ** 7, n).astype(np.uint64)
    t = eng.table_upload(table, nhash, lengths)
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n)
    got = eng.compare_tri_host(t)
    bad = np.flatnonzero((got["numer"] != numer) | (got["denom"] != denom))
    assert bad.size == 0, (kernel, s, n, bad[:5], got[bad[:5]], numer[bad[:5]], denom[bad[:5]])
    lo = n // 3
    g2 = eng.compare_tri_host(t, lo, n - 5)
    n2, d2 = _oracle_tri(oracle, table, nhash, lengths, lo, n - 5)
    assert np.array_equal(g2["numer"], n2) and np.array_equal(g2["denom"], d2)
    q = np.sort(rng.choice(n, size=min(n, 37), replace=False))
    tq = eng.table_upload(table[q], nhash[q], lengths[q])
    rect = eng.compare_rect_host(t, tq)
    for a_, qi in enumerate(q):
        for r in range(n):
            i, j = max(qi, r), min(qi, r)
            if i == j:
                continue
            idx = i * (i - 1) // 2 + j
            assert (rect["numer"][a_, r], rect["denom"][a_, r]) == (numer[idx], denom[idx]), (kernel, qi, r)
    tq.free()
    t.free()


@pytest.mark.parametrize("kernel", ["merged", "sparse", "plain", "windows40"])
@pytest.mark.parametrize("seed", range(6))
def test_compare_values_sharing_a_prefix(eng, oracle, kernel, seed, monkeypatch):
    """Different 64-bit values that share their 32-bit prefix, inside one row, across rows of a
    tile and between rows and columns: the tile's prefix/value consistency check fails, and
    matches must come from the fully verified exact path (and near-misses must not count)."""
    _set_kernel(monkeypatch, kernel)
    rng = np.random.default_rng(700 + seed)
    n, s = 40, 200
    base = np.unique(rng.integers(2 ** 40, 2 ** 63, 260).astype(np.uint64) & np.uint64(~0xFFFF & (2 ** 64 - 1)))
    table = np.full((n, s), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    nhash = np.zeros(n, dtype=np.uint32)
    for i in range(n):
        pick = rng.choice(base, size=int(rng.integers(60, 150)), replace=False)
        low = rng.integers(0, 4, len(pick)).astype(np.uint64)          # 4 variants per prefix: v, v+1, v+2, v+3
        extra = pick[: len(pick) // 3] + ((low[: len(pick) // 3] + np.uint64(1)) % np.uint64(4))
        row = np.unique(np.concatenate([pick + low, extra]))[:s]
        table[i, : len(row)] = row
        nhash[i] = len(row)
    lengths = np.full(n, 10 ** 6, dtype=np.uint64)
    t = eng.table_upload(table, nhash, lengths)
    got = eng.compare_tri_host(t)
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n)
    assert np.array_equal(got["numer"], numer) and np.array_equal(got["denom"], denom)
    assert 0 < numer.max() < nhash.max()                                # some true matches, never everything
    t.free()


def test_compare_rect_and_golden_dist(eng, oracle, golden_dir):
    """mash dist genomes.msh reads.msh == test/ref/genomes.dist, via rect compare + finish."""
    gh, glens, names = helpers.load_golden_genomes()
    rh, rlen, _ = helpers.load_golden_reads()
    ref = eng.table_upload(gh, np.full(3, 1000, np.uint32), glens)
    qry = eng.table_upload(rh[None, :], np.full(1, 1000, np.uint32), np.array([rlen], np.uint64))
    counts = eng.compare_rect_host(ref, qry)
    assert counts.shape == (1, 3)
    fin = eng.finish_rect(counts, glens, np.array([rlen], np.uint64), 21, KSPACE21)
    lines = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(golden_dir, "genomes.dist"))]
    for i in range(3):
        assert "%d/%d" % (fin["numer"][0, i], fin["denom"][0, i]) == lines[i][4]
        assert "%g" % fin["distance"][0, i] == lines[i][2]
        assert "%g" % fin["p_value"][0, i] == lines[i][3]
    # rect vs triangle consistency on a clustered table, query-major order
    table, nhash, lengths = synth.clustered_sketches(90, 1000, clusters=3, seed=4)
    t = eng.table_upload(table, nhash, lengths)
    tq = eng.table_upload(table[40:75], nhash[40:75], lengths[40:75])
    rect = eng.compare_rect_host(t, tq)
    tri = eng.compare_tri_host(t)
    for q in range(35):
        for r in range(90):
            i, j = max(q + 40, r), min(q + 40, r)
            if i == j:
                assert rect["numer"][q, r] == nhash[i] and rect["denom"][q, r] == nhash[i]
            else:
                idx = i * (i - 1) // 2 + j
                assert rect[q, r] == tri[idx]
    ref.free(); qry.free(); t.free(); tq.free()


@pytest.mark.parametrize("n,s,nq", [(6001, 1000, 1), (6001, 1000, 3), (6001, 1000, 40), (70001, 64, 1), (70001, 64, 5)])
def test_compare_few_queries_against_many_references(eng, oracle, n, s, nq):
    """The serving shape (a launch with few row tiles) cuts the columns into finer chunks
    (host_compare.cpp::run_compare); chunk boundaries must not show in the result."""
    table, nhash, lengths = synth.clustered_sketches(n + nq, s, clusters=max(1, n // 50), seed=n + nq,
                                                     pool=int(1.5 * s), private=int(0.4 * s))
    ref = eng.table_upload(table[:n], nhash[:n], lengths[:n])
    # queries: the extra rows (members of the first clusters) and, for nq > 1, a copy of a reference
    q_tab, q_nh, q_len = table[n:].copy(), nhash[n:].copy(), lengths[n:].copy()
    if nq > 1:
        q_tab[1], q_nh[1], q_len[1] = table[n // 2], nhash[n // 2], lengths[n // 2]
    qry = eng.table_upload(q_tab, q_nh, q_len)
    got = eng.compare_rect_host(ref, qry)
    assert got.shape == (nq, n)
    for q in range(nq):
        # oracle: the query appended to the references is the last row of a triangle
        t2 = np.concatenate([table[:n], q_tab[q:q + 1]])
        numer, denom = _oracle_tri(oracle, t2, np.append(nhash[:n], q_nh[q]), np.append(lengths[:n], q_len[q]), n, n + 1)
        assert np.array_equal(got["numer"][q], numer) and np.array_equal(got["denom"][q], denom)
    if nq > 1:
        assert got["numer"][1, n // 2] == nhash[n // 2]
    ref.free(); qry.free()


@pytest.mark.parametrize("s_ref,s_qry", [(1000, 600), (400, 1000), (3000, 2000), (2000, 5000)])
def test_compare_rect_with_different_sketch_sizes(eng, oracle, s_ref, s_qry):
    """`mash dist` compares on the smaller of the two sketch sizes (CommandDistance.cpp:313-315);
    the tables keep their own strides and prefix images, the kernels clamp both sides."""
    n_ref, n_qry = 40, 9
    big = max(s_ref, s_qry)
    table, nhash, lengths = synth.clustered_sketches(n_ref + n_qry, big, clusters=3, seed=s_ref + s_qry,
                                                     pool=int(1.5 * big), private=int(0.4 * big))
    rt = np.full((n_ref, s_ref), np.uint64(abi.HASH_PAD), dtype=np.uint64)
    qt = np.full((n_qry, s_qry), np.uint64(abi.HASH_PAD), dtype=np.uint64)
    rn = np.minimum(nhash[:n_ref], s_ref).astype(np.uint32)
    qn = np.minimum(nhash[n_ref:], s_qry).astype(np.uint32)
    qn[2] = s_qry // 5                                         # a short query
    for i in range(n_ref):
        rt[i, : rn[i]] = table[i, : rn[i]]
    for i in range(n_qry):
        qt[i, : qn[i]] = table[n_ref + i, : qn[i]]
    qt[4, : min(s_qry, rn[7])] = rt[7, : min(s_qry, rn[7])]     # a query equal to (a prefix of) a reference
    qn[4] = min(s_qry, rn[7])
    ref = eng.table_upload(rt, rn, lengths[:n_ref])
    qry = eng.table_upload(qt, qn, lengths[n_ref:])
    got = eng.compare_rect_host(ref, qry)
    s_cmp = min(s_ref, s_qry)
    for q in range(n_qry):
        for r in range(n_ref):
            o = oracle.compare(rt[r, : rn[r]], qt[q, : qn[q]], int(lengths[r]), int(lengths[n_ref + q]), s_cmp, 21, KSPACE21)
            assert (got["numer"][q, r], got["denom"][q, r]) == (o.numer, o.denom), (q, r)
    ref.free(); qry.free()


def _py_distance(numer, denom, k):
    """CommandDistance.cpp:387-407 with CPython's libm log (the one the reference links)."""
    import math
    if numer == denom:
        return 0.0
    if numer == 0:
        return 1.0
    j = float(numer) / float(denom)
    return min(1.0, -math.log(2 * j / (1.0 + j)) / k)


@pytest.mark.parametrize("s", [7, 400, 1000, 4096])
def test_compare_filter_edges(eng, oracle, s):
    """Device-side distance filter + compaction == the reference's `distance > maxDistance`
    test applied to every pair, in reference order (triangle and rect, ragged rows)."""
    n, k = 170, 21
    table, nhash, lengths = synth.clustered_sketches(n, s, clusters=6, seed=100 + s, pool=int(1.5 * s) + 2,
                                                     private=max(1, int(0.3 * s)))
    nhash[3] = 0
    nhash[4] = 0
    nhash[8] = max(1, s // 2)
    table[21] = table[20]
    nhash[21] = nhash[20]
    t = eng.table_upload(table, nhash, lengths)
    numer, denom = _oracle_tri(oracle, table, nhash, lengths, 0, n)
    dist = np.array([_py_distance(int(a), int(b), k) for a, b in zip(numer, denom)])
    rows = np.concatenate([np.full(i, i, np.uint32) for i in range(n)])
    cols = np.concatenate([np.arange(i, dtype=np.uint32) for i in range(n)])
    some = np.unique(dist)
    for max_d in [0.0, float(some[len(some) // 3]), float(some[len(some) // 2]), 0.05, 0.3, 0.999999, 1.0]:
        keep = dist <= max_d
        got = eng.compare_tri_filter(t, k, max_d, capacity=16)      # forces the grow-and-retry path
        assert len(got) == int(keep.sum()), max_d
        assert np.array_equal(got["row"], rows[keep]) and np.array_equal(got["col"], cols[keep])
        assert np.array_equal(got["numer"], numer[keep]) and np.array_equal(got["denom"], denom[keep])
        # row sub-range
        lo, hi = 40, 133
        sub = (rows >= lo) & (rows < hi) & keep
        got2 = eng.compare_tri_filter(t, k, max_d, lo, hi)
        assert np.array_equal(got2["row"], rows[sub]) and np.array_equal(got2["col"], cols[sub])
        assert np.array_equal(got2["numer"], numer[sub])
    # rect: queries 30..90 against all rows
    tq = eng.table_upload(table[30:90], nhash[30:90], lengths[30:90])
    full = eng.compare_rect_host(t, tq)
    dfull = np.array([[_py_distance(int(c["numer"]), int(c["denom"]), k) for c in row] for row in full])
    for max_d in [0.0, 0.05, 0.5]:
        got = eng.compare_rect_filter(t, tq, k, max_d, capacity=8)
        qq, rr = np.nonzero(dfull <= max_d)
        assert np.array_equal(got["row"], qq.astype(np.uint32)) and np.array_equal(got["col"], rr.astype(np.uint32))
        assert np.array_equal(got["numer"], full["numer"][qq, rr]) and np.array_equal(got["denom"], full["denom"][qq, rr])
        got2 = eng.compare_rect_filter(t, tq, k, max_d, 10, 25)
        m = (qq >= 10) & (qq < 25)
        assert np.array_equal(got2["row"], qq[m].astype(np.uint32)) and np.array_equal(got2["col"], rr[m].astype(np.uint32))
    t.free(); tq.free()


def _same_bits(a, b):
    return np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))


@pytest.mark.parametrize("engine", ["default", "sparse"])
@pytest.mark.parametrize("max_d,max_p", [(-1.0, -1.0), (1.0, 1.0), (0.2, -1.0), (-1.0, 1e-10), (0.08, 1e-30), (0.0, 1.0)])
def test_device_finish_equals_host_finish(eng, oracle, golden_dir, max_d, max_p, engine, monkeypatch):
    """The tail of compareSketches on the device (finish.hip: distance from a host-libm table,
    p-value by the exact double-double tail, both filters) == mg_finish_*_host BIT FOR BIT: on the
    reference-run vectors, on clustered and ragged tables (many distinct denominators), triangle
    and rect, full records and the compacted survivor list."""
    if engine == "sparse":        # with a filter on, the survivor lists then come straight from the candidate list (no matrix)
        monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "sparse")
    z = np.load(os.path.join(golden_dir, "ref_compare_vectors.npz"))
    cases = [(z["table"], z["nhash"], z["lengths"], int(z["k"]), float(z["kmer_space"]))]
    table, nh, lengths = synth.clustered_sketches(300, 400, clusters=6, seed=9)
    rng = np.random.default_rng(3)
    lengths = rng.integers(10 ** 4, 10 ** 8, len(lengths)).astype(np.uint64)
    cases.append((table, nh, lengths, 21, KSPACE21))
    t2, n2, _ = synth.clustered_sketches(120, 300, clusters=3, seed=4)
    for i in range(0, 120, 3):                                     # ragged rows: denominators below s
        n2[i] = rng.integers(1, 300)
        t2[i, n2[i]:] = np.uint64(abi.HASH_PAD)
    cases.append((t2, n2, rng.integers(500, 10 ** 6, 120).astype(np.uint64), 16, 4.0 ** 16))
    for table, nh, lengths, k, ks in cases:
        n = len(nh)
        t = eng.table_upload(table, nh, lengths)
        counts = eng.compare_tri_host(t)
        host = eng.finish_tri(counts, lengths, 0, n, k, ks, max_d, max_p)
        dev = eng.compare_tri_pairs(t, k, ks, max_d, max_p)
        assert np.array_equal(dev["numer"], host["numer"]) and np.array_equal(dev["denom"], host["denom"])
        assert np.array_equal(dev["pass"], host["pass"])
        assert _same_bits(dev["distance"], host["distance"])
        ok = host["pass"] == 1 if (0 <= max_d < 1) else np.ones(len(host), bool)   # rejected by -d: only `pass` is meaningful
        assert _same_bits(dev["p_value"][ok], host["p_value"][ok])
        # survivors only, compacted on the device, reference order
        res = eng.compare_tri_results(t, k, ks, max_d, max_p, capacity=64)          # small: exercises the retry
        keep = np.flatnonzero(host["pass"] == 1)
        assert len(res) == len(keep)
        ii = np.repeat(np.arange(n), np.arange(n))
        jj = np.concatenate([np.arange(i) for i in range(n)]) if n > 1 else np.zeros(0, int)
        assert np.array_equal(res["row"], ii[keep]) and np.array_equal(res["col"], jj[keep])
        assert np.array_equal(res["numer"], host["numer"][keep]) and np.array_equal(res["denom"], host["denom"][keep])
        assert _same_bits(res["distance"], host["distance"][keep]) and _same_bits(res["p_value"], host["p_value"][keep])
        # rect: a slice of the rows as queries against all
        q = np.arange(n // 4, n // 2)
        tq = eng.table_upload(table[q], nh[q], lengths[q])
        rc = eng.compare_rect_host(t, tq)
        hr = eng.finish_rect(rc, lengths, lengths[q], k, ks, max_d, max_p)
        dr = eng.compare_rect_pairs(t, tq, k, ks, max_d, max_p)
        assert np.array_equal(dr["pass"], hr["pass"]) and _same_bits(dr["distance"], hr["distance"])
        okr = hr["pass"] == 1 if (0 <= max_d < 1) else np.ones(hr.shape, bool)
        assert _same_bits(dr["p_value"][okr], hr["p_value"][okr])
        rr = eng.compare_rect_results(t, tq, k, ks, max_d, max_p)
        kq, kr = np.nonzero(hr["pass"] == 1)
        assert np.array_equal(rr["row"], kq) and np.array_equal(rr["col"], kr)
        assert _same_bits(rr["distance"], hr["distance"][kq, kr]) and _same_bits(rr["p_value"], hr["p_value"][kq, kr])
        tq.free()
        t.free()


def test_device_finish_on_exact_p_values(eng, golden_dir):
    """The 4812 exact p-values of tests/golden/binom_exact_pairs.json through the DEVICE tail
    (mg_finish_tri_dev, finish.hip): the counts of a pair are set to a case's {x, n} and the two rows
    carry its genome lengths, so the device computes r from the lengths and the tail exactly as for a
    real pair.  Device == host bit for bit and within 1 ulp of the exact value, down through the
    denormals to 0 (tests/test_pvalue_exact.py holds the host side of the same cases)."""
    import json
    import struct
    import torch
    cases = json.load(open(os.path.join(golden_dir, "binom_exact_pairs.json")))
    lens = sorted({c["len_ref"] for c in cases} | {c["len_qry"] for c in cases})
    pos = {l: k for k, l in enumerate(lens)}
    n = 2 * len(lens)                                              # every length twice: (len, len) pairs exist too
    lengths = np.repeat(np.array(lens, dtype=np.uint64), 2)
    smax = max(c["n"] for c in cases)                              # the table's sketch size bounds the denominators
    table = np.full((n, smax), np.uint64(abi.HASH_PAD), dtype=np.uint64)
    t = eng.table_upload(table, np.zeros(n, np.uint32), lengths)
    npairs = n * (n - 1) // 2
    by_space = {}
    for c in cases:
        a, b = pos[c["len_ref"]], pos[c["len_qry"]]
        i, j = (2 * a, 2 * b) if a != b else (2 * a + 1, 2 * a)
        if i < j:
            i, j = j, i
        by_space.setdefault(c["kmer_space"], {}).setdefault(i * (i - 1) // 2 + j, []).append(c)
    dev = torch.device("cuda", 0)
    checked = zeros = 0

    def ordint(x):
        v = struct.unpack("<q", struct.pack("<d", x))[0]
        return v if v >= 0 else -(v & 0x7FFFFFFFFFFFFFFF)

    for ks_hex, slots in by_space.items():
        ks = float.fromhex(ks_hex)
        rounds = max(len(v) for v in slots.values())
        for rnd in range(rounds):
            counts = np.zeros(npairs, dtype=abi.COUNTS_DTYPE)
            counts["denom"] = 1
            want = {}
            for idx, lst in slots.items():
                if rnd < len(lst):
                    counts[idx] = (lst[rnd]["x"], lst[rnd]["n"])
                    want[idx] = lst[rnd]
            d_counts = torch.from_numpy(counts.view(np.uint8)).to(dev)
            d_out = torch.zeros(npairs * abi.PAIR_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            eng.finish_tri_dev(t, d_counts.data_ptr(), 0, n, 21, ks, -1.0, -1.0, d_out.data_ptr())
            eng.synchronize()
            got = d_out.cpu().numpy().view(abi.PAIR_DTYPE)
            host = eng.finish_tri(counts, lengths, 0, n, 21, ks)
            assert _same_bits(got["p_value"], host["p_value"]) and _same_bits(got["distance"], host["distance"])
            for idx, c in want.items():
                exact = float.fromhex(c["exact"])
                pv = float(got["p_value"][idx])
                assert abs(ordint(pv) - ordint(exact)) <= 1, (c, pv)
                checked += 1
                zeros += exact == 0.0
    assert checked == len(cases) and zeros >= 100
    t.free()


def test_survivor_lists_from_candidates_equal_the_matrix_path(eng, monkeypatch):
    """`mash triangle -E -d`, `mash dist -d / -v` at a size where the default dispatch takes the inverted
    index: with a filter on, the survivors are computed from the candidate list alone (nothing is filled,
    no 8 B per pair read back) -- same records, same order as the matrix path, triangle (also a row
    range) and rect, three filter settings."""
    table, nh, lengths = synth.clustered_sketches(3100, 256, clusters=31, seed=77, pool=400, private=100)
    lengths = np.random.default_rng(5).integers(10 ** 5, 10 ** 7, 3100).astype(np.uint64)
    t = eng.table_upload(table, nh, lengths)
    q = np.arange(100, 1500)
    tq = eng.table_upload(table[q], nh[q], lengths[q])
    for max_d, max_p in ((0.1, -1.0), (-1.0, 1e-20), (0.3, 1e-5)):
        monkeypatch.setenv("MASHGPU_RESULTS_MATRIX", "1")
        want_t = eng.compare_tri_results(t, 21, KSPACE21, max_d, max_p, capacity=1 << 10)
        want_s = eng.compare_tri_results(t, 21, KSPACE21, max_d, max_p, row_begin=700, row_end=3000)
        want_r = eng.compare_rect_results(t, tq, 21, KSPACE21, max_d, max_p)
        monkeypatch.delenv("MASHGPU_RESULTS_MATRIX")
        eng.prof_enable(True)
        eng.prof_reset()
        got_t = eng.compare_tri_results(t, 21, KSPACE21, max_d, max_p, capacity=1 << 10)
        assert eng.prof_avg_ms("compare_merge")[1] >= 1 and eng.prof_avg_ms("compare_fill")[1] == 0      # the list path ran
        eng.prof_enable(False)
        got_s = eng.compare_tri_results(t, 21, KSPACE21, max_d, max_p, row_begin=700, row_end=3000)
        got_r = eng.compare_rect_results(t, tq, 21, KSPACE21, max_d, max_p)
        assert len(want_t) > 1000 and got_t.tobytes() == want_t.tobytes(), (max_d, max_p)
        assert got_s.tobytes() == want_s.tobytes() and got_r.tobytes() == want_r.tobytes(), (max_d, max_p)
    t.free(); tq.free()


def _device_lists(*lists):
    """Device lists of the sharded tests.  One GPU per box here, so the lists repeat device 0; on a box
    with several GPUs tools/scale_check.sh sets MASHGPU_TEST_DEVICES=0,1,...: the same tests then also run
    on DISTINCT devices (real peer copies, real RCCL rings)."""
    out = [list(l) for l in lists]
    extra = os.environ.get("MASHGPU_TEST_DEVICES")
    if extra:
        out.append([int(x) for x in extra.split(",")])
    return out


@pytest.mark.parametrize("devices", _device_lists([0], [0, 0], [0, 0, 0]))
def test_sharded_compare_equals_single_gpu(eng, oracle, devices, monkeypatch):
    """The multi-GPU entry points (mg_comm local mode, mg_dtable, mg_compare_*_sharded_host): row
    blocks on several contexts, driven by one host thread each, give byte-identical output to the
    single-context calls.  One GPU here: a device list that repeats device 0 runs the sharding,
    the threads and the table replication (device copies); the one-device communicator with
    MASHGPU_COMM_FORCE_RCCL runs the RCCL calls (ncclCommInitAll, grouped ncclBroadcast)."""
    if devices == [0]:
        monkeypatch.setenv("MASHGPU_COMM_FORCE_RCCL", "1")
    table, nh, lengths = synth.clustered_sketches(500, 1000, clusters=7, seed=21)
    rng = np.random.default_rng(8)
    lengths = rng.integers(10 ** 5, 10 ** 7, 500).astype(np.uint64)
    for i in range(0, 500, 50):
        nh[i] = rng.integers(1, 1000)
        table[i, nh[i]:] = np.uint64(abi.HASH_PAD)
    t = eng.table_upload(table, nh, lengths)
    want = eng.compare_tri_host(t)
    comm = abi.LocalComm(devices)
    assert comm.uses_rccl == (devices == [0])
    d = comm.upload(table, nh, lengths)
    got = comm.tri(d, 500)
    assert np.array_equal(got, want)
    assert np.array_equal(comm.tri(d, 500, 123, 456), eng.compare_tri_host(t, 123, 456))
    q = np.arange(40, 170)
    tq = eng.table_upload(table[q], nh[q], lengths[q])
    dq = comm.upload(table[q], nh[q], lengths[q])
    assert np.array_equal(comm.rect(d, dq, 500, len(q)), eng.compare_rect_host(t, tq))
    k, ks = 21, KSPACE21
    for max_d, max_p in ((-1.0, -1.0), (0.1, 1e-20)):
        a, b = comm.tri_pairs(d, 500, k, ks, max_d, max_p), eng.compare_tri_pairs(t, k, ks, max_d, max_p)
        assert a.tobytes() == b.tobytes()
        a, b = comm.rect_pairs(d, dq, 500, len(q), k, ks, max_d, max_p), eng.compare_rect_pairs(t, tq, k, ks, max_d, max_p)
        assert a.tobytes() == b.tobytes()
        a, b = comm.tri_results(d, 500, k, ks, max_d, max_p, capacity=16), eng.compare_tri_results(t, k, ks, max_d, max_p, capacity=1 << 18)
        assert a.tobytes() == b.tobytes() and (len(a) > 0)
        a, b = comm.rect_results(d, dq, len(q), k, ks, max_d, max_p), eng.compare_rect_results(t, tq, k, ks, max_d, max_p, capacity=1 << 18)
        assert a.tobytes() == b.tobytes()
    comm.free(dq)
    comm.free(d)
    comm.close()
    tq.free()
    t.free()


@pytest.mark.parametrize("devices", _device_lists([0], [0, 0, 0]))
def test_sharded_screen_equals_single_gpu(eng, devices, monkeypatch):
    """mg_dscreen: mixture batches dealt to the devices of a local communicator, counters summed
    (ncclReduce on the forced one-rank communicator, host adds on the repeated-device list),
    mixture sketches merged == one mg_screen fed every batch."""
    if devices == [0]:
        monkeypatch.setenv("MASHGPU_COMM_FORCE_RCCL", "1")
    rng = np.random.default_rng(17)
    genomes = [synth.synthetic_genome(40_000 * g, 60_000) for g in range(6)]
    p = eng.params(k=21, s=300)
    hashes, nhash = eng.sketch_host([[bytes(g)] for g in genomes], p)
    lengths = np.full(6, 60_000, dtype=np.uint64)
    batches = []
    for b in range(7):
        recs = []
        for _ in range(300):
            g = genomes[int(rng.integers(0, 4))]
            o = int(rng.integers(0, len(g) - 150))
            recs.append(bytes(g[o:o + 150]))
        batches.append(recs)
    t = eng.table_upload(hashes, nhash, lengths)
    want = eng.screen(t, p, batches)
    comm = abi.LocalComm(devices)
    d = comm.upload(hashes, nhash, lengths)
    got = comm.screen(d, 6, 300, p, batches)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2]
    assert want[0][:4].sum() > 0
    comm.free(d)
    comm.close()
    t.free()


@pytest.mark.parametrize("devices", _device_lists([0], [0, 0], [0, 0, 0, 0, 0]))
def test_sharded_sketch_equals_single_gpu(eng, oracle, devices):
    """mg_sketch_sharded_host: blocks of consecutive sketches balanced by bytes, one host thread per
    context, rows in input order == mg_sketch_host on one context -- sketches of very different sizes
    (one of them most of the bytes), empty ones, more devices than sketches, counts."""
    rng = np.random.default_rng(len(devices))
    sizes = [3000, 120000, 10, 0, 700, 45000, 45000, 21, 9000]
    sketches = [[synth._rand_dna(rng, n)] if n else [b""] for n in sizes]
    sketches[5] = [synth._rand_dna(rng, 20000), b"ACGT", synth._rand_dna(rng, 25000)]        # several records
    p = eng.params(k=21, s=300)
    want = eng.sketch_host(sketches, p, counts=True)
    comm = abi.LocalComm(devices)
    got = comm.sketch(sketches, p, counts=True)
    for a_, b_ in zip(got, want):
        assert np.array_equal(a_, b_)
    two = comm.sketch(sketches[:2], p)                                   # fewer sketches than devices
    assert np.array_equal(two[0], want[0][:2]) and np.array_equal(two[1], want[1][:2])
    h, _, _, _, _ = oracle.sketch_records(sketches[1], oracle.params(k=21, s=300))
    assert np.array_equal(got[0][1, : len(h)], h)
    comm.close()


@pytest.mark.parametrize("devices", _device_lists([0, 0], [0, 0, 0]))
def test_rect_split_by_reference_rows(eng, oracle, devices, monkeypatch):
    """SURVEY 8e for `mash dist`: the larger side is cut.  Few queries against many references: every
    context compares all queries with ITS block of reference rows (a view of its replica, or its own rows
    of a row-sharded table) and the blocks go back into the reference's query-major order -- counts,
    finished records and the survivor list equal the single-context calls byte for byte."""
    table, nh, lengths = synth.clustered_sketches(331, 200, clusters=5, seed=21)
    nh[7] = 50; table[7, 50:] = np.uint64(abi.HASH_PAD)
    lengths = np.random.default_rng(2).integers(10 ** 4, 10 ** 7, 331).astype(np.uint64)
    q = np.array([3, 120, 7, 330])
    t = eng.table_upload(table, nh, lengths)
    tq = eng.table_upload(table[q], nh[q], lengths[q])
    want_c = eng.compare_rect_host(t, tq)
    want_p = eng.compare_rect_pairs(t, tq, 21, KSPACE21, 0.3, 1e-3)
    want_r = eng.compare_rect_results(t, tq, 21, KSPACE21, 0.3, 1e-3)
    assert 0 < len(want_r) < want_c.size
    comm = abi.LocalComm(devices)
    dq = comm.upload(table[q], nh[q], lengths[q])
    for mode in ("replicated", "rows"):
        dr = comm.upload(table, nh, lengths) if mode == "replicated" else comm.upload_rows(table, nh, lengths)
        got_c = comm.rect(dr, dq, 331, 4)                                # 331 references > 4 queries: cut by reference rows
        assert np.array_equal(got_c, want_c), mode
        got_p = comm.rect_pairs(dr, dq, 331, 4, 21, KSPACE21, 0.3, 1e-3)
        assert np.array_equal(got_p["pass"], want_p["pass"]) and _same_bits(got_p["distance"], want_p["distance"])
        ok = want_p["pass"] == 1
        assert _same_bits(got_p["p_value"][ok], want_p["p_value"][ok])
        got_r = comm.rect_results(dr, dq, 4, 21, KSPACE21, 0.3, 1e-3, capacity=8)      # small: exercises the retry
        assert got_r.tobytes() == want_r.tobytes(), mode
        comm.free(dr)
    # the query side cut instead (as before) gives the same
    monkeypatch.setenv("MASHGPU_RECT_SPLIT", "queries")
    dr = comm.upload(table, nh, lengths)
    assert np.array_equal(comm.rect(dr, dq, 331, 4), want_c)
    monkeypatch.delenv("MASHGPU_RECT_SPLIT")
    # a row-sharded table cannot be a triangle's table
    drows = comm.upload_rows(table, nh, lengths)
    with pytest.raises(abi.MashGpuError):
        comm.tri(drows, 331)
    comm.free(drows); comm.free(dr); comm.free(dq)
    comm.close()
    t.free(); tq.free()


def test_rank_communicator_single_rank(eng):
    """mg_comm rank mode with one rank: unique id, ncclCommInitRank, mg_table_broadcast (the root
    aliases its own table), all-reduce of a u32 buffer -- the call path bench.py takes under
    torchrun, where the id travels through torch.distributed."""
    import torch
    table, nh, lengths = synth.clustered_sketches(64, 200, clusters=2, seed=5)
    t = eng.table_upload(table, nh, lengths)
    comm = abi.RankComm(eng, 1, 0, lambda b: b)
    tb = comm.table_broadcast(t, 64, 200)
    assert np.array_equal(eng.compare_tri_host(tb), eng.compare_tri_host(t))
    buf = torch.arange(1000, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    comm.allreduce_u32_sum(buf.data_ptr(), 1000)
    assert torch.equal(buf.cpu(), torch.arange(1000, dtype=torch.int32))
    tb.free()
    comm.close()
    t.free()


def test_two_host_threads_drive_one_context(eng):
    """entry points are thread-safe per context -- two host threads issue sketch and
    compare calls on ONE mg_ctx at the same time (ctypes releases the GIL); every result equals the
    single-threaded one."""
    import threading
    table, nh, lengths = synth.clustered_sketches(400, 1000, clusters=5, seed=2)
    t = eng.table_upload(table, nh, lengths)
    want_c = eng.compare_tri_host(t)
    rng = np.random.default_rng(4)
    sketches = [synth.adversarial_dna_records(rng, v) for v in (0, 1, 2, 3)] + [[bytes(synth.synthetic_genome(3, 200_000))]]
    p = eng.params(k=21, s=500)
    want_s = eng.sketch_host(sketches, p, counts=True)
    errors = []

    def compares():
        try:
            for _ in range(6):
                assert np.array_equal(eng.compare_tri_host(t), want_c)
                assert np.array_equal(eng.compare_tri_host(t, 100, 300), want_c[abi.tri_pairs(0, 100):abi.tri_pairs(0, 300)])
        except Exception as e:                                     # noqa: BLE001
            errors.append(e)

    def sketching():
        try:
            for _ in range(6):
                got = eng.sketch_host(sketches, p, counts=True)
                assert all(np.array_equal(a, b) for a, b in zip(got, want_s))
        except Exception as e:                                     # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=compares), threading.Thread(target=sketching), threading.Thread(target=compares)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    t.free()


def test_async_compare_calls_are_stream_ordered(eng):
    """mg_ctx_set_async: compare *_dev calls only queue their work (tile lists in the ring of pinned
    slots); several calls back to back -- more than the ring has slots -- then ONE synchronisation;
    every output equals the synchronous one."""
    import torch
    table, nh, lengths = synth.clustered_sketches(600, 1000, clusters=6, seed=13)
    t = eng.table_upload(table, nh, lengths)
    want = eng.compare_tri_host(t)
    nq = 9
    outs = [torch.zeros((abi.tri_pairs(0, 600), 2), dtype=torch.int32, device="cuda") for _ in range(nq)]
    torch.cuda.synchronize()
    eng.set_async(True)
    try:
        for o in outs:
            eng.compare_tri_dev(t, 0, 600, o.data_ptr())
        eng.synchronize()
    finally:
        eng.set_async(False)
    for o in outs:
        got = o.cpu().numpy().view(np.uint32)
        assert np.array_equal(got[:, 0], want["numer"]) and np.array_equal(got[:, 1], want["denom"])
    t.free()


def test_fill_beside_the_build_queued_and_from_two_contexts(eng, monkeypatch):
    """The fill beside the index build in the two settings where its second stream could be forgotten: a context in async
    mode -- per-table jobs of several tables queued back to back, ONE synchronisation of the context's stream at the end --
    and two contexts on one device driven from two threads at once.  Every output equals the one computed with the fill
    switched off."""
    import threading
    import torch
    monkeypatch.setenv("MASHGPU_FILL_ASIDE_MIN_PAIRS", "1")
    n = 3000
    tabs = [synth.clustered_sketches(n, 200, clusters=30, seed=70 + i, pool=300, private=80) for i in range(3)]
    pairs = abi.tri_pairs(0, n)
    monkeypatch.setenv("MASHGPU_FILL_ASIDE", "0")
    want = []
    for table, nh, lengths in tabs:
        t = eng.table_upload(table, nh, lengths)
        want.append(eng.compare_tri_host(t))
        t.free()
    monkeypatch.delenv("MASHGPU_FILL_ASIDE")

    def same(o, w):
        got = o.cpu().numpy().view(np.uint32)
        return np.array_equal(got[:, 0], w["numer"]) and np.array_equal(got[:, 1], w["denom"])

    # queued: three tables, each job cold (its fill beside its build), one wait
    ts = [eng.table_upload(*x) for x in tabs]
    outs = [torch.full((pairs, 2), -1, dtype=torch.int32, device="cuda") for _ in tabs]
    torch.cuda.synchronize()
    eng.prof_enable(True)
    eng.prof_reset()
    eng.set_async(True)
    try:
        for t, o in zip(ts, outs):
            eng.compare_tri_dev(t, 0, n, o.data_ptr())
        eng.synchronize()
    finally:
        eng.set_async(False)
    assert eng.prof_avg_ms("compare_fill_aside")[1] == 3
    eng.prof_enable(False)
    for o, w in zip(outs, want):
        assert same(o, w)
    for t in ts:
        t.free()
    # two contexts, two threads
    errors = []

    def worker(k):
        try:
            e2 = abi.MashGpu(0)
            e2.set_option("MASHGPU_COSTS_FIXED", "1")
            for rnd in range(3):
                i = (k + rnd) % 3
                t2 = e2.table_upload(*tabs[i])
                got = e2.compare_tri_host(t2)
                if not (np.array_equal(got["numer"], want[i]["numer"]) and np.array_equal(got["denom"], want[i]["denom"])):
                    errors.append((k, rnd))
                t2.free()
            e2.close()
        except Exception as ex:                            # noqa: BLE001
            errors.append((k, repr(ex)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors


def test_compare_c3_scale_properties(eng, oracle):
    """BASELINE config 3 shape at a size the oracle can sample: N = 6000 clustered
    s=1000 sketches (1.8e7 pairs).  Checks (a) sampled rows against the oracle,
    (b) checksum-of-checksums between the tiled and generic kernels,
    (c) within/between-cluster structure, (d) denom == s everywhere."""
    import torch
    n = 6000
    table, nhash, lengths = synth.clustered_sketches(n, 1000, clusters=60, seed=21)
    t = eng.table_upload(table, nhash, lengths)
    got = eng.compare_tri_host(t)
    assert len(got) == n * (n - 1) // 2
    assert np.all(got["denom"] == 1000)
    for i in (1, 2, 63, 64, 65, 1023, 1024, 4097, 5999):
        numer, denom = _oracle_tri(oracle, table, nhash, lengths, i, i + 1)
        row = got[i * (i - 1) // 2: i * (i - 1) // 2 + i]
        assert np.array_equal(row["numer"], numer) and np.array_equal(row["denom"], denom), i
    lo = 5000 * 4999 // 2
    for other in ("generic", "merged", "sparse"):
        os.environ["MASHGPU_COMPARE_KERNEL"] = other
        try:
            got_g = eng.compare_tri_host(t, 5000, 5400)
        finally:
            del os.environ["MASHGPU_COMPARE_KERNEL"]
        assert np.array_equal(got_g, got[lo: lo + len(got_g)]), other
    # cluster structure: same cluster <=> i % 60 == j % 60
    i = 4321
    row = got[i * (i - 1) // 2: i * (i - 1) // 2 + i]
    same = (np.arange(i) % 60) == (i % 60)
    assert row["numer"][same].min() > 300 and row["numer"][~same].max() < 50
    t.free()


# ---------------------------------------------------------------- screening

def test_screen_golden(eng, golden_dir):
    """mash screen genomes.msh reads1.fastq reads2.fastq == test/ref/screen, through the ABI:
    device hash table of the 3 golden sketches, every k-mer of the reads probed on the GPU."""
    gh, glens, names = helpers.load_golden_genomes()
    r1 = helpers.read_fastx(os.path.join(golden_dir, "reads1.fastq.gz"))
    r2 = helpers.read_fastx(os.path.join(golden_dir, "reads2.fastq.gz"))
    recs = [r[2] for r in helpers.round_robin([r1, r2]) if len(r[2]) >= 21]
    db = eng.table_upload(gh, np.full(3, 1000, np.uint32), glens)
    p = eng.params(k=21, s=1000)
    half = len(recs) // 2
    counts, mix, distinct = eng.screen(db, p, [recs[:half], recs[half:]])        # two batches
    rh, rlen, _ = helpers.load_golden_reads()
    assert np.array_equal(mix, rh)                                     # mixture bottom-s == sketch of all reads
    set_size = int(2.0 ** 64 * len(mix) / float(mix[-1]))
    assert set_size == rlen
    assert distinct == len(np.unique(gh))
    lines = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(golden_dir, "screen"))]
    for i in range(3):
        shared = int((counts[i] > 0).sum())
        depths = np.sort(counts[i][counts[i] > 0])
        assert "%d/%d" % (shared, 1000) == lines[i][1]
        assert "%g" % eng.lib.mg_identity(shared, 1000, 21) == lines[i][0]
        assert str(int(depths[shared // 2])) == lines[i][2]
        assert "%g" % eng.lib.mg_p_value_within(shared, set_size, KSPACE21, 1000) == lines[i][3]
    db.free()


def test_screen_counts_vs_oracle(eng, oracle):
    """Observation counts of every sketch hash vs a direct count over oracle hashes, incl.
    multi-chunk mixtures, N / lowercase / short records and repeats."""
    rng = np.random.default_rng(8)
    genomes = [synth._rand_dna(rng, 30000) for _ in range(4)]
    p = eng.params(k=21, s=500)
    op = oracle.params(k=21, s=500)
    hashes, nhash = eng.sketch_host([[g] for g in genomes], p)
    db = eng.table_upload(hashes, nhash, np.full(4, 30000, np.uint64))
    reads = []
    for _ in range(3000):
        g = genomes[int(rng.integers(0, 3))]                         # genome 3 is never sampled
        st = int(rng.integers(0, 30000 - 150))
        r = bytearray(g[st:st + 150])
        if rng.random() < 0.3:
            r[int(rng.integers(0, 150))] = ord("N")
        if rng.random() < 0.2:
            r = bytearray(bytes(r).lower())
        reads.append(bytes(r) if rng.random() < 0.5 else bytes(_revcomp(bytes(r).upper())))
    reads += [b"ACGT", b"", synth.adversarial_dna_records(rng, 2)[0]]
    counts, mix, _ = eng.screen(db, p, [reads[:1000], reads[1000:]])
    # expected: hash every valid canonical k-mer of every read on the CPU
    want = {}
    for r in reads:
        if len(r) < 21:
            continue
        h, c, _, _, _ = oracle.sketch_records([r], oracle.params(k=21, s=100000))
        for hv, cv in zip(h, c):
            want[int(hv)] = want.get(int(hv), 0) + int(cv)
    for i in range(4):
        exp = np.array([want.get(int(x), 0) for x in hashes[i, : nhash[i]]], dtype=np.uint32)
        assert np.array_equal(counts[i, : nhash[i]], exp), i
    assert counts[3].sum() <= counts[0].sum()
    allh = np.array(sorted(want), dtype=np.uint64)[:500]
    assert np.array_equal(mix, allh)
    db.free()


def _hits_equal_counts(hits, counts, table, nhash):
    """hits == the non-zero cells of the dense counts matrix, hash values included, ordered by row then hash"""
    r, c = np.nonzero(counts)
    assert len(hits) == len(r)
    want = np.zeros(len(r), dtype=abi.HIT_DTYPE)
    want["row"], want["count"], want["hash"] = r, counts[r, c], table[r, c]
    want = want[np.lexsort((want["hash"], want["row"]))]
    assert np.array_equal(hits, want)


@pytest.mark.parametrize("bits", [None, "12"])
def test_screen_resident_database_sparse_hits_two_tiers(eng, oracle, bits, monkeypatch):
    """A database that stays resident (the reference rebuilds hashTable for every run,
    CommandScreen.cpp:93-116): mixture after mixture against ONE mg_screen with mg_screen_reset between
    them gives what a fresh screen gives; mg_screen_finish_sparse_host returns exactly the non-zero cells
    of the dense matrix (rows that share a hash -- copies of a genome -- each get their hit).  The database
    mixes 2 kbp and 200 kbp genomes, so the key bound takes two tiers (a bitmap in front of the table for
    the hashes only small genomes reach; with 2^12 bits it is crowded and mostly says yes): same counts as
    with MASHGPU_SCREEN_TIERS=0, and as the oracle's direct count."""
    if bits:
        monkeypatch.setenv("MASHGPU_SCREEN_BITS", bits)
    rng = np.random.default_rng(31)
    small = [synth._rand_dna(rng, 2000) for _ in range(12)]
    large = [synth._rand_dna(rng, 200000) for _ in range(5)]
    genomes = large[:3] + small + large[3:] + [small[4], large[1]]           # two copies: rows that share every hash
    p = eng.params(k=21, s=200)
    hashes, nhash = eng.sketch_host([[g] for g in genomes], p)
    db = eng.table_upload(hashes, nhash, np.array([len(g) for g in genomes], dtype=np.uint64))

    def mixture(seed, sources):
        r = np.random.default_rng(seed)
        out = []
        for _ in range(4000):
            g = genomes[int(r.choice(sources))]
            st = int(r.integers(0, len(g) - 120))
            x = g[st:st + 120]
            out.append(x if r.random() < 0.5 else _revcomp(x))
        return [out[:1500], out[1500:]]

    mix_a, mix_b = mixture(1, [0, 3, 7, 5, 16]), mixture(2, [1, 7, 9, 18])
    want = {}
    for name, m in (("a", mix_a), ("b", mix_b)):
        cnt = {}
        for r in (x for part in m for x in part):
            h, c, _, _, _ = oracle.sketch_records([r], oracle.params(k=21, s=100000))
            for hv, cv in zip(h, c):
                cnt[int(hv)] = cnt.get(int(hv), 0) + int(cv)
        want[name] = np.array([[cnt.get(int(x), 0) if j < nhash[i] else 0 for j, x in enumerate(hashes[i])] for i in range(len(genomes))], dtype=np.uint32)
    with eng.screen_open(db, p) as sc:
        assert "two tiers" in sc.tier_note(), sc.tier_note()
        for part in mix_a:
            sc.add_records(part)
        counts_a, sk_a, distinct = sc.finish()
        hits_a, sk_a2, distinct2 = sc.finish_sparse()
        assert np.array_equal(counts_a, want["a"]) and distinct == distinct2 == len(np.unique(hashes[hashes != np.uint64(abi.HASH_PAD)]))
        assert np.array_equal(sk_a, sk_a2)
        _hits_equal_counts(hits_a, counts_a, hashes, nhash)
        assert np.array_equal(counts_a[7], counts_a[17]) and counts_a[7].sum() > 0      # the copy gets the same hits
        sc.reset()
        for part in mix_b:
            sc.add_records(part)
        counts_b, sk_b, _ = sc.finish()
        hits_b, _, _ = sc.finish_sparse()
        assert np.array_equal(counts_b, want["b"])
        _hits_equal_counts(hits_b, counts_b, hashes, nhash)
        sc.reset()
        none, sk0, _ = sc.finish_sparse()                                      # nothing screened: no hit, empty mixture sketch
        assert len(none) == 0 and len(sk0) == 0
    fresh_b, fresh_sk, _ = eng.screen(db, p, mix_b)
    assert np.array_equal(fresh_b, counts_b) and np.array_equal(fresh_sk, sk_b)
    monkeypatch.setenv("MASHGPU_SCREEN_TIERS", "0")
    with eng.screen_open(db, p) as sc:
        assert "off" in sc.tier_note()
        for part in mix_a:
            sc.add_records(part)
        one_tier, _, _ = sc.finish()
    assert np.array_equal(one_tier, counts_a)
    monkeypatch.delenv("MASHGPU_SCREEN_TIERS")
    # the same over three contexts: hits of every device merged on the host, and a second mixture after a reset
    comm = abi.LocalComm([0, 0, 0])
    d = comm.upload(hashes, nhash, np.array([len(g) for g in genomes], dtype=np.uint64))
    sh, ssk, sd = comm.screen(d, len(genomes), 200, p, mix_a, sparse=True)
    assert np.array_equal(sh, hits_a) and np.array_equal(ssk, sk_a) and sd == distinct
    sh2, _, _ = comm.screen(d, len(genomes), 200, p, mix_a, sparse=True, again=mix_b)
    assert np.array_equal(sh2, hits_b)
    comm.free(d)
    comm.close()
    db.free()


def test_screen_config4_scale_vs_oracle(eng, oracle):
    """BASELINE config 4 at its DATABASE scale: 100 000 sketches (10^8 keys in the open-addressing
    table: probe chains, load factor, u32 counters at the size bench.py runs) against 10^6 reads
    sampled as bench.py samples them.  The oracle hashes every k-mer of every read on the host
    (oracle_kmer_hashes: the reference's loop, Sketch.cpp:512-583 / CommandScreen.cpp:533-575); the
    observation counts of 240 database rows (rows the reads come from and rows they do not), the
    mixture's own bottom-s sketch and the number of distinct keys must agree exactly -- through one
    mg_screen and through mg_dscreen on three contexts."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from workloads import synth_torch
    dev = torch.device("cuda", 0)
    S, K, RL, NSRC, GL, NREADS, NDB = 1000, 21, 150, 1000, 1_000_000, 1_000_000, 100_000
    p = eng.params(k=K, s=S)
    op = oracle.params(k=K, s=S)
    genomes = synth_torch.synthetic_genomes(0, NSRC, GL, device=dev, stride=40000)
    gh = torch.empty((NSRC, S), dtype=torch.int64, device=dev)
    gn = torch.empty(NSRC, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng.sketch_dev(genomes.data_ptr(), NSRC * GL, np.arange(NSRC + 1, dtype=np.uint64) * np.uint64(GL), p, gh.data_ptr(), gn.data_ptr())
    fh, fn, _ = synth_torch.clustered_sketch_table(NDB - NSRC, S, clusters=(NDB - NSRC) // 100, device=dev)
    db_h = torch.cat([gh, fh], 0).contiguous()
    db_n = torch.cat([gn, fn], 0).contiguous()
    reads = synth_torch.synthetic_reads(genomes, NREADS, RL, seed=7001)          # [NREADS, RL + 1], separator included
    torch.cuda.synchronize()
    distinct_want = int(torch.unique(db_h.flatten()).numel())                    # every row is full: no padding among them
    assert int(db_n.min()) == S
    del genomes, gh, fh
    th = db_h.cpu().numpy().view(np.uint64)
    tn = db_n.cpu().numpy().astype(np.uint32)
    tl = np.full(NDB, GL, dtype=np.uint64)
    host_reads = reads.cpu().numpy()
    del reads, db_h, db_n
    torch.cuda.empty_cache()
    blobs = [np.ascontiguousarray(b).reshape(-1) for b in np.array_split(host_reads, 4)]
    db = eng.table_upload(th, tn, tl)
    counts, mix, distinct = eng.screen(db, p, blobs)
    # ---- the oracle's side
    rng = np.random.default_rng(4)
    rows = np.unique(np.concatenate([rng.choice(NSRC, 120, replace=False), NSRC + rng.choice(NDB - NSRC, 117, replace=False),
                                     [0, NSRC - 1, NDB - 1]]))
    keys = np.unique(th[rows].reshape(-1))

    def chunk(part):
        bases = np.ascontiguousarray(part).reshape(-1).copy()
        off = np.arange(len(part) + 1, dtype=np.uint64) * np.uint64(RL + 1)      # the separator ends no k-mer: not in the alphabet
        H = oracle.kmer_hashes(bases, off, op)
        idx = np.searchsorted(keys, H)
        idx[idx == len(keys)] = 0
        hit = keys[idx] == H
        return np.bincount(idx[hit], minlength=len(keys)), np.unique(H)[:S], len(H)

    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        parts = list(ex.map(chunk, np.array_split(host_reads, 40)))
    cnt = np.sum([q[0] for q in parts], axis=0)
    assert sum(q[2] for q in parts) >= NREADS * (RL - K + 1) * 0.99              # every read gave its k-mers
    want_mix = np.unique(np.concatenate([q[1] for q in parts]))[:S]
    for i in rows:
        exp = cnt[np.searchsorted(keys, th[i])].astype(np.uint32)
        assert np.array_equal(counts[i], exp), i
    assert counts[rows[rows < NSRC]].sum() > 100 * counts[rows[rows >= NSRC]].sum()   # the reads' genomes are seen, the others are not
    assert np.array_equal(mix, want_mix)
    assert distinct == distinct_want
    # ---- the same through mg_dscreen on three contexts of this device
    comm = abi.LocalComm([0, 0, 0])
    d = comm.upload(th, tn, tl)
    got = comm.screen(d, NDB, S, p, blobs)
    assert np.array_equal(got[0], counts) and np.array_equal(got[1], mix) and got[2] == distinct
    # ---- and in the sparse form (what the CLI reads): the non-zero cells, nothing else; a second mixture after a reset
    hits, hmix, hdist = comm.screen(d, NDB, S, p, blobs, sparse=True)
    _hits_equal_counts(hits, counts, th, tn)
    assert np.array_equal(hmix, mix) and hdist == distinct and len(hits) < 0.02 * NDB * S
    hits2, _, _ = comm.screen(d, NDB, S, p, blobs[:1], sparse=True, again=blobs)
    assert np.array_equal(hits2, hits)
    comm.free(d)
    comm.close()
    db.free()


@pytest.mark.parametrize("k,nonc", [(11, False), (16, True), (27, False), (32, False), (5, True)])
def test_screen_other_kmer_sizes(eng, oracle, k, nonc):
    """the fused sketch+probe instantiations at other k-mer sizes / forward-only k-mers"""
    rng = np.random.default_rng(40 + k)
    genomes = [synth._rand_dna(rng, 6000) for _ in range(3)]
    s = 200
    p = eng.params(k=k, s=s, noncanonical=nonc)
    hashes, nhash = eng.sketch_host([[g] for g in genomes], p)
    db = eng.table_upload(hashes, nhash, np.full(3, 6000, np.uint64))
    reads = []
    for _ in range(800):
        g = genomes[int(rng.integers(0, 2))]
        st = int(rng.integers(0, 6000 - 120))
        r = g[st:st + int(rng.integers(k, 120))]
        reads.append(r if (nonc or rng.random() < 0.5) else _revcomp(r))
    counts, mix, _ = eng.screen(db, p, [reads[:300], reads[300:]])
    want = {}
    op_all = oracle.params(k=k, s=10 ** 6, noncanonical=nonc)
    for r in reads:
        if len(r) < k:
            continue
        h, c, _, _, _ = oracle.sketch_records([r], op_all)
        for hv, cv in zip(h, c):
            want[int(hv)] = want.get(int(hv), 0) + int(cv)
    for i in range(3):
        exp = np.array([want.get(int(x), 0) for x in hashes[i, : nhash[i]]], dtype=np.uint32)
        assert np.array_equal(counts[i, : nhash[i]], exp), (k, i)
    assert np.array_equal(mix, np.array(sorted(want), dtype=np.uint64)[:s])
    db.free()


def test_screen_translated_vs_oracle(eng, oracle):
    """Amino-acid query sketches against a nucleotide mixture: the device translates every batch
    in six frames (CommandScreen.cpp:516-531, 617-809); expected counts come from the oracle's
    restated translate() + direct hashing of every amino-acid k-mer of every frame of every read."""
    rng = np.random.default_rng(31)
    prot = "ACDEFGHIKLMNPQRSTVWY"
    k, s = 7, 300
    genomes = [synth._rand_dna(rng, 9000) for _ in range(3)]
    p = eng.params(k=k, s=s, alphabet=prot, noncanonical=True)
    op_all = oracle.params(k=k, s=10 ** 6, alphabet=prot, noncanonical=True)
    # queries: protein sketches of the six-frame translations of each genome
    hashes, nhash = eng.sketch_host([oracle.six_frames(g) for g in genomes], p)
    for i, g in enumerate(genomes):
        oh = oracle.sketch_records(oracle.six_frames(g), oracle.params(k=k, s=s, alphabet=prot, noncanonical=True))[0]
        assert nhash[i] == len(oh) and np.array_equal(hashes[i, : len(oh)], oh)
    db = eng.table_upload(hashes, nhash, np.full(3, 9000, np.uint64))
    reads = []
    for _ in range(1500):
        g = genomes[int(rng.integers(0, 2))]                         # genome 2 is never sampled
        l = int(rng.integers(30, 160))
        st = int(rng.integers(0, 9000 - l))
        r = bytearray(g[st:st + l])
        u = rng.random()
        if u < 0.2:
            r[int(rng.integers(0, l))] = ord("N")
        elif u < 0.35:
            r = bytearray(bytes(r).lower())
        reads.append(bytes(r) if rng.random() < 0.5 else _revcomp(bytes(r).upper()))
    reads += [b"AC", b"ACGTACGTAC", b""]
    with eng.screen_open(db, p, translate=True) as sc:
        sc.add_records(reads[:700])
        sc.add_records(reads[700:])
        counts, mix, _ = sc.finish()
    want = {}
    for r in reads:
        for fr in oracle.six_frames(r):
            if len(fr) < k:
                continue
            h, c, _, _, _ = oracle.sketch_records([fr], op_all)
            for hv, cv in zip(h, c):
                want[int(hv)] = want.get(int(hv), 0) + int(cv)
    for i in range(3):
        exp = np.array([want.get(int(x), 0) for x in hashes[i, : nhash[i]]], dtype=np.uint32)
        assert np.array_equal(counts[i, : nhash[i]], exp), i
    assert counts[0].sum() > 0 and counts[2].sum() < counts[0].sum()
    assert np.array_equal(mix, np.array(sorted(want), dtype=np.uint64)[:s])
    db.free()


def test_screen_sharded_orchestrator_single_rank(eng, oracle):
    """mash_amd.screen_dist over libmashgpu (world 1): device-resident counts, host and
    device batches, same numbers as mg_screen_finish_host."""
    import torch
    from mash_amd import screen_dist
    rng = np.random.default_rng(12)
    genomes = [synth._rand_dna(rng, 20000) for _ in range(3)]
    p = eng.params(k=21, s=300)
    hashes, nhash = eng.sketch_host([[g] for g in genomes], p)
    db = eng.table_upload(hashes, nhash, np.full(3, 20000, np.uint64))
    reads = [genomes[i % 2][st:st + 120] for i, st in enumerate(rng.integers(0, 20000 - 120, 4000))]
    batches = [reads[i:i + 500] for i in range(0, 4000, 500)]
    # one batch handed over as device memory
    from mash_amd.abi import join_records
    blob = torch.from_numpy(np.frombuffer(join_records(batches[3]), dtype=np.uint8).copy()).cuda()
    mixed = list(batches)
    mixed[3] = (blob.data_ptr(), blob.numel(), blob)
    counts, mix = screen_dist.screen_sharded(screen_dist.gpu_local_screen(eng, db, p), mixed, 300)
    want_counts, want_mix, _ = eng.screen(db, p, batches)
    assert np.array_equal(counts.cpu().numpy().astype(np.uint32).reshape(3, 300), want_counts)
    assert np.array_equal(mix, want_mix)
    assert want_counts[2].sum() < want_counts[0].sum()
    db.free()


def _revcomp(b):
    return bytes({65: 84, 67: 71, 71: 67, 84: 65, 78: 78}[x] for x in reversed(b))


# ---------------------------------------------------------------- BASELINE-size runs (device-resident)

def test_c2_scale_sketch_properties(eng, oracle, monkeypatch):
    """BASELINE config 2 at 1/10 scale, device resident: 1000 synthetic 1 Mbp genomes (10^9 bases)
    sketched in one call.  (a) every sketch is full, ascending, distinct; (b) sampled genomes equal
    the oracle; (c) a different work decomposition (forced multi-chunk + merge kernel) gives the
    identical table; (d) multiplicities sum to the number of k-mers for a repeat-free genome's keys."""
    import torch
    from workloads import synth_torch
    ng, L = 1000, 1_000_000
    dev = torch.device("cuda", 0)
    bases = synth_torch.synthetic_genomes(0, ng, L, device=dev)
    off = np.arange(ng + 1, dtype=np.uint64) * np.uint64(L)
    p = eng.params(k=21, s=1000)
    out = torch.empty((ng, 1000), dtype=torch.int64, device=dev)
    nh = torch.empty(ng, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng.sketch_dev(bases.data_ptr(), ng * L, off, p, out.data_ptr(), nh.data_ptr())
    eng.synchronize()
    assert int(nh.min()) == 1000 and int(nh.max()) == 1000
    assert bool((out[:, 1:] > out[:, :-1]).all())                    # values < 2^63 here: signed compare is fine
    host = out.cpu().numpy().view(np.uint64)
    op = oracle.params(k=21, s=1000)
    for g in (0, 1, 499, 999):
        h, _, _, _, _ = oracle.sketch_records([bytes(synth.synthetic_genome(g, L))], op)
        assert np.array_equal(host[g], h), g
    monkeypatch.setenv("MASHGPU_SKETCH_MIN_CHUNK", "61440")
    monkeypatch.setenv("MASHGPU_SKETCH_ITEMS", "1000000")
    out2 = torch.empty_like(out)
    nh2 = torch.empty_like(nh)
    eng.sketch_dev(bases.data_ptr(), ng * L, off, p, out2.data_ptr(), nh2.data_ptr())
    eng.synchronize()
    assert torch.equal(out, out2) and torch.equal(nh, nh2)


def test_c3_scale_triangle_properties(eng, oracle):
    """BASELINE config 3 at N = 40 000 (8.0e8 pairs, 6.4 GB of results resident in HBM):
    (a) denom == s and numer <= s everywhere; (b) sampled rows equal the oracle;
    (c) a row block recomputed with the per-row tiled kernel and with the generic kernel is
    identical; (d) cluster structure of the synthetic table."""
    import torch
    from workloads import synth_torch
    n, s = 40000, 1000
    dev = torch.device("cuda", 0)
    hashes, nhash, lengths = synth_torch.clustered_sketch_table(n, s, clusters=400, device=dev)
    table = eng.table_wrap(hashes.data_ptr(), nhash.data_ptr(), lengths.data_ptr(), n, s)
    out = torch.empty((n * (n - 1) // 2, 2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng.compare_tri_dev(table, 0, n, out.data_ptr())
    eng.synchronize()
    assert int(out[:, 1].min()) == s and int(out[:, 1].max()) == s and int(out[:, 0].max()) <= s
    th = hashes.cpu().numpy().view(np.uint64)
    tn = nhash.cpu().numpy().astype(np.uint32)
    tl = lengths.cpu().numpy().astype(np.uint64)
    for i in (1, 17, 4096, 20001, 39999):
        numer, denom = _oracle_tri(oracle, th, tn, tl, i, i + 1)
        row = out[i * (i - 1) // 2: i * (i - 1) // 2 + i].cpu().numpy()
        assert np.array_equal(row[:, 0], numer) and np.array_equal(row[:, 1], denom), i
    lo, hi = 30000, 30200
    base = lo * (lo - 1) // 2
    npairs = hi * (hi - 1) // 2 - base
    ref_block = out[base: base + npairs].cpu()
    for other in ("merged", "sparse", "generic"):
        os.environ["MASHGPU_COMPARE_KERNEL"] = other
        try:
            o2 = torch.empty((npairs, 2), dtype=torch.int32, device=dev)
            eng.compare_tri_dev(table, lo, hi, o2.data_ptr())
            eng.synchronize()
        finally:
            del os.environ["MASHGPU_COMPARE_KERNEL"]
        assert torch.equal(o2.cpu(), ref_block), other
    i = 33333
    row = out[i * (i - 1) // 2: i * (i - 1) // 2 + i, 0].cpu().numpy()
    same = (np.arange(i) % 400) == (i % 400)
    assert row[same].min() > 300 and row[~same].max() < 50
    table.free()


@pytest.mark.parametrize("contiguous", [False, True])
def test_c3_scale_filter_and_cluster_layouts(eng, oracle, contiguous):
    """Thresholded all-pairs at N = 47 000 (1.1e9 pairs: two device row blocks) against the full
    count matrix thresholded with torch; clusters interleaved or laid out as runs of
    consecutive rows (every tile then holds 16 related rows: distinct-first buckets, single
    representative verification), sampled rows against the oracle."""
    import torch
    from workloads import synth_torch
    n, s, k = 47000, 1000, 21
    dev = torch.device("cuda", 0)
    hashes, nhash, lengths = synth_torch.clustered_sketch_table(n, s, clusters=470, device=dev, contiguous=contiguous)
    table = eng.table_wrap(hashes.data_ptr(), nhash.data_ptr(), lengths.data_ptr(), n, s)
    out = torch.empty((n * (n - 1) // 2, 2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng.compare_tri_dev(table, 0, n, out.data_ptr())
    eng.synchronize()
    th = hashes.cpu().numpy().view(np.uint64)
    tn = nhash.cpu().numpy().astype(np.uint32)
    tl = lengths.cpu().numpy().astype(np.uint64)
    for i in (3, 1234, 23456, 46999):
        numer, denom = _oracle_tri(oracle, th, tn, tl, i, i + 1)
        row = out[i * (i - 1) // 2: i * (i - 1) // 2 + i].cpu().numpy()
        assert np.array_equal(row[:, 0], numer) and np.array_equal(row[:, 1], denom), i
    max_d = 0.05
    t_min = min(x for x in range(s + 1) if _py_distance(x, s, k) <= max_d)
    assert int(out[:, 1].min()) == s
    keep = torch.nonzero(out[:, 0] >= t_min).squeeze(1)
    edges = eng.compare_tri_filter(table, k, max_d, capacity=1 << 24)
    assert len(edges) == int(keep.numel()) > 100000
    flat = (edges["row"].astype(np.int64) * (edges["row"].astype(np.int64) - 1)) // 2 + edges["col"].astype(np.int64)
    assert np.array_equal(flat, keep.cpu().numpy())                       # same pairs, reference order
    assert np.array_equal(edges["numer"], out[keep, 0].cpu().numpy().astype(np.uint32))
    table.free()


def _check_full_triangle_against_reference(out, th, tn, tl, n, k, clusters, nclu_checked, ncross_rows):
    """`out`: device tensor [pairs, 2] of a full triangle over the table (th, tn, tl).  Every pair
    inside `nclu_checked` of the interleaved clusters (the pairs that share hashes) and every pair
    among `ncross_rows` random rows (cross-cluster pairs, for all practical purposes) against the
    REFERENCE's own compareSketches where oracle/_ref travelled with the snapshot (else the C
    restatement), on all host threads."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle
    orc = pyoracle.Oracle(ref=pyoracle.ref_available())
    kspace = 4.0 ** k
    rng = np.random.default_rng(12345)
    groups = [np.arange(c, n, clusters) for c in rng.choice(clusters, size=nclu_checked, replace=False)]
    groups.append(np.sort(rng.choice(n, size=ncross_rows, replace=False)))

    def check(rows):
        sub = (np.ascontiguousarray(th[rows]), np.ascontiguousarray(tn[rows]), np.ascontiguousarray(tl[rows]))
        numer, denom, _, _ = orc.triangle(sub[0], sub[1], sub[2], 0, len(rows), k, kspace)
        x, y = np.tril_indices(len(rows), -1)                          # x > y, row-major: the triangle's order
        gi, gj = rows[x].astype(np.int64), rows[y].astype(np.int64)
        return gi * (gi - 1) // 2 + gj, numer, denom

    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:        # ctypes releases the GIL
        parts = list(ex.map(check, groups))
    idx = np.concatenate([p[0] for p in parts])
    numer = np.concatenate([p[1] for p in parts])
    denom = np.concatenate([p[2] for p in parts])
    got = out[torch.from_numpy(idx).to(out.device)].cpu().numpy()
    assert np.array_equal(got[:, 0].astype(np.uint32), numer) and np.array_equal(got[:, 1].astype(np.uint32), denom)
    return len(idx), int(np.count_nonzero(numer))


def _sums(out):
    import torch
    return (int(out[:, 0].sum(dtype=torch.int64).item()), int(out[:, 1].sum(dtype=torch.int64).item()))


def test_c3_full_size_triangle(eng, oracle, monkeypatch):
    """BASELINE config 3 at its full size: 100 000 sketches, 4.99995e9 pairs, 40 GB of counts resident
    in HBM.  This is the run bench.py's checksum constant comes from (workloads/checksums.py): ALL
    4.95e6 within-cluster pairs and 1.1e6 cross-cluster pairs of the output against the reference's
    compareSketches (CommandDistance.cpp:336-425), sampled whole rows against the oracle, denom == s
    everywhere, the sums of the default engine, of the inverted-index engine and of the tile engine
    equal to each other and to the constant, the generic kernel's sums on the first 20 000 rows."""
    import torch
    from workloads import synth_torch
    from workloads.checksums import C3_CHECKSUM
    n, s = 100000, 1000
    dev = torch.device("cuda", 0)
    hashes, nhash, lengths = synth_torch.clustered_sketch_table(n, s, clusters=1000, device=dev)
    table = eng.table_wrap(hashes.data_ptr(), nhash.data_ptr(), lengths.data_ptr(), n, s)
    out = torch.empty((n * (n - 1) // 2, 2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng.compare_tri_dev(table, 0, n, out.data_ptr())
    eng.synchronize()
    th = hashes.cpu().numpy().view(np.uint64)
    tn = nhash.cpu().numpy().astype(np.uint32)
    tl = lengths.cpu().numpy().astype(np.uint64)
    for i in (1, 16384, 65537, 99999):
        numer, denom = _oracle_tri(oracle, th, tn, tl, i, i + 1)
        row = out[i * (i - 1) // 2: i * (i - 1) // 2 + i].cpu().numpy()
        assert np.array_equal(row[:, 0], numer) and np.array_equal(row[:, 1], denom), i
    step = 1 << 28
    for o in range(0, out.shape[0], step):                           # denom == s, numer <= s, in slices
        part = out[o:o + step]
        assert int(part[:, 1].min()) == s and int(part[:, 1].max()) == s and int(part[:, 0].max()) <= s
    i = 77777
    row = out[i * (i - 1) // 2: i * (i - 1) // 2 + i, 0].cpu().numpy()
    same = (np.arange(i) % 1000) == (i % 1000)
    assert row[same].min() > 300 and row[~same].max() < 50
    checked, shared = _check_full_triangle_against_reference(out, th, tn, tl, n, 21, 1000, 1000, 1500)
    assert checked >= 4950000 + 1100000 and shared >= 4900000
    want = C3_CHECKSUM[(n, s)]
    assert _sums(out) == want
    for kernel in ("sparse", "merged"):                                # the two engines, each forced
        monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", kernel)
        out.zero_()
        eng.compare_tri_dev(table, 0, n, out.data_ptr())
        eng.synchronize()
        assert _sums(out) == want, kernel
    m = 20000
    mp = m * (m - 1) // 2
    ref_sums = _sums(out[:mp])
    monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", "generic")            # one wave per pair, binary search: shares no code with either
    out[:mp].zero_()
    eng.compare_tri_dev(table, 0, m, out.data_ptr())
    eng.synchronize()
    assert _sums(out[:mp]) == ref_sums
    monkeypatch.delenv("MASHGPU_COMPARE_KERNEL")
    del out
    table.free()


def test_c5_full_size_triangle(eng, oracle, monkeypatch):
    """BASELINE config 5 at config-3 scale (100 000 sketches of s = 10 000, 8 GB table): 100 whole
    clusters (4.95e5 pairs sharing thousands of hashes) and 2e5 cross-cluster pairs against the
    reference's compareSketches, both engines' sums equal to the constant bench.py asserts."""
    import torch
    from workloads import synth_torch
    from workloads.checksums import C3_CHECKSUM
    n, s = 100000, 10000
    dev = torch.device("cuda", 0)
    hashes, nhash, lengths = synth_torch.clustered_sketch_table(n, s, clusters=1000, pool=15000, private=4000, device=dev, block=2000)
    table = eng.table_wrap(hashes.data_ptr(), nhash.data_ptr(), lengths.data_ptr(), n, s)
    out = torch.empty((n * (n - 1) // 2, 2), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng.compare_tri_dev(table, 0, n, out.data_ptr())
    eng.synchronize()
    want = C3_CHECKSUM[(n, s)]
    assert _sums(out) == want
    rows = np.concatenate([np.arange(c, n, 1000) for c in range(0, 1000, 10)] + [np.arange(0, n, 157)])
    rows = np.unique(rows)
    # only the rows used travel to the host (the table is 8 GB)
    sub_h = hashes[torch.from_numpy(rows).to(dev)].cpu().numpy().view(np.uint64)
    sub_n = nhash[torch.from_numpy(rows).to(dev)].cpu().numpy().astype(np.uint32)
    sub_l = lengths[torch.from_numpy(rows).to(dev)].cpu().numpy().astype(np.uint64)
    # a compact table of the rows used, addressed by their rank; the helper maps back through `rows`
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle
    orc = pyoracle.Oracle(ref=pyoracle.ref_available())
    pos = {int(r): k for k, r in enumerate(rows)}
    groups = [np.arange(c, n, 1000) for c in range(0, 1000, 10)] + [np.arange(0, n, 157)]

    def check(g):
        sel = np.array([pos[int(r)] for r in g])
        numer, denom, _, _ = orc.triangle(np.ascontiguousarray(sub_h[sel]), np.ascontiguousarray(sub_n[sel]),
                                          np.ascontiguousarray(sub_l[sel]), 0, len(g), 31, 4.0 ** 31)
        x, y = np.tril_indices(len(g), -1)
        gi, gj = g[x].astype(np.int64), g[y].astype(np.int64)
        return gi * (gi - 1) // 2 + gj, numer, denom

    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        parts = list(ex.map(check, groups))
    idx = np.concatenate([p[0] for p in parts])
    numer = np.concatenate([p[1] for p in parts])
    denom = np.concatenate([p[2] for p in parts])
    got = out[torch.from_numpy(idx).to(dev)].cpu().numpy()
    assert np.array_equal(got[:, 0].astype(np.uint32), numer) and np.array_equal(got[:, 1].astype(np.uint32), denom)
    assert len(idx) >= 495000 + 200000 and int(numer.max()) > 3000
    for kernel in ("sparse", "merged"):
        monkeypatch.setenv("MASHGPU_COMPARE_KERNEL", kernel)
        out.zero_()
        eng.compare_tri_dev(table, 0, n, out.data_ptr())
        eng.synchronize()
        assert _sums(out) == want, kernel
    monkeypatch.delenv("MASHGPU_COMPARE_KERNEL")
    del out
    table.free()

