"""Pin the CPU oracle against every golden vector the reference holds for the
hot path, and against outputs of the reference's own objects
(oracle/_ref) generated in the build container (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from tests import helpers

K, S = 21, 1000
KSPACE = 4.0 ** 21


def test_murmur_known_vectors(oracle):
    # getHash over ASCII k-mer text, seed 42 (hash.cpp:10-38); values cross-checked
    # against the reference objects when available (test below) — here: stability
    # of 64- vs 32-bit views
    h64 = oracle.get_hash(b"ACGTACGTACGTACGTACGTA", 42, True)
    h32 = oracle.get_hash(b"ACGTACGTACGTACGTACGTA", 42, False)
    assert h32 == (h64 & 0xFFFFFFFF)


def test_reads_json_golden(oracle, golden_dir):
    """mash sketch -r -I reads reads1.fastq reads2.fastq == test/ref/reads.json."""
    r1 = helpers.read_fastx(os.path.join(golden_dir, "reads1.fastq.gz"))
    r2 = helpers.read_fastx(os.path.join(golden_dir, "reads2.fastq.gz"))
    recs = helpers.round_robin([r1, r2])
    assert len(recs) == 2000
    p = oracle.params(k=K, s=S, seed=42)
    h, c, length, setsz, rc = oracle.sketch_records([r[2] for r in recs], p)
    gh, glen, gcomment = helpers.load_golden_reads()
    assert rc == 0
    assert np.array_equal(h, gh)
    assert int(setsz) == glen == 502359          # reads: length = estimateSetSize (Sketch.cpp:1272-1282)
    first = recs[0]
    comment = "[%d seqs] %s %s [...]" % (len(recs), first[0].decode(), first[1].decode())
    assert comment == gcomment


def _dist_lines(golden_dir):
    out = []
    for ln in open(os.path.join(golden_dir, "genomes.dist")):
        f = ln.rstrip("\n").split("\t")
        out.append((f[0], f[1], f[2], f[3], f[4]))
    return out


def test_genomes_dist_golden(oracle, golden_dir):
    """mash dist genomes.msh reads.msh == test/ref/genomes.dist (3 lines)."""
    gh, glens, names = helpers.load_golden_genomes()
    rh, rlen, _ = helpers.load_golden_reads()
    lines = _dist_lines(golden_dir)
    for i in range(3):
        out = oracle.compare(gh[i], rh, int(glens[i]), rlen, S, K, KSPACE)
        assert out.pass_ == 1
        assert "%d/%d" % (out.numer, out.denom) == lines[i][4]
        assert helpers.fmt_g(out.distance) == lines[i][2]
        assert helpers.fmt_g(out.p_value) == lines[i][3]
        assert names[i] == lines[i][0]


def test_tutorial_known_answers(oracle):
    """doc/sphinx/tutorials.rst:24,56-57: 456/1000 -> 0.0222766 ; 1000/1000 -> 0."""
    a = np.arange(0, 2000, 2, dtype=np.uint64)          # 1000 values
    b = a.copy()
    out = oracle.compare(a, b, 4639675, 4631469, S, K, KSPACE)
    assert (out.numer, out.denom) == (1000, 1000)
    assert out.distance == 0.0 and helpers.fmt_g(out.p_value) == "0"
    # construct a pair whose bottom-1000 of the union holds exactly 456 shared
    shared = np.arange(0, 456, dtype=np.uint64) * 4
    only_a = np.arange(0, 272, dtype=np.uint64) * 4 + 1
    only_b = np.arange(0, 272, dtype=np.uint64) * 4 + 2
    tail_a = np.arange(10**6, 10**6 + 272, dtype=np.uint64) * 4 + 1
    tail_b = np.arange(10**6, 10**6 + 272, dtype=np.uint64) * 4 + 2
    A = np.sort(np.concatenate([shared, only_a, tail_a]))
    B = np.sort(np.concatenate([shared, only_b, tail_b]))
    assert len(A) == 1000 and len(B) == 1000
    out = oracle.compare(A, B, 4639675, 5498450, S, K, KSPACE)
    assert (out.numer, out.denom) == (456, 1000)
    assert helpers.fmt_g(out.distance) == "0.0222766"
    assert helpers.fmt_g(out.p_value) == "0"


def test_binomial_tail_vs_scipy_fixtures(oracle, golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "binom_sf.json")))
    worst = 0.0
    for c in cases:
        got = oracle.binomial_q(c["x"] - 1, c["r"], c["n"])
        want = c["sf"]
        if want < 1e-290:
            assert got < 1e-280
            continue
        rel = abs(got - want) / want
        worst = max(worst, rel)
        assert rel < 1e-9, (c, got)
    assert worst < 1e-9


def test_oracle_equals_reference_sketch_vectors(oracle):
    """The restatement reproduces hash lists AND counts the reference's own objects
    produced (tests/golden/ref_sketch_vectors.npz)."""
    for cfg, recs, gh, gc in helpers.load_ref_sketch_vectors():
        p = oracle.params(k=cfg["k"], s=cfg["s"], alphabet=cfg["alphabet"],
                          noncanonical=cfg["noncanonical"], preserve_case=cfg["preserve_case"])
        h, c, length, setsz, rc = oracle.sketch_records(recs, p)
        assert rc == cfg["rc"]
        assert length == cfg["length"]
        assert np.array_equal(h, gh), cfg
        assert np.array_equal(c, gc), cfg
        assert setsz == cfg["set_size"]


def test_oracle_min_copies_equals_reference_vectors(oracle):
    """`-m`: the restated pending-set logic (MinHashHeap.cpp:96-118, :126-144) reproduces what the
    reference's MinHashHeap produced, and the result has the order-independent form the GPU path
    relies on: the s smallest hashes seen >= m times, true multiplicities except the largest."""
    for cfg, recs, gh, gc in helpers.load_ref_sketch_vectors("ref_sketch_vectors_m.npz"):
        m = cfg["min_copies"]
        p = oracle.params(k=cfg["k"], s=cfg["s"], min_copies=m)
        h, c, length, setsz, rc = oracle.sketch_records(recs, p)
        assert rc == cfg["rc"] and length == cfg["length"] and setsz == cfg["set_size"]
        assert np.array_equal(h, gh) and np.array_equal(c, gc), cfg
        fh, fc, _, _, _ = oracle.sketch_records(recs, oracle.params(k=cfg["k"], s=10 ** 6))
        keep = fc >= m
        assert np.array_equal(fh[keep][: cfg["s"]], gh), cfg
        diff = np.nonzero(fc[keep][: cfg["s"]] != gc)[0]
        assert len(diff) == 0 or (len(diff) == 1 and diff[0] == len(gh) - 1 and len(gh) == cfg["s"]), cfg


def test_oracle_target_coverage_equals_reference_vectors(oracle):
    """`-c`: the restated record loop stops after the same read as the reference's (reads used),
    with the same hashes and counts (tests/golden/ref_sketch_vectors_c.npz)."""
    for cfg, recs, gh, gc in helpers.load_ref_sketch_vectors("ref_sketch_vectors_c.npz"):
        p = oracle.params(k=cfg["k"], s=cfg["s"], min_copies=cfg["min_copies"], target_cov=cfg["target_cov"])
        h, c, setsz, used, mult = oracle.sketch_reads(recs, p)
        assert used == cfg["used"] and setsz == cfg["set_size"] and mult == cfg["mult"], cfg
        assert np.array_equal(h, gh) and np.array_equal(c, gc), cfg
    assert any(c["used"] < c["nrec"] - 1 for c, _, _, _ in helpers.load_ref_sketch_vectors("ref_sketch_vectors_c.npz"))


def test_oracle_bloom_equals_reference_vectors(oracle):
    """`-b`: the restated Bloom filter (one hash function over bytes * 8 bits, hash_ap, salt) in
    front of the restated heap == the reference's MinHashHeap with the vendored bloom_filter.hpp,
    run here (tests/golden/ref_sketch_vectors_b.npz): hashes, counts, set size, multiplicity, reads
    used (one case carries -c).  The vectors include filters small enough that aliases decide
    which hashes are kept: an exact -m 2 filter gives a different sketch there."""
    differs = 0
    for cfg, recs, gh, gc in helpers.load_ref_sketch_vectors("ref_sketch_vectors_b.npz"):
        p = oracle.params(k=cfg["k"], s=cfg["s"], target_cov=cfg["target_cov"], bloom_bytes=cfg["bloom_bytes"])
        h, c, setsz, used, mult = oracle.sketch_reads(recs, p)
        assert used == cfg["used"] and setsz == cfg["set_size"] and mult == cfg["mult"], cfg
        assert np.array_equal(h, gh) and np.array_equal(c, gc), cfg
        assert c.min() >= 2, cfg                                        # a hash is kept with count 2 (MinHashHeap.cpp:85)
        h2, _, _, _, _ = oracle.sketch_reads(recs, oracle.params(k=cfg["k"], s=cfg["s"], min_copies=2, target_cov=cfg["target_cov"]))
        differs += not np.array_equal(h2, gh)
    assert differs >= 2


def test_oracle_equals_reference_objects_on_fuzz_cases():
    """The oracle against the reference's own addMinHashes / MinHashHeap / bloom_filter (oracle/_ref,
    built here from /root/reference) on the cases tests/fuzz_sketch.py generates: random k, sketch
    sizes, seeds, alphabets, dirty and low-complexity records, multiplicities, -m, -c, -b.  Skipped
    where the reference objects are not built (they are not on the GPU box's CPU-only path either)."""
    from oracle import pyoracle
    if not pyoracle.ref_available():
        pytest.skip("oracle/_ref/libmash_ref.so not built here")
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_sketch", os.path.join(os.path.dirname(__file__), "fuzz_sketch.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    assert fz.run(250, 11, 60, against="ref", quiet=True) == 0


def test_oracle_compare_equals_reference_objects_on_random_pairs():
    """compareSketches: the restated merge + distance + filters against the reference's own function
    (CommandDistance.cpp:336-425 compiled into oracle/_ref) on random pairs built for the corners: shared
    pools (every Jaccard value), one side empty or shorter than s, sketch sizes 1..5000, 32-bit hashes,
    identical lists, disjoint ranges, both filters on and off.  (The p-value goes through the same
    restated binomial tail on both sides -- GSL is absent -- so it is not an independent check here.)"""
    from oracle import pyoracle
    if not pyoracle.ref_available():
        pytest.skip("oracle/_ref/libmash_ref.so not built here")
    orc, ref = pyoracle.Oracle(), pyoracle.Oracle(ref=True)
    rng = np.random.default_rng(2026)
    checked = passed = 0
    for case in range(3000):
        s = int(rng.choice([1, 2, 5, 50, 400, 1000, 5000]))
        use64 = bool(rng.random() < 0.7)
        k = int(rng.choice([21, 31, 32])) if use64 else int(rng.choice([5, 11, 16]))
        top = 2 ** 64 - 2 if use64 else 2 ** 32 - 2
        style = rng.choice(["pool", "disjoint", "identical", "ragged", "empty"])
        pool = np.unique(rng.integers(0, top, int(s * rng.uniform(1.0, 3.0)) + 2, dtype=np.uint64))
        def draw(frac):
            v = pool[rng.random(len(pool)) < frac]
            return np.sort(v)[:s]
        if style == "pool":
            a, b = draw(rng.uniform(0.3, 1.0)), draw(rng.uniform(0.3, 1.0))
        elif style == "disjoint":
            a, b = pool[: len(pool) // 2][:s], pool[len(pool) // 2:][:s]
        elif style == "identical":
            a = draw(0.9)
            b = a.copy()
        elif style == "ragged":
            a, b = draw(0.8)[: int(rng.integers(0, s + 1))], draw(0.8)
        else:
            a, b = np.zeros(0, dtype=np.uint64), draw(0.7)
        if rng.random() < 0.5:
            a, b = b, a
        la, lb = int(rng.integers(1, 10 ** 7)), int(rng.integers(1, 10 ** 7))
        kspace = float(4 ** k)
        max_d = float(rng.choice([-1.0, 0.02, 0.2, 1.0]))
        max_p = float(rng.choice([-1.0, 1e-30, 1e-3, 1.0]))
        x = orc.compare(a, b, la, lb, s, k, kspace, max_d, max_p, use64)
        y = ref.compare(a, b, la, lb, s, k, kspace, max_d, max_p, use64)
        assert x.pass_ == y.pass_, (case, style, s, k, max_d, max_p)
        if x.pass_:                                              # a rejected pair carries no more than `pass` (CommandDistance.cpp:409-422)
            assert (x.numer, x.denom) == (y.numer, y.denom), (case, style, s, k)
            assert x.distance == y.distance and x.p_value == y.p_value, (case, style, s, k, x.distance, y.distance)
            passed += 1
        checked += 1
    assert checked == 3000 and passed > 1000


def test_oracle_bloom_hash_known_values(oracle):
    """hash_ap of bloom_filter.hpp (:526-568) with the filter's salt, restated independently here in
    Python integers: 64-bit hashes take the two-word round, 32-bit hashes the one-word branch."""
    M = 0xFFFFFFFF
    seed = (0xA5A5A5A55A5A5A5A * 0xA5A5A5A5 + 1) & 0xFFFFFFFFFFFFFFFF
    salt = (0xAAAAAAAA * 0xAAAAAAAA + (seed & M)) & M
    rng = np.random.default_rng(5)
    for v in [0, 1, M, 0xFFFFFFFFFFFFFFFF, 0x0123456789ABCDEF] + [int(x) for x in rng.integers(0, 2 ** 63, 50)]:
        h = salt
        i1, i2 = v & M, (v >> 32) & M
        h ^= ((h << 7) & M) ^ ((i1 * (h >> 3)) & M) ^ (~(((h << 11) & M) + (i2 ^ (h >> 5))) & M)
        assert oracle.lib.oracle_bloom_hash(v, 1) == h & M, hex(v)
        h = salt
        h ^= ~(((h << 11) & M) + ((v & M) ^ (h >> 5))) & M
        assert oracle.lib.oracle_bloom_hash(v & M, 0) == h & M, hex(v)


def test_oracle_translate_equals_reference_codon_table(oracle, golden_dir):
    """6-frame translation of `mash screen` (CommandScreen.cpp:617-809): the restated table vs
    the output of the reference's own aaFromCodon for all 64 codons + invalid ones."""
    import json
    tab = json.load(open(os.path.join(golden_dir, "codon_table.json")))
    assert len(tab) == 64 + 8
    for cod, aa in tab.items():
        assert oracle.translate(cod.encode()).decode() == aa, cod
    assert "".join(tab[a + b + c] for a in "ACGT" for b in "ACGT" for c in "ACGT").count("*") == 3
    fr = oracle.six_frames(b"atgGCCTAAnACGT")
    assert fr[0] == b"MA**" and len(fr) == 6 and [len(x) for x in fr] == [4, 4, 4, 4, 4, 4]


def test_oracle_equals_reference_compare_vectors(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_compare_vectors.npz"))
    numer, denom, dist, pval = oracle.triangle(z["table"], z["nhash"], z["lengths"], 0, 64,
                                               int(z["k"]), float(z["kmer_space"]), stats=True)
    assert np.array_equal(numer, z["numer"])
    assert np.array_equal(denom, z["denom"])
    assert np.array_equal(dist, z["dist"])            # same libm, bit-exact
    assert np.array_equal(pval, z["pval"])
    n2, d2, _, _ = oracle.triangle(z["table"], z["nhash"], z["lengths"], 0, 64,
                                   int(z["k"]), float(z["kmer_space"]), stats=False)
    assert np.array_equal(n2, z["numer"]) and np.array_equal(d2, z["denom"])


def test_oracle_equals_reference_large_compare_vectors(oracle, golden_dir):
    """s = 3000 reference-run triangle (the size class the GPU compares window by window)."""
    z = np.load(os.path.join(golden_dir, "ref_compare_vectors_large.npz"))
    numer, denom, dist, pval = oracle.triangle(z["table"], z["nhash"], z["lengths"], 0, 16,
                                               int(z["k"]), float(z["kmer_space"]), stats=True)
    assert np.array_equal(numer, z["numer"]) and np.array_equal(denom, z["denom"])
    assert np.array_equal(dist, z["dist"]) and np.array_equal(pval, z["pval"])


def test_oracle_vs_reference_live(oracle, ref_oracle):
    """Where oracle/_ref exists: random k-mers hash identically; random sketches too."""
    rng = np.random.default_rng(99)
    for k in (1, 7, 8, 9, 15, 16, 17, 21, 24, 31, 32):
        for _ in range(50):
            kmer = bytes(rng.integers(33, 127, k, dtype=np.uint8))
            for use64 in (True, False):
                assert oracle.get_hash(kmer, 42, use64) == ref_oracle.get_hash(kmer, 42, use64)
            assert oracle.get_hash(kmer, 7, True) == ref_oracle.get_hash(kmer, 7, True)
    from workloads import synth
    for variant in range(4):
        recs = synth.adversarial_dna_records(rng, variant)
        for (k, s) in ((21, 1000), (15, 100), (32, 77)):
            p = oracle.params(k=k, s=s)
            a = oracle.sketch_records(recs, p)
            b = ref_oracle.sketch_records(recs, p)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]


def test_reference_cli_build_passes_the_reference_make_test(golden_dir, tmp_path):
    """oracle/_ref/mash-ref = the reference's own sources behind the capnp / GSL shims (make -C oracle
    refcli).  Where it has been built, it must reproduce the reference's three `make test` checks
    (Makefile.in:94-115) -- which pins the shims (and through them this repository's .msh codec and
    binomial tail) before its outputs are used as CLI fixtures."""
    import gzip, shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = os.path.join(root, "oracle", "_ref", "mash-ref")
    mash = os.path.join(root, "mash_amd", "bin", "mash")
    if not (os.path.exists(ref) and os.path.exists(mash)):
        pytest.skip("reference CLI not built here (make -C oracle refcli)")
    for n in ("reads1.fastq", "reads2.fastq"):
        with gzip.open(os.path.join(golden_dir, n + ".gz"), "rb") as fi, open(tmp_path / n, "wb") as fo:
            shutil.copyfileobj(fi, fo)
    run = lambda *a: subprocess.run([*a], cwd=tmp_path, capture_output=True, check=True).stdout
    # testSketch (reads): sketch with the reference's code, dump with the reference's code
    run(ref, "sketch", "-r", "-I", "reads", "reads1.fastq", "reads2.fastq", "-o", "reads.msh")
    # -r implies -M (sketchParameterSetup.cpp:62-65) and the current code dumps the "counts" array
    # (CommandInfo.cpp:265-283); the golden reads.json predates that: compare without the block
    dump = run(ref, "info", "-d", "reads.msh").decode()
    a, b = dump.index('\t\t\t"counts" :'), dump.index("\t\t}\n\t]")
    assert dump[:a] + dump[b:] == open(os.path.join(golden_dir, "reads.json")).read()
    # genome inputs are not shipped with the reference: the golden sketches stand in for them
    run(mash, "json2msh", os.path.join(golden_dir, "genomes.json"), "genomes.msh")
    assert run(ref, "info", "-d", "genomes.msh") == open(os.path.join(golden_dir, "genomes.json"), "rb").read()
    assert run(ref, "dist", "genomes.msh", "reads.msh") == open(os.path.join(golden_dir, "genomes.dist"), "rb").read()
    assert run(ref, "screen", "genomes.msh", "reads1.fastq", "reads2.fastq") == open(os.path.join(golden_dir, "screen"), "rb").read()


def test_oracle_equals_reference_cli_on_random_inputs(oracle, tmp_path):
    """Random FASTA files and parameters (k 1..32, -n, -Z, N runs, lower case, empty and short
    records): `mash-ref sketch` + `info -d` (the reference's own sketchFile / addMinHashes / heap)
    against the oracle's sketch_records -- hash list and length."""
    import json, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = os.path.join(root, "oracle", "_ref", "mash-ref")
    if not os.path.exists(ref):
        pytest.skip("reference CLI not built here (make -C oracle refcli)")
    rng = np.random.default_rng(4242)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    compared = 0
    for it in range(40):
        k = int(rng.integers(1, 33)); s = int(rng.choice([1, 10, 100, 400, 1000]))
        nonc = bool(rng.random() < 0.3); pc = bool(rng.random() < 0.2)
        recs = []
        for r in range(int(rng.integers(1, 5))):
            n = int(rng.choice([0, 3, 40, 500, 5000, 20000]))
            seq = lut[rng.integers(0, 4, n)].copy()
            if n and rng.random() < 0.4:
                i = int(rng.integers(0, n)); seq[i:i + int(rng.integers(1, 30))] = ord("N")
            seq = seq.tobytes()
            if rng.random() < 0.3:
                seq = seq[: n // 2] + seq[n // 2:].lower()
            recs.append((b"r%d c%d" % (r, it), seq))
        with open(tmp_path / "x.fa", "wb") as f:
            for name, seq in recs:
                f.write(b">" + name + b"\n" + seq + b"\n")
        args = ["sketch", "-k", str(k), "-s", str(s), "-o", "x"] + (["-n"] if nonc else []) + (["-Z"] if pc else []) + ["x.fa"]
        r = subprocess.run([ref, *args], cwd=tmp_path, capture_output=True, timeout=60)
        if r.returncode != 0:
            continue                                    # nothing sketchable: the reference refuses
        j = json.loads(subprocess.run([ref, "info", "-d", "x.msh"], cwd=tmp_path, capture_output=True, timeout=60, check=True).stdout)
        h, _, length, _, _ = oracle.sketch_records([q for _, q in recs], oracle.params(k=k, s=s, noncanonical=nonc, preserve_case=pc))
        assert [int(x) for x in h] == j["sketches"][0]["hashes"], (it, args)
        assert j["sketches"][0]["length"] == length, (it, args)
        compared += 1
    assert compared >= 25
