#!/usr/bin/env python3
"""Generate tests/golden/ fixtures.  Run HERE (the container with /root/reference
mounted); the GPU box only ever sees the committed outputs.

Sources (all reference TEST DATA / golden outputs, no reference code):
  /root/reference/test/ref/genomes.json   -> genomes_sketches.npz (3 x 1000 u64 + lengths + names)
  /root/reference/test/ref/reads.json     -> reads_sketch.npz     (1000 u64 + length + comment)
  /root/reference/test/ref/genomes.dist   -> genomes.dist  (verbatim golden text)
  /root/reference/test/ref/screen         -> screen        (verbatim golden text)
  /root/reference/test/reads{1,2}.fastq   -> reads{1,2}.fastq.gz (sketch input of reads.json)
  doc/sphinx/tutorials.rst:24,56-57       -> tutorial known-answers (hard-coded below)
Independent numeric cross-check:
  scipy.stats.binom.sf (Boost-backed)     -> binom_sf.json (p-value fixtures)
Reference-run vectors (oracle/_ref = the reference's own objects, run here):
  random/adversarial sequences            -> ref_sketch_vectors.npz
  random sketch pairs                     -> ref_compare_vectors.npz, ref_compare_vectors_large.npz (s = 3000)
  read sets with -m 2..5                  -> ref_sketch_vectors_m.npz
  aaFromCodon over all codons             -> codon_table.json
  read sets with -c (and -m)              -> ref_sketch_vectors_c.npz
  read sets with -b (Bloom filter)        -> ref_sketch_vectors_b.npz
"""
import gzip
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def load_info_json(path):
    with open(path) as f:
        return json.load(f)


def main():
    g = load_info_json(f"{REF}/test/ref/genomes.json")
    hashes = np.array([s["hashes"] for s in g["sketches"]], dtype=np.uint64)
    np.savez_compressed(
        f"{HERE}/genomes_sketches.npz",
        hashes=hashes,
        lengths=np.array([s["length"] for s in g["sketches"]], dtype=np.uint64),
        names=np.array([s["name"] for s in g["sketches"]]),
        comments=np.array([s["comment"] for s in g["sketches"]]),
        kmer=g["kmer"], sketchSize=g["sketchSize"], hashSeed=g["hashSeed"], hashBits=g["hashBits"],
    )
    r = load_info_json(f"{REF}/test/ref/reads.json")
    s0 = r["sketches"][0]
    np.savez_compressed(
        f"{HERE}/reads_sketch.npz",
        hashes=np.array(s0["hashes"], dtype=np.uint64),
        length=np.uint64(s0["length"]), name=s0["name"], comment=s0["comment"],
        kmer=r["kmer"], sketchSize=r["sketchSize"], hashSeed=r["hashSeed"],
    )
    shutil.copy(f"{REF}/test/ref/genomes.dist", f"{HERE}/genomes.dist")
    shutil.copy(f"{REF}/test/ref/screen", f"{HERE}/screen")
    shutil.copy(f"{REF}/test/ref/genomes.json", f"{HERE}/genomes.json")
    shutil.copy(f"{REF}/test/ref/reads.json", f"{HERE}/reads.json")
    for n in ("reads1.fastq", "reads2.fastq"):
        with open(f"{REF}/test/{n}", "rb") as fi, gzip.GzipFile(f"{HERE}/{n}.gz", "wb", mtime=0) as fo:
            shutil.copyfileobj(fi, fo)

    # ---- scipy binomial tail fixtures -----------------------------------
    from scipy.stats import binom
    rng = np.random.default_rng(12345)
    cases = []
    for n in (1000, 999, 400, 10000, 37):
        for r_ in (1e-9, 1.0535e-7, 3.3e-6, 1e-4, 2.5e-3, 0.04, 0.3, 0.77):
            for x in sorted(set([1, 2, 3, 5, 17, 41, 100, n // 2, n - 1, n] + list(rng.integers(1, n + 1, 3)))):
                if x > n:
                    continue
                v = float(binom.sf(x - 1, n, r_))
                cases.append({"x": int(x), "n": int(n), "r": r_, "sf": v})
    with open(f"{HERE}/binom_sf.json", "w") as f:
        json.dump(cases, f)

    # ---- reference-run vectors (needs oracle/_ref) ------------------------
    from oracle import pyoracle
    pyoracle.build(ref=True)
    ref = pyoracle.Oracle(ref=True)
    from workloads import synth

    seqs, outs = [], {}
    cfgs = []
    rng = np.random.default_rng(777)
    idx = 0
    for (k, s, alphabet, nonc, pc) in [
        (21, 1000, "ACGT", False, False),
        (21, 50, "ACGT", False, False),
        (31, 400, "ACGT", False, False),
        (32, 128, "ACGT", False, False),
        (16, 300, "ACGT", False, False),      # 32-bit hashes
        (11, 64, "ACGT", False, False),
        (5, 1000, "ACGT", False, False),      # kmer space (512 canonical) < s
        (21, 200, "ACGT", True, False),       # -n
        (21, 200, "ACGT", False, True),       # -Z
        (9, 300, "ACDEFGHIKLMNPQRSTVWY", True, False),   # protein
        (3, 100, "ACDEFGHIKLMNPQRSTVWY", True, False),   # protein, 32-bit
    ]:
        p = ref.params(k=k, s=s, alphabet=alphabet, noncanonical=nonc, preserve_case=pc)
        for variant in range(4):
            if alphabet == "ACGT":
                recs = synth.adversarial_dna_records(rng, variant)
            else:
                recs = synth.random_protein_records(rng, variant)
            h, c, length, setsz, rc = ref.sketch_records(recs, p)
            cfgs.append(dict(k=k, s=s, alphabet=alphabet, noncanonical=nonc, preserve_case=pc,
                             nrec=len(recs), length=length, rc=rc, set_size=setsz, idx=idx))
            outs[f"bases_{idx}"] = np.frombuffer(b"".join(recs), dtype=np.uint8)
            outs[f"reclen_{idx}"] = np.array([len(x) for x in recs], dtype=np.uint64)
            outs[f"hashes_{idx}"] = h
            outs[f"counts_{idx}"] = c
            idx += 1
    outs["cfgs"] = np.array(json.dumps(cfgs))
    np.savez_compressed(f"{HERE}/ref_sketch_vectors.npz", **outs)

    # compare vectors
    table, nhash, lengths = synth.clustered_sketches(64, 1000, clusters=4, seed=5)
    # make some rows short / degenerate
    nhash[3] = 0
    nhash[7] = 1
    nhash[11] = 999
    nhash[13] = 500
    table[21] = table[20]
    nhash[21] = nhash[20]
    kspace = 4.0 ** 21
    numer, denom, dist, pval = ref.triangle(table, nhash, lengths, 0, 64, 21, kspace, stats=True)
    np.savez_compressed(f"{HERE}/ref_compare_vectors.npz", table=table, nhash=nhash, lengths=lengths,
                        numer=numer, denom=denom, dist=dist, pval=pval, k=21, kmer_space=kspace)
    make_large_compare_vectors(ref)
    make_mincopies_vectors(ref)
    make_codon_table(ref)
    make_cov_vectors(ref)
    make_bloom_vectors(ref)
    print("golden fixtures written to", HERE)


def make_large_compare_vectors(ref):
    """Reference-run triangle at a sketch size where the GPU path compares value window by value
    window (s = 3000, k = 31) -> ref_compare_vectors_large.npz."""
    from workloads import synth
    s = 3000
    table, nhash, lengths = synth.clustered_sketches(16, s, clusters=2, seed=31, pool=int(1.5 * s), private=int(0.4 * s))
    nhash[2] = 0
    nhash[5] = 37
    nhash[6] = s - 1
    nhash[9] = s // 2
    table[12] = table[4]
    nhash[12] = nhash[4]
    for i in range(16):
        table[i, nhash[i]:] = np.uint64(0xFFFFFFFFFFFFFFFF)
    kspace = 4.0 ** 31
    numer, denom, dist, pval = ref.triangle(table, nhash, lengths, 0, 16, 31, kspace, stats=True)
    np.savez_compressed(f"{HERE}/ref_compare_vectors_large.npz", table=table, nhash=nhash, lengths=lengths,
                        numer=numer, denom=denom, dist=dist, pval=pval, k=31, kmer_space=kspace)


def make_cov_vectors(ref):
    """`mash sketch -r -c <cov>` (and with -m): the record loop of sketchFile with the reference's
    MinHashHeap and its early stop (Sketch.cpp:1258) -> ref_sketch_vectors_c.npz (hashes, counts,
    reads used)."""
    from workloads import synth
    rng = np.random.default_rng(9090)
    outs, cfgs = {}, []
    for idx, (k, s, m, cov, glen, nreads) in enumerate([
        (21, 1000, 1, 2.0, 20000, 4000),
        (21, 200, 1, 5.0, 5000, 3000),
        (21, 200, 2, 3.0, 5000, 3000),
        (16, 100, 1, 1.2, 4000, 1500),     # 32-bit hashes
        (11, 64, 3, 6.0, 3000, 2500),
        (21, 500, 1, 50.0, 4000, 800),     # never reached: all reads used
    ]):
        g = synth._rand_dna(rng, glen)
        recs = []
        for _ in range(nreads):
            l = int(rng.integers(40, 151))
            st = int(rng.integers(0, glen - l))
            r = bytearray(g[st:st + l])
            if rng.random() < 0.1:
                r[int(rng.integers(0, l))] = ord("N")
            if rng.random() < 0.5:
                r = bytearray(bytes(r).translate(bytes.maketrans(b"ACGTN", b"TGCAN"))[::-1])
            recs.append(bytes(r))
        recs.insert(3, b"ACGTAC")
        p = ref.params(k=k, s=s, min_copies=m, target_cov=cov)
        h, c, setsz, used, mult = ref.sketch_reads(recs, p)
        cfgs.append(dict(k=k, s=s, min_copies=m, target_cov=cov, nrec=len(recs), used=used, set_size=setsz, mult=mult, idx=idx))
        outs[f"bases_{idx}"] = np.frombuffer(b"".join(recs), dtype=np.uint8)
        outs[f"reclen_{idx}"] = np.array([len(x) for x in recs], dtype=np.uint64)
        outs[f"hashes_{idx}"] = h
        outs[f"counts_{idx}"] = c
    outs["cfgs"] = np.array(json.dumps(cfgs))
    np.savez_compressed(f"{HERE}/ref_sketch_vectors_c.npz", **outs)


def make_bloom_vectors(ref):
    """`mash sketch -b <bytes>` (optionally with -c): the record loop of sketchFile with the
    reference's MinHashHeap and ITS Bloom filter (MinHashHeap.cpp:19-41,78-94; the vendored
    bloom_filter.hpp, compiled from /root/reference) -> ref_sketch_vectors_b.npz.  Small filters
    on purpose: aliasing (false positives) is what makes the result depend on the order."""
    from workloads import synth
    rng = np.random.default_rng(4242)
    outs, cfgs = {}, []
    for idx, (k, s, bloom, cov, glen, nreads) in enumerate([
        (21, 1000, 1 << 20, 0.0, 20000, 4000),
        (21, 200, 4096, 0.0, 5000, 3000),        # 32768 bits for ~3*10^5 k-mers: mostly aliases
        (21, 200, 100000, 2.5, 5000, 3000),      # with the early stop of -c
        (16, 100, 50000, 0.0, 4000, 1500),       # 32-bit hashes: the 4-byte branch of hash_ap
        (11, 64, 1000, 0.0, 3000, 2500),
        (21, 500, 1 << 16, 0.0, 40000, 300),     # low coverage: the sketch never fills
        (21, 300, 3, 0.0, 3000, 500),            # 24 bits
    ]):
        g = synth._rand_dna(rng, glen)
        recs = []
        for _ in range(nreads):
            l = int(rng.integers(40, 151))
            st = int(rng.integers(0, glen - l))
            r = bytearray(g[st:st + l])
            if rng.random() < 0.1:
                r[int(rng.integers(0, l))] = ord("N")
            if rng.random() < 0.5:
                r = bytearray(bytes(r).translate(bytes.maketrans(b"ACGTN", b"TGCAN"))[::-1])
            recs.append(bytes(r))
        recs.insert(3, b"ACGTAC")
        p = ref.params(k=k, s=s, target_cov=cov, bloom_bytes=bloom)
        h, c, setsz, used, mult = ref.sketch_reads(recs, p)
        cfgs.append(dict(k=k, s=s, bloom_bytes=bloom, target_cov=cov, nrec=len(recs), used=used, set_size=setsz, mult=mult, idx=idx))
        outs[f"bases_{idx}"] = np.frombuffer(b"".join(recs), dtype=np.uint8)
        outs[f"reclen_{idx}"] = np.array([len(x) for x in recs], dtype=np.uint64)
        outs[f"hashes_{idx}"] = h
        outs[f"counts_{idx}"] = c
    outs["cfgs"] = np.array(json.dumps(cfgs))
    np.savez_compressed(f"{HERE}/ref_sketch_vectors_b.npz", **outs)


def make_codon_table(ref):
    """aaFromCodon (CommandScreen.cpp:625-809) run on every ACGT codon and on codons holding
    other bytes -> codon_table.json"""
    tab = {}
    for a in "ACGT":
        for b in "ACGT":
            for c in "ACGT":
                tab[a + b + c] = ref.translate((a + b + c).encode()).decode()
    for cod in ("ANA", "NNN", "acg", "A*C", "RYK", "AC\n", "\nGT", "TG-"):
        tab[cod] = ref.translate(cod.encode()).decode()
    with open(f"{HERE}/codon_table.json", "w") as f:
        json.dump(tab, f, indent=0, sort_keys=True)


def make_mincopies_vectors(ref):
    """`mash sketch -r -m <m>`: read sets sketched by the reference's own MinHashHeap with
    multiplicityMinimum = m (oracle/_ref) -> ref_sketch_vectors_m.npz."""
    from workloads import synth
    rng = np.random.default_rng(4242)
    outs, cfgs = {}, []
    for idx, (k, s, m, glen, cov) in enumerate([
        (21, 1000, 2, 20000, 6.0),      # full sketch, typical
        (21, 200, 3, 6000, 8.0),
        (16, 100, 2, 5000, 5.0),        # 32-bit hashes
        (11, 64, 4, 3000, 12.0),
        (21, 5000, 2, 4000, 3.0),       # fewer than s hashes reach m copies
        (5, 300, 2, 2000, 4.0),         # k-mer space smaller than s
        (21, 300, 2, 8000, 0.7),        # low coverage: most k-mers are singletons
        (31, 500, 5, 3000, 20.0),
    ]):
        g = synth._rand_dna(rng, glen)
        recs = []
        total = 0
        while total < cov * glen:
            l = int(rng.integers(40, 151))
            st = int(rng.integers(0, glen - l))
            r = bytearray(g[st:st + l])
            u = rng.random()
            if u < 0.15:
                r[int(rng.integers(0, l))] = ord("ACGT"[int(rng.integers(0, 4))])     # substitution
            elif u < 0.20:
                r[int(rng.integers(0, l))] = ord("N")
            elif u < 0.25:
                r = bytearray(bytes(r).lower())
            if rng.random() < 0.5:
                r = bytearray(bytes(r).upper().translate(bytes.maketrans(b"ACGTN", b"TGCAN"))[::-1])
            recs.append(bytes(r))
            total += l
        recs.append(b"ACGTAC")                                  # shorter than k (k >= 11): skipped
        p = ref.params(k=k, s=s, min_copies=m)
        h, c, length, setsz, rc = ref.sketch_records(recs, p)
        cfgs.append(dict(k=k, s=s, min_copies=m, nrec=len(recs), length=length, rc=rc, set_size=setsz, idx=idx))
        outs[f"bases_{idx}"] = np.frombuffer(b"".join(recs), dtype=np.uint8)
        outs[f"reclen_{idx}"] = np.array([len(x) for x in recs], dtype=np.uint64)
        outs[f"hashes_{idx}"] = h
        outs[f"counts_{idx}"] = c
    outs["cfgs"] = np.array(json.dumps(cfgs))
    np.savez_compressed(f"{HERE}/ref_sketch_vectors_m.npz", **outs)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("mincopies", "codons", "cov", "bloom", "largecompare"):
        from oracle import pyoracle
        pyoracle.build(ref=True)
        {"mincopies": make_mincopies_vectors, "codons": make_codon_table, "cov": make_cov_vectors, "bloom": make_bloom_vectors,
         "largecompare": make_large_compare_vectors}[sys.argv[1]](pyoracle.Oracle(ref=True))
    else:
        main()
