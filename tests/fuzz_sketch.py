#!/usr/bin/env python3
"""Differential fuzz of the sketching entry points against the oracle (the C restatement of
addMinHashes + MinHashHeap, oracle/mash_oracle.c) on random parameter sets and sequences:
k 1..32, sketch sizes from 1 to beyond the LDS selector, seeds, DNA canonical / forward-only,
preserve-case, custom and protein alphabets (32- and 64-bit hashes on either side of
alphabetSize^k = 2^32), records with N runs, lowercase, IUPAC codes, bytes >= 0x80 and records
shorter than k, empty sketches, many sketches per batch, multiplicities, -m, and in reads mode
-c / -b; the streamed session must agree with the one-batch call.

    python tests/fuzz_sketch.py [--n 300] [--seed 1] [--seconds 200]        # on a GPU box
(Test infrastructure: it lives under tests/ because it calls the oracle.)"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mash_amd import abi
from oracle import pyoracle

DNA = np.frombuffer(b"ACGT", dtype=np.uint8)
PROT = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)


def rand_seq(rng, n, letters):
    return letters[rng.integers(0, len(letters), n)].tobytes()


def dirty(rng, seq, protein):
    s = bytearray(seq)
    n = len(s)
    if n == 0:
        return bytes(s)
    for _ in range(int(rng.integers(0, 4))):                         # runs of N / X
        st, ln = int(rng.integers(0, n)), int(rng.integers(1, 40))
        s[st:st + ln] = (b"X" if protein else b"N") * len(s[st:st + ln])
    for _ in range(int(rng.integers(0, 3))):                         # lowercase stretches
        st, ln = int(rng.integers(0, n)), int(rng.integers(1, 200))
        s[st:st + ln] = bytes(s[st:st + ln]).lower()
    for _ in range(int(rng.integers(0, 3))):                         # odd bytes
        s[int(rng.integers(0, n))] = int(rng.choice([ord("R"), ord("y"), ord("-"), ord("*"), ord("U"), 1, 0x80, 0xFF, ord(".")]))
    return bytes(s)


def gen_sketches(rng, protein, k):
    letters = PROT if protein else DNA
    nsk = int(rng.choice([1, 1, 2, 5, 17]))
    out = []
    for _ in range(nsk):
        style = rng.choice(["one", "multi", "repeats", "short", "empty"], p=[0.35, 0.35, 0.15, 0.1, 0.05])
        if style == "empty":
            out.append([])
        elif style == "short":
            out.append([rand_seq(rng, int(rng.integers(0, k + 2)), letters) for _ in range(int(rng.integers(1, 4)))])
        elif style == "repeats":
            unit = rand_seq(rng, int(rng.integers(1, 50)), letters)
            out.append([unit * int(rng.integers(2, 200)) + rand_seq(rng, int(rng.integers(0, 300)), letters), (b"A" if not protein else b"L") * int(rng.integers(1, 400))])
        elif style == "multi":
            out.append([dirty(rng, rand_seq(rng, int(rng.integers(0, 4000)), letters), protein) for _ in range(int(rng.integers(1, 8)))])
        else:
            out.append([dirty(rng, rand_seq(rng, int(rng.integers(k, 40000)), letters), protein)])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=200)
    ap.add_argument("--dump", default="", help="directory for the inputs of differing cases")
    ap.add_argument("--against", default="gpu", choices=["gpu", "ref"],
                    help="gpu: the engine against the oracle (GPU box); ref: the oracle against the reference's own objects "
                         "(oracle/_ref/libmash_ref.so, CPU, only where /root/reference was available to build it)")
    a = ap.parse_args()
    sys.exit(run(a.n, a.seed, a.seconds, a.dump, a.against))


def run(n_cases, seed, seconds, dump="", against="gpu", quiet=False):
    import types
    a = types.SimpleNamespace(n=n_cases, seed=seed, seconds=seconds, dump=dump)
    orc = pyoracle.Oracle()
    if against == "ref":
        return run_ref(a, orc, quiet)
    eng = abi.MashGpu(0)
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    bad = ran = 0
    for case in range(a.n):
        if time.time() - t0 > a.seconds:
            break
        protein = rng.random() < 0.2
        if protein:
            alphabet = "ACDEFGHIKLMNPQRSTVWY" if rng.random() < 0.7 else "ACDEFGHIKLMNPQRSTVWYXBZ"
            k = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 12, 20, 32]))
            noncanon = True
        else:
            alphabet = str(rng.choice(["ACGT", "ACGT", "ACGT", "ACGTN", "AC", "ACGTacgt"]))
            k = int(rng.choice([1, 2, 3, 4, 7, 11, 15, 16, 17, 21, 27, 31, 32]))
            noncanon = bool(alphabet != "ACGT" or rng.random() < 0.25)
        s = int(rng.choice([1, 2, 10, 64, 100, 400, 1000, 3000, 10000, 14000]))
        seed = int(rng.choice([0, 1, 42, 42, 4294967295, int(rng.integers(0, 2 ** 32))]))
        preserve = bool(rng.random() < 0.2)
        mode = rng.choice(["plain", "plain", "counts", "mincopies", "cov", "bloom"])
        if mode in ("cov", "bloom") and s > 3000:
            s = 1000
        sketches = gen_sketches(rng, protein, k)
        kw = dict(k=k, s=s, seed=seed, alphabet=alphabet, noncanonical=noncanon, preserve_case=preserve)
        tag = "case %d %s k=%d s=%d seed=%d alphabet=%s noncanonical=%d preserve=%d nsk=%d" % (case, mode, k, s, seed, alphabet, noncanon, preserve, len(sketches))
        try:
            if mode in ("plain", "counts", "mincopies"):
                m = int(rng.choice([2, 3])) if mode == "mincopies" else 1
                counts = mode != "plain"
                if counts and s > 10000:
                    kw["s"] = s = 10000
                p = eng.params(min_copies=m, **kw)
                got = eng.sketch_host(sketches, p, counts=counts)
                got2 = eng.sketch_stream(sketches, p, counts=counts, piece=int(rng.choice([1 << 30, 4096, 977])))
                op = orc.params(min_copies=m, **kw)
                for i, recs in enumerate(sketches):
                    oh, oc, _, _, _ = orc.sketch_records(recs, op)
                    gh = got[0][i, : got[1][i]]
                    ok = np.array_equal(gh, oh) and np.array_equal(got2[0][i, : got2[1][i]], oh)
                    if counts and ok:
                        gc = got[2][i, : got[1][i]]
                        # (the largest kept hash stops being counted when the heap stops accepting it)
                        ok = np.array_equal(gc, oc) and np.array_equal(got2[2][i, : got2[1][i]], oc)
                    if not ok:
                        print("DIFF", tag, "sketch", i, "n got/oracle", len(gh), len(oh))
                        miss = np.setdiff1d(oh, gh)[:5]
                        extra_h = np.setdiff1d(gh, oh)[:5]
                        print("   missing", [hex(int(x)) for x in miss], "extra", [hex(int(x)) for x in extra_h], "record lengths", [len(r) for r in recs][:10])
                        if counts and np.array_equal(gh, oh):
                            w = np.nonzero(gc != oc)[0][:5]
                            print("   counts differ at", w, "got", gc[w], "oracle", oc[w], "stream", got2[2][i, : got2[1][i]][w])
                        if a.dump:
                            os.makedirs(a.dump, exist_ok=True)
                            np.savez_compressed(os.path.join(a.dump, "sfuzz_%d_%d.npz" % (a.seed, case)), tag=tag, sketch=i,
                                                lens=np.array([len(r) for r in recs]), bases=np.frombuffer(b"".join(recs), dtype=np.uint8),
                                                got=gh, oracle=oh)
                        bad += 1
                        break
            else:
                recs = [r for sk in sketches for r in sk]
                if mode == "cov":
                    # (target_cov 0: plain -r / -m -- the one-shot call keeps the read set in HBM, the session does not)
                    extra = dict(target_cov=float(rng.choice([0.0, 0.0, 1.05, 1.5, 3.0, 100.0])), min_copies=int(rng.choice([1, 1, 2, 3])))
                else:
                    extra = dict(bloom_bytes=int(rng.choice([1, 9, 300, 5000, 1 << 20])), target_cov=float(rng.choice([0.0, 0.0, 2.0])))
                if not recs:
                    recs = [b""]
                p = eng.params(**kw, **extra)
                gh, gc, used = eng.sketch_reads(recs, p)
                ch, cc, cused, _ = eng.sketch_reads_chunked(recs, p, int(rng.choice([1, 3, 50])))
                oh, oc, _, oused, _ = orc.sketch_reads(recs, orc.params(**kw, **extra))
                if not (np.array_equal(gh, oh) and np.array_equal(gc, oc) and used == oused and np.array_equal(ch, oh) and np.array_equal(cc, oc) and cused == oused):
                    print("DIFF", tag, extra, "n got/oracle", len(gh), len(oh), "used", used, cused, oused)
                    bad += 1
        except abi.MashGpuError as e:
            msg = str(e)
            if "canonical k-mers need the ACGT alphabet" in msg or "unsupported" in msg.lower() and "multiplicities" in msg:
                continue
            print("ERROR", tag, msg)
            bad += 1
        ran += 1
    print("cases: %d  differing: %d  [seed %d, %.0f s]" % (ran, bad, a.seed, time.time() - t0))
    return 1 if bad else 0


def gen_case(rng):
    """the parameter / input generator shared by both modes (same draws in the same order as run())"""
    protein = rng.random() < 0.2
    if protein:
        alphabet = "ACDEFGHIKLMNPQRSTVWY" if rng.random() < 0.7 else "ACDEFGHIKLMNPQRSTVWYXBZ"
        k = int(rng.choice([1, 2, 3, 5, 7, 8, 9, 12, 20, 32]))
        noncanon = True
    else:
        alphabet = str(rng.choice(["ACGT", "ACGT", "ACGT", "ACGTN", "AC", "ACGTacgt"]))
        k = int(rng.choice([1, 2, 3, 4, 7, 11, 15, 16, 17, 21, 27, 31, 32]))
        noncanon = bool(alphabet != "ACGT" or rng.random() < 0.25)
    s = int(rng.choice([1, 2, 10, 64, 100, 400, 1000, 3000, 10000, 14000]))
    seed = int(rng.choice([0, 1, 42, 42, 4294967295, int(rng.integers(0, 2 ** 32))]))
    preserve = bool(rng.random() < 0.2)
    mode = rng.choice(["plain", "plain", "counts", "mincopies", "cov", "bloom"])
    if mode in ("cov", "bloom") and s > 3000:
        s = 1000
    sketches = gen_sketches(rng, protein, k)
    return protein, alphabet, k, noncanon, s, seed, preserve, mode, sketches


def run_ref(a, orc, quiet=False):
    """the oracle (C restatement) against the reference's own MinHashHeap / addMinHashes / bloom_filter on
    the fuzzer's cases; bytes >= 0x80 are replaced first (the reference indexes alphabet[] with a negative
    char there, Sketch.cpp:550 -- undefined behaviour, nothing to compare with)"""
    refo = pyoracle.Oracle(ref=True)
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    bad = ran = 0
    clean = bytes(range(128)) + b"N" * 128
    for case in range(a.n):
        if time.time() - t0 > a.seconds:
            break
        protein, alphabet, k, noncanon, s, seed, preserve, mode, sketches = gen_case(rng)
        sketches = [[r.translate(clean) for r in sk] for sk in sketches]
        kw = dict(k=k, s=s, seed=seed, alphabet=alphabet, noncanonical=noncanon, preserve_case=preserve)
        tag = "case %d %s k=%d s=%d seed=%d alphabet=%s noncanonical=%d preserve=%d nsk=%d" % (case, mode, k, s, seed, alphabet, noncanon, preserve, len(sketches))
        if mode in ("plain", "counts", "mincopies"):
            m = int(rng.choice([2, 3])) if mode == "mincopies" else 1
            if mode != "plain" and s > 10000:
                kw["s"] = 10000
            rng.choice([1 << 30, 4096, 977])                          # (the draw run() spends on the piece size)
            for i, recs in enumerate(sketches):
                x = orc.sketch_records(recs, orc.params(min_copies=m, **kw))
                y = refo.sketch_records(recs, refo.params(min_copies=m, **kw))
                if not (np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2:] == y[2:]):
                    print("DIFF", tag, "sketch", i)
                    bad += 1
                    break
        else:
            recs = [r for sk in sketches for r in sk] or [b""]
            if mode == "cov":
                extra = dict(target_cov=float(rng.choice([1.05, 1.5, 3.0, 100.0])), min_copies=int(rng.choice([1, 1, 2])))
            else:
                extra = dict(bloom_bytes=int(rng.choice([1, 9, 300, 5000, 1 << 20])), target_cov=float(rng.choice([0.0, 0.0, 2.0])))
            rng.choice([1, 3, 50])                                    # (the draw run() spends on the chunk size)
            x = orc.sketch_reads(recs, orc.params(**kw, **extra))
            y = refo.sketch_reads(recs, refo.params(**kw, **extra))
            if not (np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2:] == y[2:]):
                print("DIFF", tag, extra)
                bad += 1
        ran += 1
    if not quiet or bad:
        print("oracle vs reference objects: cases %d  differing %d  [seed %d, %.0f s]" % (ran, bad, a.seed, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    main()
