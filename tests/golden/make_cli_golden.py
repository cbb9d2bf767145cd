#!/usr/bin/env python3
"""Golden outputs of the REFERENCE CLI for the CLI parity test (tests/test_cli_reference.py).

    make -C oracle refcli                     # builds oracle/_ref/mash-ref from the reference's own,
                                              # unmodified sources (oracle/Makefile, "recipe B")
    python tests/golden/make_cli_golden.py    # writes tests/golden/cli/{in/*, cases.json, *.out}

Inputs are small synthetic files written here (committed, so the GPU box needs no generator); every
case is a list of set-up commands (sketches written by the same binary) and one command whose stdout
is the golden.  The test replays exactly these argument lists through mash_amd/bin/mash."""
import gzip, json, os, shutil, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "cli")
REFCLI = os.path.join(ROOT, "oracle", "_ref", "mash-ref")


def rand_dna(rng, n):
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].tobytes()


def mutate(rng, seq, rate):
    a = np.frombuffer(seq, dtype=np.uint8).copy()
    idx = np.nonzero(rng.random(len(a)) < rate)[0]
    a[idx] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, len(idx))]
    return a.tobytes()


def fasta(records, width=70):
    out = []
    for name, seq in records:
        out.append(b">" + name + b"\n")
        out += [seq[i:i + width] + b"\n" for i in range(0, len(seq), width)] or [b"\n"]
    return b"".join(out)


def make_inputs(d):
    rng = np.random.default_rng(20260925)
    g1 = rand_dna(rng, 30000)
    g2a, g2b = rand_dna(rng, 14000), rand_dna(rng, 9000)
    g2b = g2b[:3000] + g2b[3000:3600].lower() + b"N" * 40 + g2b[3640:]
    open(f"{d}/g1.fa", "wb").write(fasta([(b"g1 first genome", g1)]))
    open(f"{d}/g2.fa", "wb").write(fasta([(b"g2_c1 two contigs and a stub", g2a), (b"g2_stub", b"ACGTACGTAC"), (b"g2_c2\tsecond", g2b)]))
    open(f"{d}/g3.fa", "wb").write(fasta([(b"g3 g1 with 3% substitutions", mutate(rng, g1, 0.03))], width=60))
    open(f"{d}/g4.fa", "wb").write(fasta([(b"g4", mutate(rng, g1, 0.10)[:20000] + g2a[:5000])]))
    with gzip.GzipFile(f"{d}/g5.fa.gz", "wb", mtime=0) as f:
        f.write(fasta([(b"g5 gzipped", mutate(rng, g2a, 0.02))]))
    multi = [(b"m%d rec %d" % (i, i), rand_dna(rng, int(n))) for i, n in enumerate((8000, 15, 12000, 40, 5000, 9000))]
    multi[4] = (multi[4][0], mutate(rng, multi[0][1], 0.05)[:5000])
    open(f"{d}/multi.fa", "wb").write(fasta(multi))
    reads = []
    for i in range(400):
        src = g1 if i % 3 else g2a
        o = int(rng.integers(0, len(src) - 120))
        r = mutate(rng, src[o:o + 120], 0.01)
        reads.append(b"@read%d/1 len=120\n" % i + r + b"\n+\n" + bytes(rng.integers(40, 70, 120).astype(np.uint8)) + b"\n")
    open(f"{d}/reads.fq", "wb").write(b"".join(reads))
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    prot = [(b"p%d protein" % i, aa[rng.integers(0, 20, 1500)].tobytes()) for i in range(4)]
    prot.append((b"p4 close to p0", prot[0][1][:700] + aa[rng.integers(0, 20, 800)].tobytes()))
    open(f"{d}/prot.fa", "wb").write(fasta(prot))
    open(f"{d}/list.txt", "w").write("g1.fa\ng3.fa\ng4.fa\n")


# name, set-up commands, command whose stdout is compared
CASES = [
    ("sketch3_dump", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["info", "-d", "a.msh"]),
    ("sketch3_table", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["info", "-t", "a.msh"]),
    ("sketch3_header", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["info", "-H", "a.msh"]),
    # (plain `info` is not a fixture: its column printer wraps at the width ioctl(0, TIOCGWINSZ) reports,
    #  Command.cpp:422-426, which is uninitialised when stdin is not a terminal)
    ("sketch_individual_k16", [["sketch", "-k", "16", "-s", "120", "-i", "-o", "b", "multi.fa"]], ["info", "-d", "b.msh"]),
    ("sketch_noncanonical_k31", [["sketch", "-k", "31", "-s", "250", "-n", "-o", "c", "g1.fa", "g2.fa"]], ["info", "-d", "c.msh"]),
    ("sketch_preserve_case", [["sketch", "-Z", "-s", "200", "-o", "z", "g2.fa"]], ["info", "-d", "z.msh"]),
    ("sketch_protein", [["sketch", "-a", "-k", "9", "-s", "150", "-i", "-o", "p", "prot.fa"]], ["info", "-d", "p.msh"]),
    ("sketch_alphabet", [["sketch", "-z", "ACGTN", "-k", "12", "-s", "100", "-o", "y", "g2.fa"]], ["info", "-d", "y.msh"]),
    ("sketch_reads", [["sketch", "-r", "-s", "200", "-o", "r", "reads.fq"]], ["info", "-d", "r.msh"]),
    ("sketch_reads_m2", [["sketch", "-r", "-m", "2", "-s", "200", "-o", "r2", "reads.fq"]], ["info", "-d", "r2.msh"]),
    ("sketch_reads_c", [["sketch", "-r", "-c", "1.5", "-s", "100", "-o", "rc", "reads.fq"]], ["info", "-d", "rc.msh"]),
    ("sketch_counts", [["sketch", "-M", "-s", "100", "-o", "m", "g1.fa", "g3.fa"]], ["info", "-d", "m.msh"]),
    ("sketch_counts_table", [["sketch", "-M", "-s", "100", "-o", "m", "g1.fa", "g3.fa"]], ["info", "-c", "m.msh"]),
    ("sketch_seed", [["sketch", "-S", "7", "-s", "100", "-o", "sd", "g1.fa"]], ["info", "-d", "sd.msh"]),
    ("sketch_list", [["sketch", "-s", "150", "-l", "-o", "l", "list.txt"]], ["info", "-d", "l.msh"]),
    ("sketch_gz", [["sketch", "-s", "150", "-o", "gz", "g5.fa.gz"]], ["info", "-d", "gz.msh"]),
    ("sketch_threads", [["sketch", "-p", "3", "-s", "150", "-o", "t", "g1.fa", "g2.fa", "g3.fa", "g4.fa", "g5.fa.gz"]], ["info", "-d", "t.msh"]),
    ("sketch_id_comment", [["sketch", "-I", "myid", "-C", "my comment", "-s", "50", "-o", "ic", "g2.fa"]], ["info", "-t", "ic.msh"]),
    ("dist_files", [], ["dist", "-s", "400", "g1.fa", "g3.fa", "g4.fa"]),
    ("dist_sketch_vs_file", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "a.msh", "g4.fa"]),
    ("dist_self", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "a.msh", "a.msh"]),
    ("dist_table", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "-t", "a.msh", "a.msh", "g4.fa"]),
    ("dist_maxd", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "-d", "0.05", "a.msh", "a.msh"]),
    ("dist_maxp", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "-v", "1e-30", "a.msh", "a.msh"]),
    ("dist_comment", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "-C", "a.msh", "g4.fa"]),
    ("dist_individual", [], ["dist", "-i", "-k", "16", "-s", "120", "multi.fa", "multi.fa"]),
    ("dist_list", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "-l", "a.msh", "list.txt"]),
    ("dist_table_maxd", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "-t", "-d", "0.2", "a.msh", "a.msh"]),
    ("triangle_sketch", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["triangle", "a.msh"]),
    ("triangle_edge", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["triangle", "-E", "a.msh"]),
    ("triangle_files", [], ["triangle", "-s", "200", "g1.fa", "g3.fa", "g4.fa", "g2.fa"]),
    ("triangle_comment", [], ["triangle", "-C", "-s", "200", "g1.fa", "g3.fa", "g4.fa"]),
    ("triangle_one_multifasta", [], ["triangle", "-k", "16", "-s", "120", "multi.fa"]),
    ("triangle_edge_maxd", [], ["triangle", "-d", "0.1", "-s", "200", "g1.fa", "g3.fa", "g4.fa", "g2.fa"]),
    ("triangle_list", [], ["triangle", "-l", "-s", "100", "list.txt"]),
    ("screen_default", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["screen", "a.msh", "reads.fq"]),
    ("screen_winner", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["screen", "-w", "a.msh", "reads.fq"]),
    ("screen_filters", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["screen", "-i", "0.5", "-v", "1e-5", "a.msh", "reads.fq"]),
    ("screen_two_files", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["screen", "a.msh", "reads.fq", "g4.fa"]),
    ("paste_table", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"], ["sketch", "-s", "300", "-o", "d", "g4.fa"],
                     ["paste", "both", "a.msh", "d.msh"]], ["info", "-t", "both.msh"]),
    ("paste_dump", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"], ["sketch", "-s", "300", "-o", "d", "g4.fa"],
                    ["paste", "both", "a.msh", "d.msh"]], ["info", "-d", "both.msh"]),
]


# Second batch (all 22 replayed on the GPU with tools/cli_reference_report.py --extra: identical).
# New cases start with "confirmed_on_gpu": false in cases.json, which the test treats as
# non-blocking (xfail(strict=False)) until a GPU run has confirmed them.
EXTRA = [
    ("x_sketch_k32", [["sketch", "-k", "32", "-s", "200", "-o", "k32", "g1.fa", "g3.fa"]], ["info", "-d", "k32.msh"]),
    ("x_sketch_k5", [["sketch", "-k", "5", "-s", "1000", "-o", "k5", "g1.fa"]], ["info", "-d", "k5.msh"]),
    ("x_sketch_s1", [["sketch", "-s", "1", "-o", "s1", "g1.fa", "g2.fa"]], ["info", "-d", "s1.msh"]),
    ("x_sketch_s5000", [["sketch", "-s", "5000", "-o", "s5k", "g1.fa", "g3.fa"]], ["info", "-t", "s5k.msh"]),
    ("x_triangle_s5000", [], ["triangle", "-s", "5000", "g1.fa", "g3.fa", "g4.fa", "g2.fa"]),
    ("x_dist_s5000", [], ["dist", "-s", "5000", "g1.fa", "g3.fa", "g4.fa"]),
    ("x_sketch_genome_size", [["sketch", "-r", "-g", "30k", "-s", "100", "-o", "gs", "reads.fq"]], ["info", "-t", "gs.msh"]),
    ("x_sketch_reads_m3", [["sketch", "-m", "3", "-s", "100", "-o", "m3", "reads.fq"]], ["info", "-d", "m3.msh"]),
    ("x_sketch_protein_concat", [["sketch", "-a", "-s", "200", "-o", "pc", "prot.fa"]], ["info", "-d", "pc.msh"]),
    ("x_sketch_stdin_name", [["sketch", "-s", "50", "-o", "two", "g1.fa", "g1.fa"]], ["info", "-t", "two.msh"]),
    ("x_dist_size_mismatch", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"], ["sketch", "-s", "100", "-o", "e", "g4.fa", "g3.fa"]],
     ["dist", "a.msh", "e.msh"]),
    ("x_dist_threads", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "-p", "4", "a.msh", "a.msh", "g4.fa"]),
    ("x_dist_table_comment", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "-t", "-C", "a.msh", "a.msh"]),
    ("x_dist_protein", [], ["dist", "-a", "-i", "-s", "150", "prot.fa", "prot.fa"]),
    ("x_dist_noncanonical", [], ["dist", "-n", "-k", "17", "-s", "200", "g1.fa", "g3.fa"]),
    ("x_triangle_threads_edge", [], ["triangle", "-p", "3", "-E", "-s", "200", "g1.fa", "g3.fa", "g4.fa", "g2.fa"]),
    ("x_triangle_maxp", [], ["triangle", "-v", "1e-20", "-s", "200", "g1.fa", "g3.fa", "g4.fa", "g2.fa"]),
    ("x_triangle_protein", [], ["triangle", "-a", "-s", "150", "prot.fa"]),
    ("x_screen_threads", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["screen", "-p", "4", "a.msh", "reads.fq"]),
    ("x_screen_k16", [["sketch", "-k", "16", "-s", "200", "-o", "a16", "g1.fa", "g2.fa", "g3.fa"]], ["screen", "a16.msh", "reads.fq"]),
    ("x_screen_genomes", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["screen", "a.msh", "g3.fa", "g4.fa"]),
    ("x_paste_three", [["sketch", "-s", "100", "-o", "q1", "g1.fa"], ["sketch", "-s", "100", "-o", "q2", "g2.fa"], ["sketch", "-s", "100", "-o", "q3", "g3.fa"],
                       ["paste", "q", "q1.msh", "q2.msh", "q3.msh"]], ["dist", "q.msh", "q.msh"]),
]


# Third batch: BASELINE.json config 5 exactly (k = 31, 64-bit hashes, s = 10 000: the NT = 1024 sketch
# selector and the value-window compare path) through the CLI.
C5 = [
    ("c5_sketch_k31_s10000", [["sketch", "-k", "31", "-s", "10000", "-o", "c5", "g1.fa", "g3.fa", "g4.fa", "g2.fa"]], ["info", "-t", "c5.msh"]),
    ("c5_triangle_k31_s10000", [], ["triangle", "-k", "31", "-s", "10000", "g1.fa", "g3.fa", "g4.fa", "g2.fa"]),
    ("c5_dist_k31_s10000", [["sketch", "-k", "31", "-s", "10000", "-o", "c5", "g1.fa", "g3.fa", "g4.fa", "g2.fa"]], ["dist", "c5.msh", "c5.msh"]),
    ("c5_dist_msh_vs_fasta", [["sketch", "-k", "31", "-s", "10000", "-o", "c5", "g1.fa", "g3.fa"]], ["dist", "-t", "c5.msh", "g4.fa", "g2.fa"]),
]


# Fourth batch: -b, the Bloom filter in front of the heap (MinHashHeap.cpp:19-41,78-94).  Small filters:
# aliasing is what makes the outcome depend on the order of the k-mers.
BLOOM = [
    ("b_sketch_bloom_1k", [["sketch", "-b", "1K", "-s", "100", "-o", "b1", "reads.fq"]], ["info", "-d", "b1.msh"]),
    ("b_sketch_bloom_300", [["sketch", "-b", "300", "-s", "200", "-o", "b2", "reads.fq"]], ["info", "-d", "b2.msh"]),
    ("b_sketch_bloom_cov", [["sketch", "-b", "2K", "-c", "1.5", "-s", "100", "-o", "b3", "reads.fq"]], ["info", "-d", "b3.msh"]),
    ("b_sketch_bloom_k16", [["sketch", "-k", "16", "-b", "5K", "-s", "150", "-o", "b4", "reads.fq"]], ["info", "-d", "b4.msh"]),
    ("b_sketch_bloom_two_files", [["sketch", "-b", "1M", "-s", "300", "-o", "b5", "reads.fq", "g4.fa"]], ["info", "-d", "b5.msh"]),
    ("b_dist_bloom_query", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "-b", "4K", "a.msh", "reads.fq"]),
]

# Fifth batch: reads options outside `mash sketch` -- initFromFiles hands every query FILE to sketchFile
# with the reads-mode heap (Sketch.cpp:1156) and the estimated / -g genome size as its length (:1272-1282),
# which the p-values of dist / triangle then use.
READS_Q = [
    ("r_dist_reads_query", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "-r", "a.msh", "reads.fq"]),
    ("r_dist_reads_m2", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "-m", "2", "a.msh", "reads.fq"]),
    ("r_dist_reads_cov", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "-c", "1.02", "a.msh", "reads.fq"]),
    ("r_dist_reads_genome_size", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "-g", "25k", "a.msh", "reads.fq"]),
    ("r_dist_reads_both_sides", [], ["dist", "-r", "-s", "200", "reads.fq", "g1.fa", "reads.fq"]),
    ("r_triangle_reads", [], ["triangle", "-r", "-s", "200", "reads.fq", "g1.fa", "g3.fa"]),
    ("r_dist_reads_small_sketch", [], ["dist", "-r", "-s", "25", "g1.fa", "reads.fq", "g3.fa"]),            # p-values that show the length
    ("r_dist_genome_size_small_sketch", [], ["dist", "-g", "25k", "-s", "25", "g1.fa", "reads.fq", "g3.fa"]),
    ("r_triangle_reads_small_sketch", [], ["triangle", "-r", "-s", "25", "-E", "reads.fq", "g1.fa", "g3.fa"]),
]


# Sixth batch: -m 0.  multiplicityMinimum - 1 wraps (uint64_t), no pending count ever equals it, the heap
# admits nothing (MinHashHeap.cpp:96-118): an empty sketch, refused unless -g supplies a length
# (Sketch.cpp:1272-1314; the refusal itself is compared on the CPU in tests/test_cli.py).
M0 = [
    ("m0_sketch_genome_size", [["sketch", "-m", "0", "-g", "30k", "-s", "100", "-o", "m0", "reads.fq"]], ["info", "-d", "m0.msh"]),
    ("m0_sketch_cov", [["sketch", "-m", "0", "-c", "2", "-g", "1000", "-s", "50", "-o", "m0c", "reads.fq"]], ["info", "-t", "m0c.msh"]),
    ("m0_dist_genome_size", [["sketch", "-s", "300", "-o", "a", "g1.fa", "g2.fa", "g3.fa"]], ["dist", "-m", "0", "-g", "500", "a.msh", "reads.fq"]),
]

UNCONFIRMED = set()        # names of cases not yet replayed on a GPU (the m0_* cases were confirmed in round 3, gpurun C)


def main():
    if not os.path.exists(REFCLI):
        sys.exit("build the reference CLI first: make -C oracle refcli")
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(f"{OUT}/in")
    make_inputs(f"{OUT}/in")
    manifest = []
    for name, setup, cmd in CASES + EXTRA + C5 + BLOOM + READS_Q + M0:
        d = tempfile.mkdtemp(prefix="cligold_")
        for f in os.listdir(f"{OUT}/in"):
            shutil.copy(f"{OUT}/in/{f}", d)
        for s in setup:
            r = subprocess.run([REFCLI, *s], cwd=d, capture_output=True, timeout=120)
            assert r.returncode == 0, (name, s, r.stderr[-300:])
        r = subprocess.run([REFCLI, *cmd], cwd=d, capture_output=True, timeout=120)
        assert r.returncode == 0, (name, cmd, r.stderr[-300:])
        open(f"{OUT}/{name}.out", "wb").write(r.stdout)
        manifest.append({"name": name, "setup": setup, "cmd": cmd, "stdout_bytes": len(r.stdout),
                         "confirmed_on_gpu": name not in UNCONFIRMED})
        shutil.rmtree(d)
        print(f"{name:28s} {len(r.stdout):8d} bytes")
    json.dump(manifest, open(f"{OUT}/cases.json", "w"), indent=1)


if __name__ == "__main__":
    main()
