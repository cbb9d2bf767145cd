#!/usr/bin/env python3
"""Differential fuzz of the CLI against the reference CLI (oracle/_ref/mash-ref, built from the
reference's own sources; it travels to the GPU box with the snapshot).  Random command lines from a
small grammar over the committed inputs of tests/golden/cli/in -- sketching modes and reads options,
dist / triangle / screen / paste / info with their option variants -- are run through BOTH binaries
in separate scratch directories; exit codes and stdout must agree (stderr is shown on a difference).

    python tests/fuzz_cli.py [--n 200] [--seed 1] [--seconds 240]      # on a GPU box

Prints one line per differing case and a summary; exit code 1 if anything differed.
(Test infrastructure: it lives under tests/ because it executes oracle/_ref.)"""
import argparse, os, random, shutil, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IN = os.path.join(ROOT, "tests", "golden", "cli", "in")
OURS = os.path.join(ROOT, "mash_amd", "bin", "mash")
REF = os.path.join(ROOT, "oracle", "_ref", "mash-ref")

DNA = ["g1.fa", "g2.fa", "g3.fa", "g4.fa", "g5.fa.gz", "multi.fa"]
READS = ["reads.fq"]
EDGE = ["lowc.fa", "tiny.fa", "empty.fa", "allN.fa", "mixed.fq"]       # written by edge_inputs() into every scratch directory


def edge_inputs(d):
    """inputs for the corners: low-complexity sequence (tandem repeats, homopolymers: the same few k-mers
    over and over), records shorter than any k, an empty file, a record of N only, a FASTQ whose
    records are partly too short"""
    r = random.Random(99)
    unit = "".join(r.choice("ACGT") for _ in range(23))
    lowc = unit * 900 + "A" * 3000 + "AC" * 2500 + "".join(r.choice("ACGT") for _ in range(400)) + unit[:11] * 700
    with open(os.path.join(d, "lowc.fa"), "w") as f:
        f.write(">lowc tandem repeats\n")
        for i in range(0, len(lowc), 80):
            f.write(lowc[i:i + 80] + "\n")
    with open(os.path.join(d, "tiny.fa"), "w") as f:
        f.write(">t1\nACG\n>t2\nACGTAC\n>t3\n\n")
    open(os.path.join(d, "empty.fa"), "w").close()
    with open(os.path.join(d, "allN.fa"), "w") as f:
        f.write(">n only\n" + "N" * 500 + "\n")
    with open(os.path.join(d, "mixed.fq"), "w") as f:
        for i in range(60):
            l = 10 if i % 4 == 0 else 90
            seq = "".join(r.choice("ACGT") for _ in range(l))
            f.write("@m%d\n%s\n+\n%s\n" % (i, seq, "I" * l))


def pick(rng, xs, lo=1, hi=None):
    hi = hi or len(xs)
    return rng.sample(xs, rng.randint(lo, min(hi, len(xs))))


def split_stdin(args):
    """a trailing "<file" argument means: run with that file on stdin"""
    if args and args[-1].startswith("<"):
        return args[:-1], args[-1][1:]
    return args, None


def sketch_opts(rng, allow_reads=True, protein=False):
    o = []
    if protein:
        o += ["-a"] if rng.random() < 0.7 else ["-z", "ACDEFGHIKLMNPQRSTVWY"]
        if rng.random() < 0.6:
            o += ["-k", str(rng.choice([3, 5, 7, 9, 12]))]
    else:
        if rng.random() < 0.6:
            o += ["-k", str(rng.choice([4, 7, 11, 15, 16, 17, 21, 27, 31, 32]))]
        if rng.random() < 0.2:
            o += ["-n"]
        if rng.random() < 0.15:
            o += ["-Z"]
        if rng.random() < 0.08:
            o += ["-z", rng.choice(["ACGTN", "ACGTacgt", "AC"])]
    if rng.random() < 0.7:
        o += ["-s", str(rng.choice([1, 2, 10, 50, 100, 300, 1000, 2500]))]
    if rng.random() < 0.2:
        o += ["-S", str(rng.choice([0, 1, 7, 1000, 4294967295]))]
    if rng.random() < 0.2:
        o += ["-p", str(rng.choice([1, 2, 5]))]
    if rng.random() < 0.1:
        o += ["-w", rng.choice(["0.5", "0.001", "1"])]
    if allow_reads and not protein and rng.random() < 0.35:
        kind = rng.choice(["r", "m", "c", "b", "g", "mc", "bc", "rg"])
        if "r" in kind:
            o += ["-r"]
        if "m" in kind:
            o += ["-m", str(rng.choice([1, 2, 3]))]
        if "c" in kind:
            o += ["-c", rng.choice(["1.05", "1.5", "3", "40"])]
        if "b" in kind:
            o += ["-b", rng.choice(["7", "300", "3K", "2M"])]
        if "g" in kind:
            o += ["-g", rng.choice(["10k", "123456", "2M"])]
    return o


def gen_case(rng, idx):
    """-> (setup command lists, final command list)"""
    kind = rng.choice(["sketch", "sketch", "dist", "dist", "triangle", "screen", "paste", "info"])
    protein = kind in ("sketch", "dist", "triangle") and rng.random() < 0.12
    files = ["prot.fa"] if protein else None
    if kind == "sketch":
        o = sketch_opts(rng, protein=protein)
        reads_mode = any(x in o for x in ("-r", "-m", "-c", "-b", "-g"))
        if rng.random() < 0.25 and not reads_mode:
            o += ["-i"]
        if rng.random() < 0.15 and not reads_mode:
            o += ["-M"]
        if rng.random() < 0.15:
            o += ["-I", "my id"]
        if rng.random() < 0.15:
            o += ["-C", "a comment"]
        if files is None:
            files = pick(rng, READS + DNA[:2], 1, 2) if reads_mode and rng.random() < 0.8 else pick(rng, DNA + READS, 1, 4)
            if rng.random() < 0.25:                          # corner inputs, alone or among the others
                files = pick(rng, EDGE, 1, 2) + (files[:1] if rng.random() < 0.5 else [])
                rng.shuffle(files)
        if rng.random() < 0.1 and not protein and not reads_mode:
            o += ["-l"]
            files = ["list.txt"]
        r = rng.random()
        if r < 0.12 and files[0] not in ("list.txt",):
            # no -o: the output is named after the first input (CommandSketch.cpp:137-150)
            return [["sketch", *o, *files]], ["info", rng.choice(["-d", "-t"]), files[0] + ".msh"]
        if r < 0.2 and "-l" not in o:
            # first input from stdin
            return [["sketch", *o, "-o", "out", "-", *files[1:], "<" + files[0]]], ["info", rng.choice(["-d", "-t"]), "out.msh"]
        setup = [["sketch", *o, "-o", "out", *files]]
        return setup, ["info", rng.choice(["-d", "-t", "-H", "-c"]), "out.msh"]
    if kind == "info":
        o = sketch_opts(rng, allow_reads=False)
        o = [x for i, x in enumerate(o) if x not in ("-p", "-w") and (i == 0 or o[i - 1] not in ("-p", "-w"))]
        setup = [["sketch", *o, *(["-M"] if rng.random() < 0.5 else []), "-o", "out", *pick(rng, DNA, 1, 3)]]
        return setup, ["info", rng.choice(["-d", "-t", "-H", "-c"]), "out.msh"]
    if kind == "paste":
        s = str(rng.choice([10, 100, 300]))
        k = str(rng.choice([16, 21]))
        setup = [["sketch", "-s", s, "-k", k, "-o", "p%d" % i, *pick(rng, DNA, 1, 2)] for i in range(rng.randint(2, 3))]
        setup.append(["paste", "joined", *["p%d.msh" % i for i in range(len(setup))]])
        return setup, rng.choice([["info", "-t", "joined.msh"], ["info", "-d", "joined.msh"], ["dist", "joined.msh", "joined.msh"]])
    if kind == "screen":
        k = rng.choice([16, 21])
        setup = [["sketch", "-s", str(rng.choice([100, 300, 1000])), "-k", str(k), "-o", "db", *pick(rng, DNA[:5], 2, 5)]]
        if rng.random() < 0.25:                              # amino-acid database: the mixture is translated in six frames
            setup = [["sketch", "-a", "-i", "-k", str(rng.choice([5, 7, 9])), "-s", str(rng.choice([50, 200, 1000])), "-o", "db", "prot.fa"]]
        o = []
        if rng.random() < 0.4:
            o += ["-w"]
        if rng.random() < 0.3:
            o += ["-i", rng.choice(["0", "0.5", "0.9", "-1"])]
        if rng.random() < 0.3:
            o += ["-v", rng.choice(["1", "0.01", "1e-10"])]
        if rng.random() < 0.3:
            o += ["-p", str(rng.choice([1, 3]))]
        return setup, ["screen", *o, "db.msh", *pick(rng, READS + DNA[:4] + (EDGE if rng.random() < 0.3 else []), 1, 2)]
    # dist / triangle
    o = sketch_opts(rng, protein=protein)
    reads_mode = any(x in o for x in ("-r", "-m", "-c", "-b", "-g"))
    if rng.random() < 0.2 and not reads_mode:
        o += ["-i"]
    if rng.random() < 0.25:
        o += ["-d", rng.choice(["0.05", "0.2", "1", "0"])]
    if rng.random() < 0.25:
        o += ["-v", rng.choice(["1", "1e-5", "1e-100", "0"])]
    pool = ["prot.fa"] if protein else (DNA + READS if reads_mode or rng.random() < 0.2 else DNA)
    if not protein and rng.random() < 0.2:
        pool = pool + EDGE
    if kind == "triangle":
        if rng.random() < 0.4:
            o += ["-E"]
        if rng.random() < 0.2:
            o += ["-C"]
        if rng.random() < 0.15 and not protein:
            return [], ["triangle", *o, "-l", "list.txt", *pick(rng, ["list.txt"], 0, 1)]
        return [], ["triangle", *o, *pick(rng, pool, 1 if protein or "-i" in o else 2, 4)]
    if rng.random() < 0.3:
        o += ["-t"]
    if rng.random() < 0.15:
        o += ["-C"]
    if rng.random() < 0.4 and not protein:
        # reference given as a sketch: k, s, alphabet are inherited; -k / -n / -a / -z are then refused
        so = [x for x in sketch_opts(rng, allow_reads=False) if True]
        setup = [["sketch", *so, "-o", "refdb", *pick(rng, DNA, 1, 3)]]
        o = [x for i, x in enumerate(o) if x not in ("-k", "-n", "-z", "-Z", "-S") and (i == 0 or o[i - 1] not in ("-k", "-z", "-S"))]
        return setup, ["dist", *o, "refdb.msh", *pick(rng, pool, 1, 3)]
    if rng.random() < 0.15 and not protein:
        return [], ["dist", *o, "-l", *pick(rng, pool, 1, 1), "list.txt"]
    if rng.random() < 0.1 and not protein:
        return [], ["dist", *o, *pick(rng, pool, 1, 1), "-", "<" + rng.choice(pool)]       # query from stdin
    return [], ["dist", *o, *pick(rng, pool, 1, 1), *pick(rng, pool, 1, 3)]


def run_one(binary, args, d, env):
    args, stdin_file = split_stdin(args)
    if stdin_file is None:
        return subprocess.run([binary, *args], cwd=d, capture_output=True, timeout=300, env=env, stdin=subprocess.DEVNULL)
    with open(os.path.join(d, stdin_file), "rb") as f:
        return subprocess.run([binary, *args], cwd=d, capture_output=True, timeout=300, env=env, stdin=f)


def run_seq(binary, setup, cmd, d, env=None):
    for s in setup:
        r = run_one(binary, s, d, env)
        if r.returncode != 0:
            return ("setup", s, r.returncode, b"", r.stderr)
    r = run_one(binary, cmd, d, env)
    return ("cmd", cmd, r.returncode, r.stdout, r.stderr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--env", action="append", default=[], help="KEY=VALUE for our binary only, e.g. MASH_GPU_DEVICES=0,0 (the sharded paths)")
    a = ap.parse_args()
    ours_env = dict(os.environ, **dict(kv.split("=", 1) for kv in a.env)) if a.env else None
    for b in (OURS, REF):
        if not os.path.exists(b):
            sys.exit("missing " + b)
    rng = random.Random(a.seed)
    t0 = time.time()
    bad = same = refused = crashed = 0
    for idx in range(a.n):
        if time.time() - t0 > a.seconds:
            break
        setup, cmd = gen_case(rng, idx)
        res = []
        for binary in (REF, OURS):
            d = tempfile.mkdtemp(prefix="clifuzz_")
            for f in os.listdir(IN):
                shutil.copy(os.path.join(IN, f), d)
            edge_inputs(d)
            try:
                res.append(run_seq(binary, setup, cmd, d, ours_env if binary == OURS else None))
            except subprocess.TimeoutExpired:
                res.append(("timeout", cmd, -999, b"", b""))
            shutil.rmtree(d)
        (ws, wc, wrc, wout, werr), (gs, gc, grc, gout, gerr) = res
        ok = ws == gs and wrc == grc and wout == gout
        if wrc < 0 and wrc != -999:
            # the reference died from a signal (e.g. SIGFPE in `dist` when the reference side holds no
            # sketch: pairCount % 0, CommandDistance.cpp:196-214): nothing to be identical to
            crashed += 1
            print("REF-CRASH #%d signal %d cmd=%s (ours: rc %d)" % (idx, -wrc, cmd, grc))
            continue
        if ok:
            same += 1
            refused += wrc != 0
            continue
        bad += 1
        print("DIFF #%d setup=%s cmd=%s" % (idx, setup, cmd))
        print("   ref : stage %s rc %d stdout %d B; stderr tail %r" % (ws, wrc, len(wout), werr[-200:]))
        print("   ours: stage %s rc %d stdout %d B; stderr tail %r" % (gs, grc, len(gout), gerr[-200:]))
        if wrc == grc == 0:
            x, y = wout.splitlines(), gout.splitlines()
            i = next((k for k in range(min(len(x), len(y))) if x[k] != y[k]), min(len(x), len(y)))
            print("   first differing line %d: ref %r ours %r" % (i, x[i][:140] if i < len(x) else None, y[i][:140] if i < len(y) else None))
        sys.stdout.flush()
    print("cases run: %d  identical: %d (of which refused by both: %d)  differing: %d  reference crashed: %d  [seed %d, %.0f s]"
          % (same + bad + crashed, same, refused, bad, crashed, a.seed, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
