#!/usr/bin/env python3
"""`mash sketch -r` end to end on one large read file: the constant-memory route (reads session, default)
against the route that keeps the read set in HBM (MASH_AMD_READS_RESIDENT=1) and the reference CLI.

    python tools/reads_e2e.py [--reads 4000000] [--len 150] [--genome 5000000]

One FASTA file of `reads` reads sampled from a random genome (both strands, 0.5 % substitutions).  Wall
times of the whole process, peak RSS of EACH child on its own (os.wait4: the rusage of that one process -- RUSAGE_CHILDREN
is a maximum over all children so far and made every run report the largest), same sketches (info -d) everywhere."""
import argparse, json, os, resource, shutil, subprocess, sys, tempfile, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASH = os.path.join(ROOT, "mash_amd", "bin", "mash")
REF = os.path.join(ROOT, "oracle", "_ref", "mash-ref")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--len", type=int, default=150)
    ap.add_argument("--genome", type=int, default=5_000_000)
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="mash_reads_")
    try:
        rng = np.random.default_rng(5)
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        comp = np.zeros(256, np.uint8)
        comp[lut] = lut[::-1]
        g = lut[rng.integers(0, 4, a.genome)]
        fn = os.path.join(d, "reads.fa")
        t0 = time.perf_counter()
        with open(fn, "wb") as f:
            for o in range(0, a.reads, 500_000):
                n = min(500_000, a.reads - o)
                st = rng.integers(0, a.genome - a.len, n)
                r = g[st[:, None] + np.arange(a.len)[None, :]]
                flip = rng.random(n) < 0.5
                r[flip] = comp[r[flip][:, ::-1]]
                err = rng.random(r.shape) < 0.005
                r[err] = lut[rng.integers(0, 4, int(err.sum()))]
                rows = np.concatenate([np.full((n, 1), ord(">"), np.uint8), np.full((n, 1), ord("r"), np.uint8), np.full((n, 1), 10, np.uint8),
                                       r, np.full((n, 1), 10, np.uint8)], axis=1)
                f.write(rows.tobytes())
        res = {"reads": a.reads, "len": a.len, "bp": a.reads * a.len, "file_bytes": os.path.getsize(fn), "generate_s": round(time.perf_counter() - t0, 1)}

        def run(tag, exe, args, env=None):
            t = time.perf_counter()
            errf = os.path.join(d, tag + ".stderr")
            with open(errf, "wb") as fe:
                p = subprocess.Popen([exe, "sketch", *args, "-o", os.path.join(d, tag), fn], stdout=subprocess.DEVNULL, stderr=fe,
                                     env=dict(os.environ, **(env or {})))
                _, status, ru = os.wait4(p.pid, 0)                   # the rusage of THIS child alone
                p.returncode = os.waitstatus_to_exitcode(status)
            err = open(errf, "rb").read().decode()
            assert p.returncode == 0, err[-400:]
            res[tag + "_s"] = round(time.perf_counter() - t, 3)
            res[tag + "_maxrss_mb"] = round(ru.ru_maxrss / 1024)
            res[tag + "_stderr"] = [l for l in err.splitlines() if l.startswith("Estimated")]

        dump = lambda tag: subprocess.run([MASH, "info", "-d", os.path.join(d, tag + ".msh")], capture_output=True, check=True).stdout
        for opts, name in ((["-r"], "r"), (["-r", "-m", "2"], "m2")):
            run("session_" + name, MASH, opts)
            run("resident_" + name, MASH, opts, {"MASH_AMD_READS_RESIDENT": "1"})
            res["same_" + name] = dump("session_" + name) == dump("resident_" + name)
            if os.path.exists(REF):
                run("ref_" + name, REF, opts)
                res["same_as_ref_" + name] = dump("session_" + name).replace(b"session_", b"ref_") == dump("ref_" + name) or \
                    dump("session_" + name).split(b'"hashes"')[1] == dump("ref_" + name).split(b'"hashes"')[1]
        print(json.dumps(res))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
