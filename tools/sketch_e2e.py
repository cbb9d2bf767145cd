#!/usr/bin/env python3
"""`mash sketch` end to end, ours and the reference CLI ON THE SAME FILES IN THE SAME RUN:

    python tools/sketch_e2e.py [--genomes 12000] [--len 50000] [--threads 16] [--variants] [--reps 3]

FASTA files (80 columns, like NCBI's) -> .msh.  Wall time of the whole process as a caller sees it
(fork/exec to exit), best of --reps; with MASH_AMD_TIMING the stages: exec -> main, device context,
ingest + sketch, write, teardown, main -> exit.  The two .msh files must describe the same sketches
(`mash info -d` of both, byte for byte).  `run()` is what bench.py's `cli_e2e` object calls."""
import argparse, json, os, shutil, subprocess, sys, tempfile, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASH = os.path.join(ROOT, "mash_amd", "bin", "mash")
REF = os.path.join(ROOT, "oracle", "_ref", "mash-ref")


def make_set(d, genomes, length, seed=1):
    """`genomes` FASTA files of `length` bases: one random base sequence, 5 % of the positions re-drawn per genome."""
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    base = lut[rng.integers(0, 4, length)]
    names = []
    nl = np.full((length // 80, 1), 10, np.uint8)
    for g in range(genomes):
        seq = base.copy()
        idx = rng.integers(0, length, max(1, length // 20))
        seq[idx] = lut[rng.integers(0, 4, len(idx))]
        body = np.concatenate([seq[:length // 80 * 80].reshape(-1, 80), nl], axis=1).tobytes()
        if length % 80:
            body += seq[length // 80 * 80:].tobytes() + b"\n"
        fn = os.path.join(d, "g%05d.fa" % g)
        with open(fn, "wb") as f:
            f.write(b">g%05d synthetic\n" % g + body)
        names.append(fn)
    lst = os.path.join(d, "list.txt")
    open(lst, "w").write("\n".join(names) + "\n")
    return lst


def timed(cmd, cwd, env=None, reps=3):
    """best-of-reps wall time of one command + the stage times of the best run"""
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
        t1 = time.perf_counter()
        if r.returncode != 0:
            raise RuntimeError(f"{cmd[:4]}: rc {r.returncode}: {r.stderr.decode()[-400:]}")
        st = {}
        for ln in r.stderr.decode().splitlines():
            if not ln.startswith("timing:"):
                continue
            w = ln[7:].replace(";", " ").split()
            if w[0] in ("main_begin", "main_end"):
                st[w[0]] = float(w[1])
            elif w[0] == "of":
                st["in_library_sketch_call"] = float(w[3])
            else:
                for k in range(0, len(w) - 2, 3):
                    st[w[k]] = float(w[k + 1])
        if "main_begin" in st:                       # time.perf_counter and steady_clock are both CLOCK_MONOTONIC
            st["exec_to_main"] = round(st.pop("main_begin") - t0, 4)
            st["main_to_exit"] = round(t1 - st.pop("main_end"), 4)
        if best is None or t1 - t0 < best[0]:
            best = (t1 - t0, st)
    return round(best[0], 4), best[1]


def run(genomes=12000, length=50000, threads=16, reps=3, variants=False, ref_threads=()):
    d = tempfile.mkdtemp(prefix="mash_sk_")
    try:
        t0 = time.perf_counter()
        lst = make_set(d, genomes, length)
        res = {"set": f"{genomes} FASTA files x {length} bp (80 columns)", "bp": genomes * length, "threads": threads,
               "generate_s": round(time.perf_counter() - t0, 2), "reps": reps}
        env = dict(os.environ, MASH_AMD_TIMING="1")
        s, st = timed([MASH, "sketch", "-p", str(threads), "-l", "-o", "ours", lst], d, env, reps)
        res["ours_s"], res["ours_stages"] = s, st
        res["ours_bp_per_s"] = res["bp"] / s
        if variants:
            for tag, extra in (("p1", None), ("early_parse", {"MASH_AMD_EARLY_PARSE": "1"}), ("no_groups", {"MASH_AMD_NO_GROUPS": "1"}),
                               ("early_no_groups", {"MASH_AMD_EARLY_PARSE": "1", "MASH_AMD_NO_GROUPS": "1"}), ("slow_exit", {"MASH_AMD_SLOW_EXIT": "1"}),
                               ("one_batch", {"MASH_AMD_BATCH_BYTES": str(2 << 30)}), ("batch_16M", {"MASH_AMD_BATCH_BYTES": str(16 << 20)}),
                               ("batch_256M", {"MASH_AMD_BATCH_BYTES": str(256 << 20)}), ("p4", None), ("p8", None), ("p32", None), ("p64", None)):
                th = tag[1:] if tag[0] == "p" and tag[1:].isdigit() else str(threads)
                s, st = timed([MASH, "sketch", "-p", th, "-l", "-o", "v_" + tag, lst], d, dict(env, **(extra or {})), reps)
                res["ours_" + tag + "_s"], res["ours_" + tag + "_stages"] = s, st
                assert open(os.path.join(d, "v_" + tag + ".msh"), "rb").read() == open(os.path.join(d, "ours.msh"), "rb").read(), tag
        if os.path.exists(REF):
            for th in (threads, *ref_threads):
                s, _ = timed([REF, "sketch", "-p", str(th), "-l", "-o", "ref", lst], d, None, max(1, reps - 1))
                res[f"ref_p{th}_s"] = s
            res["ref_s"] = res[f"ref_p{threads}_s"]
            res["ref_bp_per_s"] = res["bp"] / res["ref_s"]
            res["speedup_vs_ref_same_threads"] = round(res["ref_s"] / res["ours_s"], 3)
            dump = lambda f: subprocess.run([MASH, "info", "-d", f], cwd=d, capture_output=True, check=True).stdout
            res["same_sketches_as_ref"] = dump("ours.msh") == dump("ref.msh")
        else:
            res["ref_s"] = None
        return res
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=12000)
    ap.add_argument("--len", type=int, default=50000)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--variants", action="store_true", help="also time ours with each ingest change switched off, -p 1 and -p 64")
    ap.add_argument("--ref-threads", type=int, nargs="*", default=[], help="further -p values for the reference CLI")
    a = ap.parse_args()
    print(json.dumps(run(a.genomes, a.len, a.threads, a.reps, a.variants, tuple(a.ref_threads))))
