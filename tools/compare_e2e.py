#!/usr/bin/env python3
"""The compare commands end to end, ours and the reference CLI ON THE SAME .msh IN THE SAME RUN:

    python tools/compare_e2e.py [--n-big 20000] [--n-filter 100000] [--n-small 3000] [--threads 16]

  triangle        `mash triangle`            (Phylip matrix: every cell printed)
  triangle_edge   `mash triangle -E -d 0.05` (edge list of the close pairs)
  dist_filter     `mash dist -d 0.05`        (table against itself, close pairs only)

The reference CLI (oracle/_ref/mash-ref, `-p <threads>`) does ~2e6 pairs/s on 16 cores, so it is timed on the first
--n-small sketches of the same table (a few seconds); ours is timed on that sample too -- outputs compared byte for
byte, `speedup_vs_reference` = reference / ours ON THE SAME INPUT -- and on the full size, whose rate is set against
the reference's sample rate (`speedup_at_full_size`: the reference's cost per pair does not fall with n).
Wall time of the whole process (fork/exec to exit, .msh read, output formatted; the samples' output goes to files on
/tmp and is compared, the full-size output to /dev/null).
`run()` is what bench.py's `cli_e2e` object calls."""
import argparse, ctypes as C, json, os, shutil, subprocess, sys, tempfile, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MASH = os.path.join(ROOT, "mash_amd", "bin", "mash")
REF = os.path.join(ROOT, "oracle", "_ref", "mash-ref")


def write_msh(path, hashes, nhash, lengths, k=21, seed=42):
    lib = C.CDLL(os.path.join(ROOT, "tests", "libmshio.so"))
    lib.mshio_write_table.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_char_p]
    h = np.ascontiguousarray(hashes, dtype=np.uint64)
    nh = np.ascontiguousarray(nhash, dtype=np.uint32)
    ln = np.ascontiguousarray(lengths, dtype=np.uint64)
    rc = lib.mshio_write_table(path.encode(), h.ctypes.data, nh.ctypes.data, ln.ctypes.data, h.shape[0], h.shape[1], k, seed, b"g")
    if rc != 0:
        raise RuntimeError("mshio_write_table failed")


def timed(cmd, out_path, reps=1):
    best = None
    for _ in range(reps):
        with open(out_path, "wb") as fo:
            t0 = time.perf_counter()
            r = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
        if r.returncode != 0:
            raise RuntimeError(f"{cmd[:3]}: rc {r.returncode}: {r.stderr.decode()[-300:]}")
        best = dt if best is None else min(best, dt)
    return best


def same(a, b):
    if os.path.getsize(a) != os.path.getsize(b):
        return False
    with open(a, "rb") as fa, open(b, "rb") as fb:
        while True:
            x, y = fa.read(1 << 24), fb.read(1 << 24)
            if x != y:
                return False
            if not x:
                return True


def run(n_big=20000, n_filter=100000, n_small=3000, threads=16, table=None):
    """table: (hashes u64[n, s], nhash, lengths) numpy, at least max(n_big, n_filter) rows (default: the C3 generator)"""
    if table is None:
        import torch
        from workloads import synth_torch
        n = max(n_big, n_filter)
        h, nh, ln = synth_torch.clustered_sketch_table(n, 1000, clusters=max(1, n // 100), device="cuda")
        table = (h.cpu().numpy().view(np.uint64), nh.cpu().numpy().astype(np.uint32), ln.cpu().numpy().astype(np.uint64))
    hashes, nhash, lengths = table
    d = tempfile.mkdtemp(prefix="mash_cmp_")
    res = {"threads": threads, "sketch_size": int(hashes.shape[1]), "n_small": n_small}
    try:
        files = {}
        for tag, n in (("small", n_small), ("big", n_big), ("filter", n_filter)):
            n = min(n, hashes.shape[0])
            files[tag] = (os.path.join(d, f"{tag}.msh"), n)
            write_msh(files[tag][0], hashes[:n], nhash[:n], lengths[:n])
        have_ref = os.path.exists(REF)
        legs = [("triangle", lambda f: ["triangle", "-p", str(threads), f], "big"),
                ("triangle_edge", lambda f: ["triangle", "-p", str(threads), "-E", "-d", "0.05", f], "filter"),
                ("dist_filter", lambda f: ["dist", "-p", str(threads), "-d", "0.05", f, f], "filter")]
        for name, args, big in legs:
            r = {}
            fs, ns = files["small"]
            fb, nb = files[big]
            pairs_s = ns * (ns - 1) // 2 if name != "dist_filter" else ns * ns
            pairs_b = nb * (nb - 1) // 2 if name != "dist_filter" else nb * nb
            r["ours_small_s"] = round(timed([MASH] + args(fs), os.path.join(d, "o_small.txt"), reps=2), 4)
            if have_ref:
                r["ref_small_s"] = round(timed([REF] + args(fs), os.path.join(d, "r_small.txt")), 4)
                r["same_bytes_as_reference"] = same(os.path.join(d, "o_small.txt"), os.path.join(d, "r_small.txt"))
                r["speedup_vs_reference"] = round(r["ref_small_s"] / r["ours_small_s"], 2)
                r["ref_pairs_per_s"] = pairs_s / r["ref_small_s"]
            if name == "triangle":
                # the Phylip writer works from the sparse result (mash_main.cpp); the dense path on the same sample: same bytes
                env = dict(os.environ, MASH_AMD_DENSE_MATRIX="1")
                with open(os.path.join(d, "d_small.txt"), "wb") as fo:
                    t0 = time.perf_counter()
                    subprocess.run([MASH] + args(fs), stdout=fo, stderr=subprocess.PIPE, env=env, check=True)
                    r["ours_small_dense_path_s"] = round(time.perf_counter() - t0, 4)
                r["same_bytes_as_dense_path"] = same(os.path.join(d, "o_small.txt"), os.path.join(d, "d_small.txt"))
                ff, nf = files["filter"]
                if nf > nb:                                            # the whole collection: 2 bytes of text per pair and more
                    r["ours_full_s"] = round(timed([MASH] + args(ff), "/dev/null"), 4)
                    r["n_full"] = nf
                    r["ours_full_pairs_per_s"] = nf * (nf - 1) / 2 / r["ours_full_s"]
            r["ours_s"] = round(timed([MASH] + args(fb), "/dev/null"), 4)      # (a 20 000-row matrix is 1.3 GB of text: formatted, not kept)
            r["n"], r["pairs"] = nb, pairs_b
            r["ours_pairs_per_s"] = pairs_b / r["ours_s"]
            if have_ref:
                r["speedup_at_full_size"] = round(r["ours_pairs_per_s"] / r["ref_pairs_per_s"], 1)
            res[name] = r
        return res
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-big", type=int, default=20000)
    ap.add_argument("--n-filter", type=int, default=100000)
    ap.add_argument("--n-small", type=int, default=3000)
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    print(json.dumps(run(a.n_big, a.n_filter, a.n_small, a.threads)))
