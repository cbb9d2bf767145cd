#!/usr/bin/env python3
"""End-to-end timing of the CLI on synthetic FASTA files (host ingest + GPU + output):
    python tools/cli_e2e.py [--genomes 500] [--len 1000000] [--threads 16]
Prints one JSON object with wall times of `mash sketch`, `mash triangle`, `mash dist`."""
import argparse, json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASH = os.path.join(ROOT, "mash_amd", "bin", "mash")

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=500)
    ap.add_argument("--len", type=int, default=1_000_000)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--only-edge", action="store_true", help="time only the thresholded edge-list run")
    ap.add_argument("--only-sketch", action="store_true", help="time only `mash sketch`")
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="mash_e2e_")
    os.environ["MASH_AMD_TIMING"] = "1"
    rng = np.random.default_rng(1)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    base = lut[rng.integers(0, 4, a.len)]
    names = []
    t0 = time.perf_counter()
    for g in range(a.genomes):
        seq = base.copy()
        idx = rng.integers(0, a.len, a.len // (20 if g % 10 else 2))     # 5 % (or 50 %) of positions re-drawn
        seq[idx] = lut[rng.integers(0, 4, len(idx))]
        fn = os.path.join(d, "g%04d.fa" % g)
        with open(fn, "wb") as f:
            f.write(b">g%04d synthetic\n" % g)
            lines = seq.tobytes()
            f.write(b"\n".join(lines[i:i + 80] for i in range(0, len(lines), 80)) + b"\n")
        names.append(fn)
    res = {"genomes": a.genomes, "len": a.len, "generate_s": time.perf_counter() - t0}
    lst = os.path.join(d, "list.txt")
    open(lst, "w").write("\n".join(names) + "\n")

    def run(tag, *cmd, out=None):
        t = time.perf_counter()
        with open(out or os.devnull, "wb") as fo:
            r = subprocess.run([MASH, *cmd], stdout=fo, stderr=subprocess.PIPE, cwd=d)
        assert r.returncode == 0, r.stderr.decode()[-500:]
        res[tag] = time.perf_counter() - t
        stages = [ln[len("timing:"):].strip() for ln in r.stderr.decode().splitlines() if ln.startswith("timing:")]
        if stages:
            res[tag + "_stages"] = " | ".join(stages)

    run("sketch_p1_s", "sketch", "-l", "-o", "seq", lst)
    run("sketch_pN_s", "sketch", "-p", str(a.threads), "-l", "-o", "par", lst)
    assert open(os.path.join(d, "seq.msh"), "rb").read() == open(os.path.join(d, "par.msh"), "rb").read()
    if a.only_sketch:
        print(json.dumps(res))
        subprocess.run(["rm", "-rf", d])
        return
    if a.only_edge:
        run("triangle_edge_d0.2_s", "triangle", "-E", "-d", "0.2", "par.msh", out=os.path.join(d, "edge.txt"))
        os.environ["MASH_AMD_EMIT_THREADS"] = "1"
        run("triangle_edge_d0.2_1thread_s", "triangle", "-E", "-d", "0.2", "par.msh", out=os.path.join(d, "edge1.txt"))
        res["edge_lines"] = sum(1 for _ in open(os.path.join(d, "edge.txt")))
        assert open(os.path.join(d, "edge.txt"), "rb").read() == open(os.path.join(d, "edge1.txt"), "rb").read()
        print(json.dumps(res))
        subprocess.run(["rm", "-rf", d])
        return
    run("triangle_s", "triangle", "par.msh", out=os.path.join(d, "tri.txt"))
    run("triangle_edge_d0.2_s", "triangle", "-E", "-d", "0.2", "par.msh", out=os.path.join(d, "edge.txt"))
    if a.genomes <= 5000:                                    # n^2 text lines: keep the file bounded
        run("dist_s", "dist", "par.msh", "par.msh", out=os.path.join(d, "dist.txt"))
        res["dist_lines"] = sum(1 for _ in open(os.path.join(d, "dist.txt")))
    else:
        run("dist_table_s", "dist", "-t", "par.msh", "par.msh", out=os.path.join(d, "dist.txt"))
    res["triangle_bytes"] = os.path.getsize(os.path.join(d, "tri.txt"))
    os.environ["MASH_AMD_EMIT_THREADS"] = "1"
    run("triangle_1thread_format_s", "triangle", "par.msh", out=os.path.join(d, "tri1.txt"))
    del os.environ["MASH_AMD_EMIT_THREADS"]
    assert open(os.path.join(d, "tri.txt"), "rb").read() == open(os.path.join(d, "tri1.txt"), "rb").read()
    res["edge_lines"] = sum(1 for _ in open(os.path.join(d, "edge.txt")))
    res["bases_per_s_sketch_pN"] = a.genomes * a.len / res["sketch_pN_s"]
    print(json.dumps(res))
    subprocess.run(["rm", "-rf", d])

if __name__ == "__main__":
    main()
