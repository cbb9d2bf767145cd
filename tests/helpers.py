"""Shared test helpers (pure Python, CPU)."""
import gzip
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_fastx(path):
    """Minimal FASTA/FASTQ reader with kseq.h semantics (kseq.h:171-208): header
    char > or @, name up to first whitespace, comment = rest of line, sequence =
    all isgraph bytes up to the next >, @ or + line start.  Returns
    [(name, comment, seq bytes)]."""
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rb") as f:
        data = f.read()
    recs = []
    lines = data.split(b"\n")
    i = 0
    n = len(lines)
    while i < n:
        ln = lines[i]
        if not ln or ln[:1] not in (b">", b"@"):
            i += 1
            continue
        hdr = ln[1:]
        parts = hdr.split(None, 1)
        name = parts[0] if parts else b""
        comment = parts[1] if len(parts) > 1 else b""
        i += 1
        seq = bytearray()
        while i < n and lines[i][:1] not in (b">", b"@", b"+"):
            seq += bytes(c for c in lines[i] if 33 <= c <= 126)
            i += 1
        if i < n and lines[i][:1] == b"+":
            i += 1
            q = 0
            while i < n and q < len(seq):
                q += len(lines[i])
                i += 1
        recs.append((name, comment, bytes(seq)))
    return recs


def round_robin(lists):
    """Interleave records of several files the way sketchFile does (Sketch.cpp:1200-1270)."""
    out = []
    iters = [iter(l) for l in lists]
    while iters:
        nxt = []
        for it in iters:
            try:
                out.append(next(it))
                nxt.append(it)
            except StopIteration:
                pass
        iters = nxt
    return out


def fmt_g(x):
    """ostream << double with default precision (6 significant digits, %g style)."""
    return "%g" % x


def load_golden_genomes():
    z = np.load(os.path.join(GOLDEN, "genomes_sketches.npz"))
    return z["hashes"], z["lengths"], [str(s) for s in z["names"]]


def load_golden_reads():
    z = np.load(os.path.join(GOLDEN, "reads_sketch.npz"))
    return z["hashes"], int(z["length"]), str(z["comment"])


def load_ref_sketch_vectors(name="ref_sketch_vectors.npz"):
    z = np.load(os.path.join(GOLDEN, name))
    cfgs = json.loads(str(z["cfgs"]))
    out = []
    for c in cfgs:
        i = c["idx"]
        bases = z[f"bases_{i}"].tobytes()
        lens = z[f"reclen_{i}"]
        recs, o = [], 0
        for l in lens:
            recs.append(bases[o:o + int(l)])
            o += int(l)
        out.append((c, recs, z[f"hashes_{i}"], z[f"counts_{i}"]))
    return out
