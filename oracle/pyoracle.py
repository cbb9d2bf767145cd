"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE — see mash_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (mash_amd) never does.

`Oracle()` wraps oracle/libmash_oracle.so (the plain-C restatement);
`Oracle(ref=True)` wraps oracle/_ref/libmash_ref.so (the reference's own
objects, built from /root/reference by `make -C oracle ref`) and exposes the
same calls under the same names.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OracleParams(C.Structure):
    _fields_ = [
        ("kmer_size", C.c_int),
        ("sketch_size", C.c_uint64),
        ("seed", C.c_uint32),
        ("use64", C.c_int),
        ("noncanonical", C.c_int),
        ("preserve_case", C.c_int),
        ("alphabet", C.c_uint8 * 256),
        ("min_copies", C.c_uint32),
        ("target_cov", C.c_double),
        ("bloom_bytes", C.c_uint64),
    ]


class OraclePair(C.Structure):
    _fields_ = [
        ("numer", C.c_uint64),
        ("denom", C.c_uint64),
        ("distance", C.c_double),
        ("p_value", C.c_double),
        ("pass_", C.c_int),
    ]


def build(ref=False, cli=False):
    """Compile the oracle (and, where /root/reference exists, oracle/_ref; cli: also the reference CLI,
    plain and bound to libmashgpu.so -- the latter needs mash_amd/libmashgpu.so to be built)."""
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    if ref and os.path.isdir("/root/reference/src/mash"):
        subprocess.run(["make", "-C", _HERE, "-s", "ref"], check=True)
        if cli:
            subprocess.run(["make", "-C", _HERE, "-s", "refcli"], check=True)
            if os.path.exists(os.path.join(os.path.dirname(_HERE), "mash_amd", "libmashgpu.so")):
                subprocess.run(["make", "-C", _HERE, "-s", "refcli-gpu"], check=True)


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libmash_ref.so"))


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _f64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Oracle:
    def __init__(self, ref=False):
        self.is_ref = ref
        if ref:
            path = os.path.join(_HERE, "_ref", "libmash_ref.so")
            self.prefix = "ref_"
        else:
            path = os.path.join(_HERE, "libmash_oracle.so")
            if not os.path.exists(path):
                build()
            self.prefix = "oracle_"
        self.lib = C.CDLL(path)
        L = self.lib
        L.oracle_set_alphabet.argtypes = [C.POINTER(OracleParams), C.c_char_p]
        L.oracle_binomial_q.restype = C.c_double
        L.oracle_binomial_q.argtypes = [C.c_uint, C.c_double, C.c_uint]
        L.oracle_p_value.restype = C.c_double
        L.oracle_p_value.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_uint64]
        L.oracle_triangle.restype = C.c_uint64
        L.oracle_bloom_hash.restype = C.c_uint32
        L.oracle_bloom_hash.argtypes = [C.c_uint64, C.c_int]
        gh = getattr(L, self.prefix + "get_hash")
        gh.restype = C.c_uint64
        gh.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_int]
        self._get_hash = gh
        sk = getattr(L, self.prefix + "sketch_records")
        sk.restype = C.c_int
        self._sketch = sk
        self._compare = getattr(L, self.prefix + "compare_sketches")
        self._compare.restype = None
        self._translate = getattr(L, self.prefix + "translate")
        self._translate.restype = None
        self._translate.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64]
        if ref:
            L.ref_table_new.restype = C.c_void_p
            L.ref_table_free.argtypes = [C.c_void_p]
            L.ref_triangle.restype = C.c_uint64

    # -- parameters --------------------------------------------------------
    def params(self, k=21, s=1000, seed=42, alphabet="ACGT", noncanonical=False,
               preserve_case=False, min_copies=1, target_cov=0.0, bloom_bytes=0):
        p = OracleParams()
        p.min_copies = min_copies
        p.target_cov = target_cov
        p.bloom_bytes = bloom_bytes
        p.kmer_size = k
        p.sketch_size = s
        p.seed = seed
        p.noncanonical = int(noncanonical)
        p.preserve_case = int(preserve_case)
        self.lib.oracle_set_alphabet(C.byref(p), alphabet.encode())
        return p

    # -- hashing -----------------------------------------------------------
    def get_hash(self, kmer: bytes, seed=42, use64=True):
        return int(self._get_hash(kmer, len(kmer), seed, int(use64)))

    # -- translation (mash screen, amino-acid queries) ---------------------------
    def translate(self, seq: bytes) -> bytes:
        """translate(), CommandScreen.cpp:617-623: one amino acid per full codon of `seq`"""
        n = len(seq) // 3
        out = C.create_string_buffer(max(n, 1))
        self._translate(seq, out, n)
        return out.raw[:n]

    def six_frames(self, seq: bytes, preserve_case=False):
        """the six translated strings hashSequence walks for one chunk (CommandScreen.cpp:498-531):
        frames 0..2 of the (upper-cased) chunk, then frames 0..2 of its reverse complement"""
        if not preserve_case:
            seq = bytes(b - 32 if 96 < b < 123 else b for b in seq)
        comp = {65: 84, 67: 71, 71: 67, 84: 65}
        rc = bytes(comp.get(b, 78) for b in reversed(seq))      # non-ACGT never survives translation
        return [self.translate(src[f:]) for src in (seq, rc) for f in range(3)]

    # -- sketching ---------------------------------------------------------
    def sketch_records(self, records, p):
        """records: list of bytes. Returns (hashes u64[n], counts u32[n], length, set_size)."""
        bases = b"".join(records)
        off = np.zeros(len(records) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in records], dtype=np.uint64)
        s = int(p.sketch_size)
        hashes = np.zeros(s, dtype=np.uint64)
        counts = np.zeros(s, dtype=np.uint32)
        n = C.c_uint64(0)
        length = C.c_uint64(0)
        setsz = C.c_double(0)
        rc = self._sketch(C.c_char_p(bases), _u64p(off), C.c_uint64(len(records)), C.byref(p),
                          _u64p(hashes), _u32p(counts), C.byref(n), C.byref(length), C.byref(setsz))
        return hashes[: n.value].copy(), counts[: n.value].copy(), int(length.value), float(setsz.value), rc

    def sketch_reads(self, records, p):
        """reads mode with the -c early stop: (hashes, counts, set_size, records_used, multiplicity)"""
        bases = b"".join(records)
        off = np.zeros(len(records) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(r) for r in records], dtype=np.uint64)
        s = int(p.sketch_size)
        hashes = np.zeros(s, dtype=np.uint64)
        counts = np.zeros(s, dtype=np.uint32)
        n, length, used = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        setsz, mult = C.c_double(0), C.c_double(0)
        fn = getattr(self.lib, self.prefix + "sketch_reads")
        fn.restype = C.c_int
        fn(C.c_char_p(bases), _u64p(off), C.c_uint64(len(records)), C.byref(p), _u64p(hashes), _u32p(counts),
           C.byref(n), C.byref(length), C.byref(setsz), C.byref(used), C.byref(mult))
        return hashes[: n.value].copy(), counts[: n.value].copy(), float(setsz.value), int(used.value), float(mult.value)

    def kmer_hashes(self, bases, rec_off, p):
        """Hash of every valid k-mer of every record (C restatement only): bases = writable uint8 array
        (uppercased in place), rec_off = uint64[nrec + 1]."""
        from ctypes import CDLL
        lib = self.lib if not self.is_ref else CDLL(os.path.join(_HERE, "libmash_oracle.so"))
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        lens = np.diff(rec_off).astype(np.int64)
        k = int(p.kmer_size)
        cap = int(np.maximum(lens - k + 1, 0).sum())
        out = np.zeros(max(cap, 1), dtype=np.uint64)
        lib.oracle_kmer_hashes.restype = C.c_uint64
        n = lib.oracle_kmer_hashes(bases.ctypes.data_as(C.c_char_p), _u64p(rec_off), C.c_uint64(len(rec_off) - 1), C.byref(p), _u64p(out))
        return out[: int(n)]

    # -- comparing ---------------------------------------------------------
    def compare(self, a, b, len_a, len_b, s, k, kmer_space, max_d=-1.0, max_p=-1.0, use64=True):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        out = OraclePair()
        args = [C.byref(out), _u64p(a), C.c_uint64(len(a)), C.c_uint64(len_a),
                _u64p(b), C.c_uint64(len(b)), C.c_uint64(len_b),
                C.c_uint64(s), C.c_int(k), C.c_double(kmer_space), C.c_double(max_d), C.c_double(max_p)]
        if self.is_ref:
            args.append(C.c_int(int(use64)))
        self._compare(*args)
        return out

    def p_value(self, x, len_ref, len_qry, kmer_space, s):
        return float(self.lib.oracle_p_value(x, len_ref, len_qry, kmer_space, s))

    def binomial_q(self, k, p, n):
        return float(self.lib.oracle_binomial_q(k, p, n))

    # -- one table shared by many callers (bench.py's cpu_baseline: the table is materialised ONCE and only read) ----------
    def table_open(self, table, nhash, lengths):
        """The reference's vector<Sketch::Reference> of a dense table (ref) or the arrays themselves (port), for triangle_run."""
        table = np.ascontiguousarray(table, dtype=np.uint64)
        nhash = np.ascontiguousarray(nhash, dtype=np.uint32)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint64)
        n, s = table.shape
        h = None
        if self.is_ref:
            h = self.lib.ref_table_new(_u64p(table), _u32p(nhash), _u64p(lengths), C.c_uint64(n), C.c_uint64(s))
        return {"h": h, "table": table, "nhash": nhash, "lengths": lengths, "n": n, "s": s}

    def table_close(self, t):
        if self.is_ref and t["h"]:
            self.lib.ref_table_free(C.c_void_p(t["h"]))
            t["h"] = None

    def triangle_run(self, t, row_begin, row_end, k, kmer_space):
        """compareSketches incl. distance and p-value for rows [row_begin, row_end) of an open table, results computed and
        dropped (nothing is allocated or copied per call: any number of threads may run on one table).  Returns the pairs."""
        if self.is_ref:
            return int(self.lib.ref_triangle(C.c_void_p(t["h"]), C.c_uint64(row_begin), C.c_uint64(row_end), C.c_int(k), C.c_double(kmer_space),
                                             None, None, None, None))
        return int(self.lib.oracle_triangle(_u64p(t["table"]), _u32p(t["nhash"]), _u64p(t["lengths"]), C.c_uint64(t["n"]), C.c_uint64(t["s"]),
                                            C.c_uint64(row_begin), C.c_uint64(row_end), C.c_int(k), C.c_double(kmer_space), C.c_int(1),
                                            None, None, None, None))

    def triangle(self, table, nhash, lengths, row_begin, row_end, k, kmer_space, stats=False):
        """Rows [row_begin,row_end) of the lower triangle, reference order.
        Returns (numer u32[], denom u32[], dist f64[]|None, pval f64[]|None)."""
        table = np.ascontiguousarray(table, dtype=np.uint64)
        nhash = np.ascontiguousarray(nhash, dtype=np.uint32)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint64)
        n, s = table.shape
        row_end = min(row_end, n)
        npairs = sum(range(row_begin, row_end))
        numer = np.zeros(npairs, dtype=np.uint32)
        denom = np.zeros(npairs, dtype=np.uint32)
        dist = np.zeros(npairs, dtype=np.float64) if stats else None
        pval = np.zeros(npairs, dtype=np.float64) if stats else None
        if self.is_ref:
            t = self.lib.ref_table_new(_u64p(table), _u32p(nhash), _u64p(lengths), C.c_uint64(n), C.c_uint64(s))
            try:
                self.lib.ref_triangle(C.c_void_p(t), C.c_uint64(row_begin), C.c_uint64(row_end), C.c_int(k),
                                      C.c_double(kmer_space), _u32p(numer), _u32p(denom),
                                      _f64p(dist) if stats else None, _f64p(pval) if stats else None)
            finally:
                self.lib.ref_table_free(C.c_void_p(t))
        else:
            self.lib.oracle_triangle(_u64p(table), _u32p(nhash), _u64p(lengths), C.c_uint64(n), C.c_uint64(s),
                                     C.c_uint64(row_begin), C.c_uint64(row_end), C.c_int(k),
                                     C.c_double(kmer_space), C.c_int(int(stats)), _u32p(numer), _u32p(denom),
                                     _f64p(dist) if stats else None, _f64p(pval) if stats else None)
        return numer, denom, dist, pval
