"""ctypes binding of libmashgpu.so (include/mashgpu.h) — plumbing for tests and bench.py.

The product is the C ABI; this module only loads it.  There is no CPU fallback:
if the HIP library is missing or no GPU is visible, calls fail loudly.

Note for processes that also use torch on the GPU: call torch.cuda.init() BEFORE creating a
MashGpu context — torch wheels bundle their own HIP runtime and it must be the one that
initialises first (bench.py and the tests do this); the library itself never needs torch.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmashgpu.so")

MG_OK = 0
MG_ERR_NOMEM = -4
HASH_PAD = 0xFFFFFFFFFFFFFFFF
RECORD_SEP = 0x0A

EXPORTS = [
    "mg_device_count", "mg_ctx_create", "mg_ctx_destroy", "mg_last_error", "mg_ctx_set_stream", "mg_ctx_synchronize", "mg_ctx_set_async", "mg_ctx_set_option", "mg_ctx_trim",
    "mg_ctx_cu_count", "mg_params_init", "mg_sketch_host", "mg_sketch_dev", "mg_packed_bytes", "mg_packed_mask_bytes", "mg_pack_bases",
    "mg_sketch_host_packed", "mg_sketch_dev_packed", "mg_sketch_reads_host", "mg_sketch_begin", "mg_sketch_add",
    "mg_sketch_stage_capacity", "mg_sketch_stage", "mg_sketch_commit", "mg_sketch_end_sketch", "mg_sketch_pending", "mg_sketch_finish", "mg_sketch_session_free",
    "mg_reads_begin", "mg_reads_add_host", "mg_reads_finish", "mg_reads_reset", "mg_reads_free", "mg_table_upload",
    "mg_table_wrap_dev", "mg_table_free", "mg_table_invalidate", "mg_table_rows", "mg_table_sketch_size",
    "mg_compare_tri_dev", "mg_compare_tri_host", "mg_compare_rect_dev", "mg_compare_rect_host",
    "mg_compare_tri_filter_host", "mg_compare_rect_filter_host", "mg_compare_tri_sparse_host", "mg_compare_rect_sparse_host", "mg_expand_tri_sparse",
    "mg_finish_tri_host", "mg_finish_rect_host", "mg_distance", "mg_p_value",
    "mg_finish_tri_dev", "mg_finish_rect_dev", "mg_compare_tri_pairs_host", "mg_compare_rect_pairs_host",
    "mg_compare_tri_results_host", "mg_compare_rect_results_host",
    "mg_prof_enable", "mg_prof_reset", "mg_prof_avg_ms",
    "mg_screen_create", "mg_screen_create_translated", "mg_screen_add_host", "mg_screen_add_dev", "mg_screen_finish_host", "mg_screen_counts_dev", "mg_screen_free",
    "mg_screen_reset", "mg_screen_finish_sparse_host", "mg_screen_tier_note", "mg_dscreen_finish_sparse_host", "mg_dscreen_reset",
    "mg_identity", "mg_p_value_within",
    "mg_comm_create_local", "mg_comm_unique_id", "mg_comm_create_rank", "mg_comm_destroy", "mg_comm_size", "mg_comm_rank",
    "mg_comm_uses_rccl", "mg_comm_ctx", "mg_comm_last_error", "mg_shard_tri_rows", "mg_shard_tri_rows_weighted", "mg_shard_tri_rows_costed", "mg_shard_rows", "mg_dtable_upload",
    "mg_dtable_free", "mg_dtable_local", "mg_table_broadcast", "mg_comm_allreduce_u32_sum", "mg_dtable_upload_rows", "mg_sketch_sharded_host",
    "mg_compare_tri_sharded_host", "mg_compare_rect_sharded_host", "mg_compare_tri_pairs_sharded_host",
    "mg_compare_rect_pairs_sharded_host", "mg_compare_tri_results_sharded_host", "mg_compare_rect_results_sharded_host",
    "mg_dscreen_create", "mg_dscreen_add_host", "mg_dscreen_finish_host", "mg_dscreen_free",
]


class MgParams(C.Structure):
    _fields_ = [
        ("kmer_size", C.c_int32),
        ("seed", C.c_uint32),
        ("sketch_size", C.c_uint64),
        ("alphabet_size", C.c_uint32),
        ("alphabet", C.c_uint8 * 256),
        ("preserve_case", C.c_uint8),
        ("use64", C.c_uint8),
        ("noncanonical", C.c_uint8),
        ("counts", C.c_uint8),
        ("min_copies", C.c_uint32),
        ("target_cov", C.c_double),
        ("bloom_bytes", C.c_uint64),
    ]


class MgPair(C.Structure):
    _fields_ = [
        ("numer", C.c_uint32),
        ("denom", C.c_uint32),
        ("distance", C.c_double),
        ("p_value", C.c_double),
        ("pass_", C.c_uint8),
        ("_pad", C.c_uint8 * 7),
    ]


PAIR_DTYPE = np.dtype([("numer", "<u4"), ("denom", "<u4"), ("distance", "<f8"), ("p_value", "<f8"),
                       ("pass", "u1"), ("_pad", "u1", 7)])
COUNTS_DTYPE = np.dtype([("numer", "<u4"), ("denom", "<u4")])
EDGE_DTYPE = np.dtype([("row", "<u4"), ("col", "<u4"), ("numer", "<u4"), ("denom", "<u4")])
RESULT_DTYPE = np.dtype([("row", "<u4"), ("col", "<u4"), ("numer", "<u4"), ("denom", "<u4"),
                         ("distance", "<f8"), ("p_value", "<f8")])


class ScreenSession:
    """One mg_screen: add batches of the mixture, then read the per-hash observation counts.
    Use as a context manager or call close()."""

    def __init__(self, eng, db, p, translate=False):
        self.eng, self.db, self.p = eng, db, p
        self.h = C.c_void_p()
        create = eng.lib.mg_screen_create_translated if translate else eng.lib.mg_screen_create
        eng._check(create(eng.ctx, C.byref(p), db.handle, C.byref(self.h)))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self.h:
            self.eng.lib.mg_screen_free(self.h)
            self.h = C.c_void_p()

    def add_records(self, recs):
        blob = np.frombuffer(join_records(recs), dtype=np.uint8)
        self.add_host(blob)

    def add_host(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self.eng._check(self.eng.lib.mg_screen_add_host(self.h, blob.ctypes.data, len(blob)))

    def add_dev(self, ptr, nbytes):
        """records joined with MG_RECORD_SEP, device memory, 16-byte aligned"""
        self.eng._check(self.eng.lib.mg_screen_add_dev(self.h, ptr, nbytes))

    def counts_dev(self, out_ptr):
        """u32[db.rows * db.sketch_size] into device memory (operand of the all-reduce)"""
        self.eng._check(self.eng.lib.mg_screen_counts_dev(self.h, out_ptr))

    def finish(self, want_counts=True, want_distinct=True):
        """(counts u32[n, s] or None, mixture sketch u64[<=s], distinct table hashes or None)"""
        n, s = self.db.rows, self.db.sketch_size
        counts = np.zeros((n, s), dtype=np.uint32) if want_counts else None
        mix = np.zeros(int(self.p.sketch_size), dtype=np.uint64)
        mn, dist = C.c_uint32(0), C.c_uint64(0)
        self.eng._check(self.eng.lib.mg_screen_finish_host(
            self.h, counts.ctypes.data if want_counts else None, mix.ctypes.data, C.byref(mn),
            C.byref(dist) if want_distinct else None))
        return counts, mix[: mn.value].copy(), (int(dist.value) if want_distinct else None)

    def finish_sparse(self):
        """(hits HIT_DTYPE[nhits] ordered by row then hash, mixture sketch, distinct table hashes): the non-zero
        cells of finish()'s counts"""
        return _finish_sparse(self.eng, self.eng.lib.mg_screen_finish_sparse_host, self.h, int(self.p.sketch_size))

    def reset(self):
        """ready for the next mixture against the same (resident) database"""
        self.eng._check(self.eng.lib.mg_screen_reset(self.h))

    def tier_note(self):
        return self.eng.lib.mg_screen_tier_note(self.h).decode()


HIT_DTYPE = np.dtype([("row", np.uint32), ("count", np.uint32), ("hash", np.uint64)])


def _finish_sparse(owner, fn, h, s):
    n = C.c_uint64(0)
    owner._check(fn(h, None, 0, C.byref(n), None, None, None))
    hits = np.zeros(n.value, dtype=HIT_DTYPE)
    mix = np.zeros(s, dtype=np.uint64)
    mn, dist = C.c_uint32(0), C.c_uint64(0)
    owner._check(fn(h, hits.ctypes.data if n.value else None, n.value, C.byref(n), mix.ctypes.data, C.byref(mn), C.byref(dist)))
    return hits, mix[: mn.value].copy(), int(dist.value)


def hits_of_counts(counts):
    """the sparse form of a dense counts matrix (row-major order = by row; hashes not known here)"""
    r, c = np.nonzero(counts)
    return r.astype(np.uint32), counts[r, c]


class MashGpuError(RuntimeError):
    pass


def load_library():
    if not os.path.exists(LIB_PATH):
        raise MashGpuError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    vp, u64, u32, i32, dbl = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_double
    lib.mg_device_count.argtypes = []
    lib.mg_ctx_create.argtypes = [i32, C.POINTER(vp)]
    lib.mg_ctx_destroy.argtypes = [vp]
    lib.mg_ctx_destroy.restype = None
    lib.mg_last_error.argtypes = [vp]
    lib.mg_last_error.restype = C.c_char_p
    lib.mg_ctx_set_stream.argtypes = [vp, vp]
    lib.mg_ctx_synchronize.argtypes = [vp]
    lib.mg_ctx_set_async.argtypes = [vp, i32]
    lib.mg_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.mg_ctx_cu_count.argtypes = [vp]
    lib.mg_params_init.argtypes = [C.POINTER(MgParams), i32, u64, u32, C.c_char_p, i32, i32]
    lib.mg_sketch_host.argtypes = [vp, C.POINTER(MgParams), vp, u64, vp, u64, vp, vp, vp]
    lib.mg_sketch_dev.argtypes = [vp, C.POINTER(MgParams), vp, u64, vp, u64, vp, vp, vp]
    lib.mg_packed_bytes.argtypes = [u64]
    lib.mg_packed_bytes.restype = u64
    lib.mg_packed_mask_bytes.argtypes = [u64]
    lib.mg_packed_mask_bytes.restype = u64
    lib.mg_pack_bases.argtypes = [vp, u64, i32, vp, vp, C.POINTER(u64)]
    lib.mg_sketch_host_packed.argtypes = [vp, C.POINTER(MgParams), vp, vp, u64, vp, u64, vp, vp, vp]
    lib.mg_sketch_dev_packed.argtypes = [vp, C.POINTER(MgParams), vp, vp, u64, vp, u64, vp, vp, vp]
    lib.mg_sketch_reads_host.argtypes = [vp, C.POINTER(MgParams), vp, u64, vp, vp, vp, vp]
    lib.mg_sketch_begin.argtypes = [vp, C.POINTER(MgParams), C.POINTER(vp)]
    lib.mg_sketch_add.argtypes = [vp, vp, u64]
    lib.mg_sketch_end_sketch.argtypes = [vp]
    lib.mg_sketch_stage_capacity.argtypes = [vp]
    lib.mg_sketch_stage_capacity.restype = u64
    lib.mg_sketch_stage.argtypes = [vp, u64, C.POINTER(vp)]
    lib.mg_sketch_commit.argtypes = [vp, u64]
    lib.mg_sketch_pending.argtypes = [vp]
    lib.mg_sketch_pending.restype = u64
    lib.mg_sketch_finish.argtypes = [vp, vp, vp, vp]
    lib.mg_sketch_session_free.argtypes = [vp]
    lib.mg_sketch_session_free.restype = None
    lib.mg_reads_begin.argtypes = [vp, C.POINTER(MgParams), C.POINTER(vp)]
    lib.mg_reads_add_host.argtypes = [vp, vp, u64, C.POINTER(C.c_int)]
    lib.mg_reads_finish.argtypes = [vp, vp, vp, vp, vp]
    lib.mg_reads_reset.argtypes = [vp]
    lib.mg_reads_free.argtypes = [vp]
    lib.mg_reads_free.restype = None
    lib.mg_table_upload.argtypes = [vp, vp, vp, vp, u64, u64, C.POINTER(vp)]
    lib.mg_table_wrap_dev.argtypes = [vp, vp, vp, vp, u64, u64, C.POINTER(vp)]
    lib.mg_table_free.argtypes = [vp]
    lib.mg_table_free.restype = None
    lib.mg_table_invalidate.argtypes = [vp]
    lib.mg_ctx_trim.argtypes = [vp]
    lib.mg_table_rows.argtypes = [vp]
    lib.mg_table_rows.restype = u64
    lib.mg_table_sketch_size.argtypes = [vp]
    lib.mg_table_sketch_size.restype = u64
    lib.mg_compare_tri_dev.argtypes = [vp, vp, u64, u64, vp]
    lib.mg_compare_tri_host.argtypes = [vp, vp, u64, u64, vp]
    lib.mg_compare_rect_dev.argtypes = [vp, vp, vp, u64, u64, vp]
    lib.mg_compare_rect_host.argtypes = [vp, vp, vp, u64, u64, vp]
    lib.mg_compare_tri_filter_host.argtypes = [vp, vp, u64, u64, C.c_int, C.c_double, vp, u64, vp]
    lib.mg_compare_rect_filter_host.argtypes = [vp, vp, vp, u64, u64, C.c_int, C.c_double, vp, u64, vp]
    lib.mg_compare_tri_sparse_host.argtypes = [vp, vp, u64, u64, vp, u64, vp]
    lib.mg_compare_rect_sparse_host.argtypes = [vp, vp, vp, u64, u64, vp, u64, vp]
    lib.mg_expand_tri_sparse.argtypes = [vp, u64, vp, u64, u64, u64, vp]
    lib.mg_finish_tri_host.argtypes = [vp, vp, u64, u64, i32, dbl, dbl, dbl, vp]
    lib.mg_finish_rect_host.argtypes = [vp, vp, u64, vp, u64, i32, dbl, dbl, dbl, vp]
    lib.mg_finish_tri_dev.argtypes = [vp, vp, vp, u64, u64, i32, dbl, dbl, dbl, vp]
    lib.mg_finish_rect_dev.argtypes = [vp, vp, vp, vp, u64, u64, i32, dbl, dbl, dbl, vp]
    lib.mg_compare_tri_pairs_host.argtypes = [vp, vp, u64, u64, i32, dbl, dbl, dbl, vp]
    lib.mg_compare_rect_pairs_host.argtypes = [vp, vp, vp, u64, u64, i32, dbl, dbl, dbl, vp]
    lib.mg_compare_tri_results_host.argtypes = [vp, vp, u64, u64, i32, dbl, dbl, dbl, vp, u64, vp]
    lib.mg_compare_rect_results_host.argtypes = [vp, vp, vp, u64, u64, i32, dbl, dbl, dbl, vp, u64, vp]
    lib.mg_comm_create_local.argtypes = [C.POINTER(C.c_int), i32, C.POINTER(vp)]
    lib.mg_comm_unique_id.argtypes = [vp, C.c_size_t]
    lib.mg_comm_create_rank.argtypes = [vp, vp, C.c_size_t, i32, i32, C.POINTER(vp)]
    lib.mg_comm_destroy.argtypes = [vp]
    lib.mg_comm_destroy.restype = None
    lib.mg_comm_size.argtypes = [vp]
    lib.mg_comm_rank.argtypes = [vp]
    lib.mg_comm_uses_rccl.argtypes = [vp]
    lib.mg_comm_ctx.argtypes = [vp, i32]
    lib.mg_comm_ctx.restype = vp
    lib.mg_comm_last_error.argtypes = [vp]
    lib.mg_comm_last_error.restype = C.c_char_p
    lib.mg_shard_tri_rows.argtypes = [u64, u64, i32, i32, C.POINTER(u64), C.POINTER(u64)]
    lib.mg_shard_tri_rows.restype = None
    lib.mg_shard_tri_rows_weighted.argtypes = [u64, u64, i32, i32, C.c_double, C.POINTER(u64), C.POINTER(u64)]
    lib.mg_shard_tri_rows_weighted.restype = None
    lib.mg_shard_tri_rows_costed.argtypes = [u64, u64, i32, i32, C.c_double, C.c_double, C.POINTER(u64), C.POINTER(u64)]
    lib.mg_shard_tri_rows_costed.restype = None
    lib.mg_shard_rows.argtypes = [u64, u64, i32, i32, C.POINTER(u64), C.POINTER(u64)]
    lib.mg_shard_rows.restype = None
    lib.mg_dtable_upload.argtypes = [vp, vp, vp, vp, u64, u64, C.POINTER(vp)]
    lib.mg_dtable_upload_rows.argtypes = [vp, vp, vp, vp, u64, u64, C.POINTER(vp)]
    lib.mg_sketch_sharded_host.argtypes = [vp, C.POINTER(MgParams), vp, u64, vp, u64, vp, vp, vp]
    lib.mg_dtable_free.argtypes = [vp]
    lib.mg_dtable_free.restype = None
    lib.mg_dtable_local.argtypes = [vp, i32]
    lib.mg_dtable_local.restype = vp
    lib.mg_table_broadcast.argtypes = [vp, vp, i32, u64, u64, C.POINTER(vp)]
    lib.mg_comm_allreduce_u32_sum.argtypes = [vp, vp, u64]
    lib.mg_compare_tri_sharded_host.argtypes = [vp, vp, u64, u64, vp]
    lib.mg_compare_rect_sharded_host.argtypes = [vp, vp, vp, u64, u64, vp]
    lib.mg_compare_tri_pairs_sharded_host.argtypes = [vp, vp, u64, u64, i32, dbl, dbl, dbl, vp]
    lib.mg_compare_rect_pairs_sharded_host.argtypes = [vp, vp, vp, u64, u64, i32, dbl, dbl, dbl, vp]
    lib.mg_compare_tri_results_sharded_host.argtypes = [vp, vp, u64, u64, i32, dbl, dbl, dbl, vp, u64, vp]
    lib.mg_compare_rect_results_sharded_host.argtypes = [vp, vp, vp, u64, u64, i32, dbl, dbl, dbl, vp, u64, vp]
    lib.mg_dscreen_create.argtypes = [vp, C.POINTER(MgParams), vp, i32, C.POINTER(vp)]
    lib.mg_dscreen_add_host.argtypes = [vp, vp, u64]
    lib.mg_dscreen_finish_host.argtypes = [vp, vp, vp, C.POINTER(u32), C.POINTER(u64)]
    lib.mg_dscreen_free.argtypes = [vp]
    lib.mg_dscreen_free.restype = None
    lib.mg_distance.argtypes = [u32, u32, i32]
    lib.mg_distance.restype = dbl
    lib.mg_p_value.argtypes = [u64, u64, u64, dbl, u64]
    lib.mg_p_value.restype = dbl
    lib.mg_prof_enable.argtypes = [vp, i32]
    lib.mg_prof_reset.argtypes = [vp]
    lib.mg_prof_reset.restype = None
    lib.mg_prof_avg_ms.argtypes = [vp, C.c_char_p, C.POINTER(u64)]
    lib.mg_prof_avg_ms.restype = dbl
    lib.mg_screen_create.argtypes = [vp, C.POINTER(MgParams), vp, C.POINTER(vp)]
    lib.mg_screen_create_translated.argtypes = [vp, C.POINTER(MgParams), vp, C.POINTER(vp)]
    lib.mg_screen_add_host.argtypes = [vp, vp, u64]
    lib.mg_screen_add_dev.argtypes = [vp, vp, u64]
    lib.mg_screen_finish_host.argtypes = [vp, vp, vp, C.POINTER(u32), C.POINTER(u64)]
    lib.mg_screen_counts_dev.argtypes = [vp, vp]
    lib.mg_screen_free.argtypes = [vp]
    lib.mg_screen_free.restype = None
    lib.mg_screen_reset.argtypes = [vp]
    lib.mg_screen_finish_sparse_host.argtypes = [vp, vp, u64, C.POINTER(u64), vp, C.POINTER(u32), C.POINTER(u64)]
    lib.mg_screen_tier_note.argtypes = [vp]
    lib.mg_screen_tier_note.restype = C.c_char_p
    lib.mg_dscreen_finish_sparse_host.argtypes = [vp, vp, u64, C.POINTER(u64), vp, C.POINTER(u32), C.POINTER(u64)]
    lib.mg_dscreen_reset.argtypes = [vp]
    lib.mg_identity.argtypes = [u64, u64, i32]
    lib.mg_identity.restype = dbl
    lib.mg_p_value_within.argtypes = [u64, u64, dbl, u64]
    lib.mg_p_value_within.restype = dbl
    return lib


def tri_pairs(row_begin, row_end):
    t = lambda x: x * (x - 1) // 2 if x else 0
    return t(row_end) - t(row_begin)


def make_params(lib, k=21, s=1000, seed=42, alphabet="ACGT", noncanonical=False, preserve_case=False,
                min_copies=1, target_cov=0.0, bloom_bytes=0):
    p = MgParams()
    rc = lib.mg_params_init(C.byref(p), k, s, seed, alphabet.encode(), int(noncanonical), int(preserve_case))
    if rc != MG_OK:
        raise MashGpuError(f"mg_params_init failed ({rc})")
    p.min_copies = min_copies
    p.target_cov = target_cov
    p.bloom_bytes = bloom_bytes
    return p


def join_records(records):
    """Records of ONE sketch -> byte string with MG_RECORD_SEP between records."""
    return bytes([RECORD_SEP]).join(records)


class Table:
    def __init__(self, eng, handle, keep=()):
        self.eng, self.handle, self._keep = eng, handle, keep

    @property
    def rows(self):
        return int(self.eng.lib.mg_table_rows(self.handle))

    @property
    def sketch_size(self):
        return int(self.eng.lib.mg_table_sketch_size(self.handle))

    def free(self):
        if self.handle:
            self.eng.lib.mg_table_free(self.handle)
            self.handle = None

    def invalidate(self):
        """The buffers' contents changed: drop everything derived from them (mg_table_invalidate)."""
        self.eng._check(self.eng.lib.mg_table_invalidate(self.handle))

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def shard_tri_rows_weighted(lib, row_begin, row_end, nranks, rank, row_weight):
    """row block of `rank` when a row costs `row_weight` pairs on top of its pairs (mg_shard_tri_rows_weighted)"""
    b, e = C.c_uint64(0), C.c_uint64(0)
    lib.mg_shard_tri_rows_weighted(row_begin, row_end, nranks, rank, float(row_weight), C.byref(b), C.byref(e))
    return int(b.value), int(e.value)


def shard_tri_rows_costed(lib, row_begin, row_end, nranks, rank, row_weight, prefix_weight):
    """row block of `rank` when the block [lo, hi) costs pairs + row_weight (hi - lo) + prefix_weight hi (mg_shard_tri_rows_costed)"""
    b, e = C.c_uint64(0), C.c_uint64(0)
    lib.mg_shard_tri_rows_costed(row_begin, row_end, nranks, rank, float(row_weight), float(prefix_weight), C.byref(b), C.byref(e))
    return b.value, e.value


def shard_tri_rows(lib, row_begin, row_end, nranks, rank):
    """equal-area row block of `rank` (mg_shard_tri_rows)"""
    b, e = C.c_uint64(0), C.c_uint64(0)
    lib.mg_shard_tri_rows(row_begin, row_end, nranks, rank, C.byref(b), C.byref(e))
    return int(b.value), int(e.value)


class LocalComm:
    """mg_comm in local mode: one process, a context per listed device, tables replicated on all of
    them (RCCL broadcast from device 0), compares sharded by row blocks."""

    def __init__(self, devices):
        self.lib = load_library()
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self.lib.mg_comm_create_local(arr, len(devices), C.byref(h))
        if rc != MG_OK:
            raise MashGpuError("mg_comm_create_local: " + self.lib.mg_comm_last_error(None).decode())
        self.h = h
        self.size = len(devices)

    def _check(self, rc):
        if rc != MG_OK:
            raise MashGpuError(f"libmashgpu error {rc}: " + self.lib.mg_comm_last_error(self.h).decode())

    @property
    def uses_rccl(self):
        return bool(self.lib.mg_comm_uses_rccl(self.h))

    def close(self):
        if self.h:
            self.lib.mg_comm_destroy(self.h)
            self.h = None

    def upload(self, hashes, nhash, lengths):
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        nhash = np.ascontiguousarray(nhash, dtype=np.uint32)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint64)
        n, s = hashes.shape
        d = C.c_void_p()
        self._check(self.lib.mg_dtable_upload(self.h, hashes.ctypes.data, nhash.ctypes.data, lengths.ctypes.data, n, s, C.byref(d)))
        return d

    def free(self, d):
        self.lib.mg_dtable_free(d)

    def screen(self, d, n, s, p, batches, translate=False, sparse=False, again=None):
        """sharded screen of `batches` (lists of records) against replicated table d; sparse: hits instead of
        the dense matrix; again: a second list of batches screened after a reset of the same (resident)
        database -- its result is returned instead"""
        h = C.c_void_p()
        self._check(self.lib.mg_dscreen_create(self.h, C.byref(p), d, int(translate), C.byref(h)))
        try:
            for rnd in ([batches] if again is None else [batches, again]):
                if rnd is again:
                    self._check(self.lib.mg_dscreen_reset(h))
                for recs in rnd:                      # a list of records, or the bytes already joined (uint8 array)
                    blob = np.ascontiguousarray(recs) if isinstance(recs, np.ndarray) else np.frombuffer(join_records(recs), dtype=np.uint8)
                    self._check(self.lib.mg_dscreen_add_host(h, blob.ctypes.data, len(blob)))
            if sparse:
                return _finish_sparse(self, self.lib.mg_dscreen_finish_sparse_host, h, int(p.sketch_size))
            counts = np.zeros((n, s), dtype=np.uint32)
            mix = np.zeros(int(p.sketch_size), dtype=np.uint64)
            mn, dist = C.c_uint32(0), C.c_uint64(0)
            self._check(self.lib.mg_dscreen_finish_host(h, counts.ctypes.data, mix.ctypes.data, C.byref(mn), C.byref(dist)))
            return counts, mix[: mn.value].copy(), int(dist.value)
        finally:
            self.lib.mg_dscreen_free(h)

    def tri(self, d, n, row_begin=0, row_end=None):
        row_end = n if row_end is None else row_end
        out = np.zeros(tri_pairs(row_begin, row_end), dtype=COUNTS_DTYPE)
        self._check(self.lib.mg_compare_tri_sharded_host(self.h, d, row_begin, row_end, out.ctypes.data))
        return out

    def upload_rows(self, hashes, nhash, lengths):
        """row-sharded table (one block of consecutive rows per device): the reference side of rect jobs"""
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        nhash = np.ascontiguousarray(nhash, dtype=np.uint32)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint64)
        n, s = hashes.shape
        d = C.c_void_p()
        self._check(self.lib.mg_dtable_upload_rows(self.h, hashes.ctypes.data, nhash.ctypes.data, lengths.ctypes.data, n, s, C.byref(d)))
        return d

    def sketch(self, sketches, p, counts=False):
        """mg_sketch_sharded_host: lists of records per sketch, on every device of the communicator"""
        blobs = [join_records(recs) for recs in sketches]
        bases = np.frombuffer(b"".join(blobs), dtype=np.uint8)
        off = np.zeros(len(blobs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(b) for b in blobs], dtype=np.uint64)
        s = int(p.sketch_size)
        hashes = np.zeros((len(blobs), s), dtype=np.uint64)
        nhash = np.zeros(len(blobs), dtype=np.uint32)
        cnt = np.zeros((len(blobs), s), dtype=np.uint32) if counts else None
        self._check(self.lib.mg_sketch_sharded_host(self.h, C.byref(p), bases.ctypes.data if len(bases) else None, len(bases), off.ctypes.data,
                                                    len(blobs), hashes.ctypes.data, nhash.ctypes.data, cnt.ctypes.data if counts else None))
        return (hashes, nhash, cnt) if counts else (hashes, nhash)

    def rect(self, dref, dqry, nref, nq, q_begin=0, q_end=None):
        q_end = nq if q_end is None else q_end
        out = np.zeros((q_end - q_begin, nref), dtype=COUNTS_DTYPE)
        self._check(self.lib.mg_compare_rect_sharded_host(self.h, dref, dqry, q_begin, q_end, out.ctypes.data))
        return out

    def tri_pairs(self, d, n, k, kmer_space, max_d=-1.0, max_p=-1.0):
        out = np.zeros(tri_pairs(0, n), dtype=PAIR_DTYPE)
        self._check(self.lib.mg_compare_tri_pairs_sharded_host(self.h, d, 0, n, k, kmer_space, max_d, max_p, out.ctypes.data))
        return out

    def rect_pairs(self, dref, dqry, nref, nq, k, kmer_space, max_d=-1.0, max_p=-1.0):
        out = np.zeros((nq, nref), dtype=PAIR_DTYPE)
        self._check(self.lib.mg_compare_rect_pairs_sharded_host(self.h, dref, dqry, 0, nq, k, kmer_space, max_d, max_p, out.ctypes.data))
        return out

    def _results(self, call, capacity):
        n = C.c_uint64(0)
        for _ in range(2):
            out = np.zeros(max(int(capacity), 1), dtype=RESULT_DTYPE)
            rc = call(out.ctypes.data, int(capacity), C.byref(n))
            if rc == MG_OK:
                return out[:n.value]
            if rc != MG_ERR_NOMEM or n.value <= capacity:
                self._check(rc)
            capacity = n.value
        self._check(rc)

    def tri_results(self, d, n, k, kmer_space, max_d=-1.0, max_p=-1.0, capacity=1 << 16):
        return self._results(lambda o, c, cnt: self.lib.mg_compare_tri_results_sharded_host(
            self.h, d, 0, n, k, kmer_space, max_d, max_p, o, c, cnt), capacity)

    def rect_results(self, dref, dqry, nq, k, kmer_space, max_d=-1.0, max_p=-1.0, capacity=1 << 16):
        return self._results(lambda o, c, cnt: self.lib.mg_compare_rect_results_sharded_host(
            self.h, dref, dqry, 0, nq, k, kmer_space, max_d, max_p, o, c, cnt), capacity)


class RankComm:
    """mg_comm in rank mode: one process per GPU.  `exchange(bytes_or_None) -> bytes` hands rank 0's
    128-byte id to every rank (torch.distributed, a file, anything)."""

    def __init__(self, eng, nranks, rank, exchange):
        self.eng, self.lib = eng, eng.lib
        idbuf = C.create_string_buffer(128)
        err = None
        if rank == 0 and self.lib.mg_comm_unique_id(idbuf, 128) != MG_OK:
            err = "mg_comm_unique_id: " + self.lib.mg_comm_last_error(None).decode()
        # (a failure on rank 0 still goes through the exchange, as an empty id: the other ranks are
        #  waiting in it and must fail with rank 0, not hang)
        raw = exchange((bytes(idbuf.raw) if err is None else b"") if rank == 0 else None)
        if not raw or len(raw) != 128:
            raise MashGpuError(err or "rank 0 could not create a communicator id")
        idbuf = C.create_string_buffer(raw, 128)
        h = C.c_void_p()
        rc = self.lib.mg_comm_create_rank(eng.ctx, idbuf, 128, nranks, rank, C.byref(h))
        if rc != MG_OK:
            raise MashGpuError("mg_comm_create_rank: " + self.lib.mg_comm_last_error(None).decode())
        self.h, self.nranks, self.rank = h, nranks, rank

    def _check(self, rc):
        if rc != MG_OK:
            raise MashGpuError(f"libmashgpu error {rc}: " + self.lib.mg_comm_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.mg_comm_destroy(self.h)
            self.h = None

    def table_broadcast(self, src_table, n, s, root=0):
        """table of the root on every rank (ncclBroadcast of hashes, nhash, lengths)"""
        h = C.c_void_p()
        self._check(self.lib.mg_table_broadcast(self.h, src_table.handle if src_table is not None else None, root, n, s, C.byref(h)))
        return Table(self.eng, h, keep=(src_table,))

    def allreduce_u32_sum(self, ptr, count):
        self._check(self.lib.mg_comm_allreduce_u32_sum(self.h, ptr, count))


def pack_bases(bases, preserve_case=False, lib=None, threads=1):
    """mg_pack_bases (host only: no context, no GPU): bases u8[n] -> (packed u8[(n + 3) / 4], invalid_mask u8[(n + 7) / 8],
    number of invalid bases).  threads > 1: disjoint ranges (starts multiples of 8) packed side by side."""
    lib = lib or load_library()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    n = len(bases)
    packed = np.zeros(int(lib.mg_packed_bytes(n)) + 16, dtype=np.uint8)[: int(lib.mg_packed_bytes(n))]
    mask = np.zeros(int(lib.mg_packed_mask_bytes(n)) + 16, dtype=np.uint8)[: int(lib.mg_packed_mask_bytes(n))]
    if threads <= 1 or n < (1 << 20):
        ninv = C.c_uint64(0)
        rc = lib.mg_pack_bases(bases.ctypes.data, n, 1 if preserve_case else 0, packed.ctypes.data, mask.ctypes.data, C.byref(ninv))
        if rc != MG_OK:
            raise MashGpuError(f"mg_pack_bases: {rc}")
        return packed, mask, int(ninv.value)
    from concurrent.futures import ThreadPoolExecutor
    step = ((n + threads - 1) // threads + 7) & ~7
    def one(b0):
        b1 = min(n, b0 + step)
        c = C.c_uint64(0)
        rc = lib.mg_pack_bases(bases.ctypes.data + b0, b1 - b0, 1 if preserve_case else 0, packed.ctypes.data + b0 // 4, mask.ctypes.data + b0 // 8, C.byref(c))
        if rc != MG_OK:
            raise MashGpuError(f"mg_pack_bases: {rc}")
        return int(c.value)
    with ThreadPoolExecutor(threads) as ex:
        ninv = sum(ex.map(one, range(0, n, step)))
    return packed, mask, ninv


class MashGpu:
    """One context = one process + one GPU (mg_ctx)."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.mg_ctx_create(device, C.byref(h))
        if rc != MG_OK:
            raise MashGpuError("mg_ctx_create: " + self.lib.mg_last_error(None).decode())
        self.ctx = h
        if stream is not None:
            self._check(self.lib.mg_ctx_set_stream(self.ctx, C.c_void_p(stream)))

    def set_option(self, name, value):
        """mg_ctx_set_option: a knob of this context (value None: back to the environment's setting)"""
        self._check(self.lib.mg_ctx_set_option(self.ctx, name.encode(), None if value is None else str(value).encode()))

    def close(self):
        if self.ctx:
            self.lib.mg_ctx_destroy(self.ctx)
            self.ctx = None

    def _check(self, rc):
        if rc != MG_OK:
            raise MashGpuError(f"libmashgpu error {rc}: " + self.lib.mg_last_error(self.ctx).decode())

    def params(self, **kw):
        return make_params(self.lib, **kw)

    def synchronize(self):
        self._check(self.lib.mg_ctx_synchronize(self.ctx))

    def trim(self):
        """Return the context's cached device blocks to the driver (mg_ctx_trim)."""
        self._check(self.lib.mg_ctx_trim(self.ctx))

    def set_async(self, on=True):
        self._check(self.lib.mg_ctx_set_async(self.ctx, int(on)))

    # ---- sketching ---------------------------------------------------------
    def sketch_host(self, sketches, p, counts=False):
        """sketches: list (one per sketch) of lists of record bytes.
        Returns (hashes u64[n, s], nhash u32[n]) and, with counts=True, multiplicities u32[n, s]."""
        blobs = [join_records(r) for r in sketches]
        bases = np.frombuffer(b"".join(blobs), dtype=np.uint8)
        off = np.zeros(len(blobs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(b) for b in blobs], dtype=np.uint64)
        return self.sketch_host_raw(bases, off, p, counts)

    def sketch_host_raw(self, bases, off, p, counts=False):
        n = len(off) - 1
        s = int(p.sketch_size)
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        hashes = np.zeros((n, s), dtype=np.uint64)
        nhash = np.zeros(n, dtype=np.uint32)
        cnt = np.zeros((n, s), dtype=np.uint32) if counts else None
        self._check(self.lib.mg_sketch_host(self.ctx, C.byref(p), bases.ctypes.data, len(bases),
                                            off.ctypes.data, n, hashes.ctypes.data, nhash.ctypes.data,
                                            cnt.ctypes.data if counts else None))
        return (hashes, nhash, cnt) if counts else (hashes, nhash)

    def sketch_host_packed_raw(self, packed, mask, nbases, off, p, counts=False):
        """mg_sketch_host_packed: packed u8[(nbases + 3) / 4], mask u8[(nbases + 7) / 8] or None, off u64[n + 1] in bases"""
        n = len(off) - 1
        s = int(p.sketch_size)
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        hashes = np.zeros((n, s), dtype=np.uint64)
        nhash = np.zeros(n, dtype=np.uint32)
        cnt = np.zeros((n, s), dtype=np.uint32) if counts else None
        self._check(self.lib.mg_sketch_host_packed(self.ctx, C.byref(p), packed.ctypes.data, None if mask is None else mask.ctypes.data,
                                                   nbases, off.ctypes.data, n, hashes.ctypes.data, nhash.ctypes.data,
                                                   cnt.ctypes.data if counts else None))
        return (hashes, nhash, cnt) if counts else (hashes, nhash)

    def sketch_dev_packed(self, packed_ptr, mask_ptr, nbases, off, p, hashes_ptr, nhash_ptr):
        off = np.ascontiguousarray(off, dtype=np.uint64)
        self._check(self.lib.mg_sketch_dev_packed(self.ctx, C.byref(p), packed_ptr, mask_ptr, nbases, off.ctypes.data,
                                                  len(off) - 1, hashes_ptr, nhash_ptr, None))

    def sketch_stream(self, sketches, p, counts=False, piece=None, windows=False):
        """same result as sketch_host, through a session: bytes handed over piece by piece -- copied by
        mg_sketch_add, or (windows) written by the caller into windows lent by mg_sketch_stage: as many
        whole sketches per window as fit, committed one by one; larger sketches go through mg_sketch_add"""
        h = C.c_void_p()
        self._check(self.lib.mg_sketch_begin(self.ctx, C.byref(p), C.byref(h)))
        try:
            blobs = [np.frombuffer(join_records(recs), dtype=np.uint8) for recs in sketches]
            cap = int(self.lib.mg_sketch_stage_capacity(h))
            i = 0
            while i < len(blobs):
                if windows and len(blobs[i]) <= cap:
                    j, total = i, 0
                    while j < len(blobs) and total + len(blobs[j]) <= cap:
                        total += len(blobs[j])
                        j += 1
                    win = C.c_void_p()
                    self._check(self.lib.mg_sketch_stage(h, min(cap, total + (piece or 0) % 3), C.byref(win)))     # (a window may be lent larger than used)
                    at = 0
                    for b in blobs[i:j]:
                        if len(b):
                            C.memmove(win.value + at, b.ctypes.data, len(b))
                        at += len(b)
                    for b in blobs[i:j]:
                        self._check(self.lib.mg_sketch_commit(h, len(b)))
                        self._check(self.lib.mg_sketch_end_sketch(h))
                    i = j
                    continue
                blob = blobs[i]
                step = piece or max(1, len(blob))
                for o in range(0, len(blob), step):
                    part = np.ascontiguousarray(blob[o:o + step])
                    self._check(self.lib.mg_sketch_add(h, part.ctypes.data, len(part)))
                self._check(self.lib.mg_sketch_end_sketch(h))
                i += 1
            n, s = int(self.lib.mg_sketch_pending(h)), int(p.sketch_size)
            hashes = np.zeros((n, s), dtype=np.uint64)
            nhash = np.zeros(n, dtype=np.uint32)
            cnt = np.zeros((n, s), dtype=np.uint32) if counts else None
            self._check(self.lib.mg_sketch_finish(h, hashes.ctypes.data, nhash.ctypes.data, cnt.ctypes.data if counts else None))
            assert int(self.lib.mg_sketch_pending(h)) == 0
            return (hashes, nhash, cnt) if counts else (hashes, nhash)
        finally:
            self.lib.mg_sketch_session_free(h)

    def sketch_reads(self, records, p):
        """reads mode, one sketch over `records` in order, honouring p.target_cov (-c) and
        p.bloom_bytes (-b):
        (hashes u64[n], counts u32[n], records_used)"""
        blob = np.frombuffer(join_records(records), dtype=np.uint8)
        s = int(p.sketch_size)
        hashes = np.full(s, HASH_PAD, dtype=np.uint64)
        counts = np.zeros(s, dtype=np.uint32)
        n, used = C.c_uint32(0), C.c_uint64(0)
        self._check(self.lib.mg_sketch_reads_host(self.ctx, C.byref(p), blob.ctypes.data, len(blob), hashes.ctypes.data,
                                                  C.byref(n), counts.ctypes.data, C.byref(used)))
        return hashes[: n.value].copy(), counts[: n.value].copy(), int(used.value)

    def sketch_reads_chunked(self, records, p, per_chunk, first=None):
        """sketch_reads through a session, `per_chunk` records at a time; stops feeding at the stop:
        (hashes, counts, records_used, chunks_fed).  first: another read set pushed through the session
        before (and discarded by mg_reads_reset) -- the result must not depend on it"""
        h = C.c_void_p()
        self._check(self.lib.mg_reads_begin(self.ctx, C.byref(p), C.byref(h)))
        try:
            if first is not None:
                blob = np.frombuffer(join_records(first) + bytes([RECORD_SEP]), dtype=np.uint8)
                self._check(self.lib.mg_reads_add_host(h, blob.ctypes.data, len(blob), None))
                self._check(self.lib.mg_reads_reset(h))
            fed = 0
            stopped = C.c_int(0)
            for o in range(0, len(records), per_chunk):
                blob = np.frombuffer(join_records(records[o:o + per_chunk]) + bytes([RECORD_SEP]), dtype=np.uint8)
                self._check(self.lib.mg_reads_add_host(h, blob.ctypes.data, len(blob), C.byref(stopped)))
                fed += 1
                if stopped.value:
                    break
            s = int(p.sketch_size)
            hashes = np.full(s, HASH_PAD, dtype=np.uint64)
            counts = np.zeros(s, dtype=np.uint32)
            n, used = C.c_uint32(0), C.c_uint64(0)
            self._check(self.lib.mg_reads_finish(h, hashes.ctypes.data, C.byref(n), counts.ctypes.data, C.byref(used)))
            return hashes[: n.value].copy(), counts[: n.value].copy(), int(used.value), fed
        finally:
            self.lib.mg_reads_free(h)

    def sketch_dev(self, bases_ptr, nbases, off, p, hashes_ptr, nhash_ptr):
        off = np.ascontiguousarray(off, dtype=np.uint64)
        self._check(self.lib.mg_sketch_dev(self.ctx, C.byref(p), bases_ptr, nbases, off.ctypes.data,
                                           len(off) - 1, hashes_ptr, nhash_ptr, None))

    # ---- tables --------------------------------------------------------------
    def table_upload(self, hashes, nhash, lengths=None):
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        nhash = np.ascontiguousarray(nhash, dtype=np.uint32)
        n, s = hashes.shape
        lp = None
        if lengths is not None:
            lengths = np.ascontiguousarray(lengths, dtype=np.uint64)
            lp = lengths.ctypes.data
        h = C.c_void_p()
        self._check(self.lib.mg_table_upload(self.ctx, hashes.ctypes.data, nhash.ctypes.data, lp, n, s, C.byref(h)))
        return Table(self, h)

    def table_wrap(self, hashes_ptr, nhash_ptr, lengths_ptr, n, s, keep=()):
        h = C.c_void_p()
        self._check(self.lib.mg_table_wrap_dev(self.ctx, hashes_ptr, nhash_ptr, lengths_ptr, n, s, C.byref(h)))
        return Table(self, h, keep)

    # ---- comparing -------------------------------------------------------------
    def compare_tri_host(self, table, row_begin=0, row_end=None, out=None):
        row_end = table.rows if row_end is None else min(row_end, table.rows)
        if out is None:
            out = np.zeros(tri_pairs(row_begin, row_end), dtype=COUNTS_DTYPE)
        assert out.dtype == COUNTS_DTYPE and len(out) >= tri_pairs(row_begin, row_end) and out.flags.c_contiguous
        self._check(self.lib.mg_compare_tri_host(self.ctx, table.handle, row_begin, row_end, out.ctypes.data))
        return out

    def compare_tri_dev(self, table, row_begin, row_end, out_ptr):
        self._check(self.lib.mg_compare_tri_dev(self.ctx, table.handle, row_begin, row_end, out_ptr))

    def compare_rect_host(self, ref, qry, q_begin=0, q_end=None):
        q_end = qry.rows if q_end is None else min(q_end, qry.rows)
        out = np.zeros((q_end - q_begin, ref.rows), dtype=COUNTS_DTYPE)
        self._check(self.lib.mg_compare_rect_host(self.ctx, ref.handle, qry.handle, q_begin, q_end, out.ctypes.data))
        return out

    def compare_rect_dev(self, ref, qry, q_begin, q_end, out_ptr):
        self._check(self.lib.mg_compare_rect_dev(self.ctx, ref.handle, qry.handle, q_begin, q_end, out_ptr))

    def _filter(self, call, capacity, dtype=EDGE_DTYPE):
        """run a *_filter_host / *_results_host call, growing the buffer once if the first guess was too small"""
        n = C.c_uint64(0)
        for _ in range(2):
            out = np.zeros(max(int(capacity), 1), dtype=dtype)
            rc = call(out.ctypes.data, int(capacity), C.byref(n))
            if rc == MG_OK:
                return out[:n.value]
            if rc != MG_ERR_NOMEM or n.value <= capacity:               # MG_ERR_NOMEM with a count: retry
                self._check(rc)
            capacity = n.value
        self._check(rc)

    def compare_tri_filter(self, table, k, max_d, row_begin=0, row_end=None, capacity=1 << 20):
        """pairs of rows [row_begin,row_end) x earlier rows with distance <= max_d, reference order"""
        row_end = table.rows if row_end is None else min(row_end, table.rows)
        return self._filter(lambda o, c, n: self.lib.mg_compare_tri_filter_host(
            self.ctx, table.handle, row_begin, row_end, k, max_d, o, c, n), capacity)

    def compare_rect_filter(self, ref, qry, k, max_d, q_begin=0, q_end=None, capacity=1 << 20):
        q_end = qry.rows if q_end is None else min(q_end, qry.rows)
        return self._filter(lambda o, c, n: self.lib.mg_compare_rect_filter_host(
            self.ctx, ref.handle, qry.handle, q_begin, q_end, k, max_d, o, c, n), capacity)

    def compare_tri_sparse(self, table, row_begin=0, row_end=None, capacity=1 << 20):
        """the triangle's EXCEPTIONS: every pair with numer >= 1 as {row, col, numer, denom}, reference order; every other
        pair is {0, min(s, |A| + |B|)} (mg_compare_tri_sparse_host)"""
        row_end = table.rows if row_end is None else min(row_end, table.rows)
        return self._filter(lambda o, c, n: self.lib.mg_compare_tri_sparse_host(self.ctx, table.handle, row_begin, row_end, o, c, n), capacity)

    def compare_rect_sparse(self, ref, qry, q_begin=0, q_end=None, capacity=1 << 20):
        q_end = qry.rows if q_end is None else min(q_end, qry.rows)
        return self._filter(lambda o, c, n: self.lib.mg_compare_rect_sparse_host(self.ctx, ref.handle, qry.handle, q_begin, q_end, o, c, n), capacity)

    def expand_tri_sparse(self, edges, nhash, s, row_begin, row_end):
        """the dense triangle out of the rule and the exceptions (host arithmetic)"""
        out = np.zeros(tri_pairs(row_begin, row_end), dtype=COUNTS_DTYPE)
        edges = np.ascontiguousarray(edges)
        nhash = np.ascontiguousarray(nhash, dtype=np.uint32)
        rc = self.lib.mg_expand_tri_sparse(edges.ctypes.data, len(edges), nhash.ctypes.data, int(s), int(row_begin), int(row_end), out.ctypes.data)
        if rc != MG_OK:
            raise MashGpuError(f"mg_expand_tri_sparse: {rc}")
        return out

    # ---- compare + finish on the device ---------------------------------------------
    def compare_tri_pairs(self, table, k, kmer_space, max_d=-1.0, max_p=-1.0, row_begin=0, row_end=None):
        """every pair of rows [row_begin,row_end) x earlier rows as PairOutput records (device finish)"""
        row_end = table.rows if row_end is None else min(row_end, table.rows)
        out = np.zeros(tri_pairs(row_begin, row_end), dtype=PAIR_DTYPE)
        self._check(self.lib.mg_compare_tri_pairs_host(self.ctx, table.handle, row_begin, row_end, k, kmer_space,
                                                       max_d, max_p, out.ctypes.data))
        return out

    def compare_rect_pairs(self, ref, qry, k, kmer_space, max_d=-1.0, max_p=-1.0, q_begin=0, q_end=None):
        q_end = qry.rows if q_end is None else min(q_end, qry.rows)
        out = np.zeros((q_end - q_begin, ref.rows), dtype=PAIR_DTYPE)
        self._check(self.lib.mg_compare_rect_pairs_host(self.ctx, ref.handle, qry.handle, q_begin, q_end, k, kmer_space,
                                                        max_d, max_p, out.ctypes.data))
        return out

    def compare_tri_results(self, table, k, kmer_space, max_d=-1.0, max_p=-1.0, row_begin=0, row_end=None,
                            capacity=1 << 20):
        """survivors of both filters with distance and p-value, reference order (all on the device)"""
        row_end = table.rows if row_end is None else min(row_end, table.rows)
        return self._filter(lambda o, c, n: self.lib.mg_compare_tri_results_host(
            self.ctx, table.handle, row_begin, row_end, k, kmer_space, max_d, max_p, o, c, n), capacity, RESULT_DTYPE)

    def compare_rect_results(self, ref, qry, k, kmer_space, max_d=-1.0, max_p=-1.0, q_begin=0, q_end=None,
                             capacity=1 << 20):
        q_end = qry.rows if q_end is None else min(q_end, qry.rows)
        return self._filter(lambda o, c, n: self.lib.mg_compare_rect_results_host(
            self.ctx, ref.handle, qry.handle, q_begin, q_end, k, kmer_space, max_d, max_p, o, c, n), capacity, RESULT_DTYPE)

    def finish_tri_dev(self, table, counts_ptr, row_begin, row_end, k, kmer_space, max_d, max_p, out_ptr):
        self._check(self.lib.mg_finish_tri_dev(self.ctx, table.handle, counts_ptr, row_begin, row_end, k, kmer_space,
                                               max_d, max_p, out_ptr))

    # ---- finishing (host arithmetic) --------------------------------------------
    def finish_tri(self, counts, lengths, row_begin, row_end, k, kmer_space, max_d=-1.0, max_p=-1.0):
        counts = np.ascontiguousarray(counts)
        lengths = np.ascontiguousarray(lengths, dtype=np.uint64)
        out = np.zeros(len(counts), dtype=PAIR_DTYPE)
        rc = self.lib.mg_finish_tri_host(counts.ctypes.data, lengths.ctypes.data, row_begin, row_end, k,
                                         kmer_space, max_d, max_p, out.ctypes.data)
        self._check(rc)
        return out

    def finish_rect(self, counts, len_ref, len_qry, k, kmer_space, max_d=-1.0, max_p=-1.0):
        counts = np.ascontiguousarray(counts)
        len_ref = np.ascontiguousarray(len_ref, dtype=np.uint64)
        len_qry = np.ascontiguousarray(len_qry, dtype=np.uint64)
        out = np.zeros(counts.shape, dtype=PAIR_DTYPE)
        rc = self.lib.mg_finish_rect_host(counts.ctypes.data, len_ref.ctypes.data, len(len_ref),
                                          len_qry.ctypes.data, len(len_qry), k, kmer_space, max_d, max_p,
                                          out.ctypes.data)
        self._check(rc)
        return out

    # ---- screening -------------------------------------------------------------------
    def screen_open(self, db, p, translate=False):
        """incremental screen against table `db` (see ScreenSession); translate=True: amino-acid
        sketches against a nucleotide mixture, translated in six frames on the device"""
        return ScreenSession(self, db, p, translate)

    def screen(self, db, p, batches):
        """Containment counts of every hash of table `db` in a mixture given as batches of
        record lists.  Returns (counts u32[n, s], mixture sketch u64[<=s], distinct hashes)."""
        h = C.c_void_p()
        self._check(self.lib.mg_screen_create(self.ctx, C.byref(p), db.handle, C.byref(h)))
        try:
            for recs in batches:                      # a list of records, or the bytes already joined (uint8 array)
                blob = np.ascontiguousarray(recs) if isinstance(recs, np.ndarray) else np.frombuffer(join_records(recs), dtype=np.uint8)
                self._check(self.lib.mg_screen_add_host(h, blob.ctypes.data, len(blob)))
            n, s = db.rows, db.sketch_size
            counts = np.zeros((n, s), dtype=np.uint32)
            mix = np.zeros(int(p.sketch_size), dtype=np.uint64)
            mn, dist = C.c_uint32(0), C.c_uint64(0)
            self._check(self.lib.mg_screen_finish_host(h, counts.ctypes.data, mix.ctypes.data, C.byref(mn), C.byref(dist)))
            return counts, mix[: mn.value].copy(), int(dist.value)
        finally:
            self.lib.mg_screen_free(h)

    # ---- profiling hook -----------------------------------------------------------
    def prof_enable(self, on=True):
        self._check(self.lib.mg_prof_enable(self.ctx, int(on)))

    def prof_reset(self):
        self.lib.mg_prof_reset(self.ctx)

    def prof_avg_ms(self, name):
        n = C.c_uint64(0)
        ms = self.lib.mg_prof_avg_ms(self.ctx, name.encode(), C.byref(n))
        return float(ms), int(n.value)
