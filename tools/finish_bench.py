#!/usr/bin/env python3
"""Host-side finishing rate (distance + p-value, mg_finish_tri_host) on this box's cores."""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mash_amd import abi
lib = C.CDLL(os.path.join(os.path.dirname(abi.__file__), "libmashgpu.so"))
vp, u64 = C.c_void_p, C.c_uint64
lib.mg_finish_tri_host.argtypes = [vp, vp, u64, u64, C.c_int, C.c_double, C.c_double, C.c_double, vp]
rng = np.random.default_rng(1)
res = {"cores": os.cpu_count()}
for tag, lo, hi in (("related_x300", 250, 450), ("mixed_x0_1000", 0, 1001)):
    n = 8000
    npairs = n * (n - 1) // 2
    counts = np.zeros(npairs, dtype=abi.COUNTS_DTYPE)
    counts["numer"] = rng.integers(lo, hi, npairs)
    counts["denom"] = 1000
    lengths = rng.integers(45000, 55000, n).astype(np.uint64)
    out = np.zeros(npairs, dtype=abi.PAIR_DTYPE)
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        lib.mg_finish_tri_host(counts.ctypes.data, lengths.ctypes.data, 0, n, 21, 4.0 ** 21, -1.0, -1.0, out.ctypes.data)
        ts.append(time.perf_counter() - t)
    res[tag] = {"pairs": npairs, "seconds": round(min(ts), 4), "Mpairs_per_s": round(npairs / min(ts) / 1e6, 1),
                "first_call_seconds": round(ts[0], 4)}
print(json.dumps(res))
