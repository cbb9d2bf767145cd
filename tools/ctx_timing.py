#!/usr/bin/env python3
"""Where the start-up time of a process that sketches goes: library load, context creation, the first
launch of each kernel family (code object load), steady state.  One JSON line."""
import ctypes as C, json, os, sys, time
t0 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mash_amd import abi
t1 = time.perf_counter()
lib = abi.load_library()
t2 = time.perf_counter()
eng = abi.MashGpu(0)
t3 = time.perf_counter()
rng = np.random.default_rng(0)
g = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 50000)].tobytes()
res = {"import_numpy_abi_s": t1 - t0, "dlopen_s": t2 - t1, "ctx_create_s": t3 - t2}
for k in (21, 21, 31, 21):
    p = eng.params(k=k, s=1000)
    t = time.perf_counter()
    eng.sketch_host([[g]], p)
    res.setdefault("sketch_calls_s", []).append([k, time.perf_counter() - t])
print(json.dumps(res))
