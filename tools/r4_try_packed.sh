#!/bin/bash
# tools/r4_try_packed.sh -- first GPU contact of the packed ingest: its tests, then host -> host rates (ASCII vs packed, 2000 x 1 Mbp)
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "packed" 2>&1 | tail -5 )
timeout 300 python - <<'PY'
import sys, time, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from mash_amd import abi
from workloads import synth_torch
eng = abi.MashGpu(0)
ng, L = 2000, 1_000_000
bases = synth_torch.synthetic_genomes(0, ng, L, device=torch.device("cuda", 0))
hb = bases[:ng].cpu().numpy().reshape(-1)
del bases
off = np.arange(ng + 1, dtype=np.uint64) * np.uint64(L)
p = eng.params(k=21, s=1000)
res = {}
for thr in (1, 16, 64):
    t0 = time.perf_counter(); packed, mask, ninv = abi.pack_bases(hb, threads=thr); res[f"pack_{thr}_threads_bp_s"] = len(hb) / (time.perf_counter() - t0)
def best(fn, reps=3):
    b = None
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); d = time.perf_counter() - t0
        b = d if b is None else min(b, d)
    return b, out
ta, (h0, n0) = best(lambda: eng.sketch_host_raw(hb, off, p))
tp, (h1, n1) = best(lambda: eng.sketch_host_packed_raw(packed, mask, len(hb), off, p))
tn, (h2, n2) = best(lambda: eng.sketch_host_packed_raw(packed, None, len(hb), off, p))
assert np.array_equal(h0, h1) and np.array_equal(n0, n1) and np.array_equal(h0, h2)
res.update({"ninvalid": ninv, "ascii_bp_s": len(hb) / ta, "packed_bp_s": len(hb) / tp, "packed_nomask_bp_s": len(hb) / tn,
            "ascii_ms": ta * 1e3, "packed_ms": tp * 1e3, "packed_nomask_ms": tn * 1e3})
for piece in (1 << 26, 1 << 27, 1 << 29):
    os.environ["MASHGPU_PACKED_PIECE"] = str(piece)
    t, _ = best(lambda: eng.sketch_host_packed_raw(packed, mask, len(hb), off, p), 2)
    res[f"packed_piece_{piece}_bp_s"] = len(hb) / t
print(json.dumps(res, indent=1))
json.dump(res, open("gpurun_out/packed_h2h.json", "w"), indent=1)
PY
