#!/usr/bin/env python3
"""A/B of the packed sketch path: the sketch kernel expanding the 2-bit codes itself while it stages its tiles
(round 5) against unpack_bases_kernel + the ASCII kernel (round 4), and the ASCII path, all on device-resident input; then host
memory to host memory (mg_sketch_host_packed) both ways.  Same sketches required."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mash_amd import abi
from workloads import synth_torch
torch.cuda.init()
dev = torch.device("cuda", 0)
eng = abi.MashGpu(0, stream=torch.cuda.current_stream().cuda_stream)
ng, L, S, K = int(os.environ.get("NG", 2000)), 1_000_000, 1000, 21
p = eng.params(k=K, s=S)
bases = synth_torch.synthetic_genomes(0, ng, L, device=dev)
off = np.arange(ng + 1, dtype=np.uint64) * np.uint64(L)
hb = bases.cpu().numpy().reshape(-1)
packed, mask, ninv = abi.pack_bases(hb, threads=32)
dpk = torch.from_numpy(np.concatenate([packed, np.zeros(64, np.uint8)])).to(dev)
dmk = torch.from_numpy(np.concatenate([mask, np.zeros(64, np.uint8)])).to(dev)
sk = [torch.empty((ng, S), dtype=torch.int64, device=dev) for _ in range(3)]
nh = [torch.empty(ng, dtype=torch.int32, device=dev) for _ in range(3)]
res = {"bases": ng * L, "invalid": int(ninv)}


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        d = time.perf_counter() - t0
        best = d if best is None else min(best, d)
    return best


print("stage ascii", flush=True)
d = timed(lambda: eng.sketch_dev(bases.data_ptr(), ng * L, off, p, sk[0].data_ptr(), nh[0].data_ptr()))
res["ascii_dev_bp_s"] = ng * L / d
os.environ.pop("MASHGPU_PACKED_UNPACK", None)
print("stage in-kernel", flush=True)
d = timed(lambda: eng.sketch_dev_packed(dpk.data_ptr(), dmk.data_ptr(), ng * L, off, p, sk[1].data_ptr(), nh[1].data_ptr()))
res["packed_in_kernel_dev_bp_s"] = ng * L / d
os.environ["MASHGPU_PACKED_UNPACK"] = "1"
print("stage unpack", flush=True)
d = timed(lambda: eng.sketch_dev_packed(dpk.data_ptr(), dmk.data_ptr(), ng * L, off, p, sk[2].data_ptr(), nh[2].data_ptr()))
res["packed_unpack_kernel_dev_bp_s"] = ng * L / d
res["same_sketches"] = bool(torch.equal(sk[0], sk[1]) and torch.equal(sk[0], sk[2]) and torch.equal(nh[0], nh[1]) and torch.equal(nh[0], nh[2]))
for tag, env in (("in_kernel", None), ("unpack_kernel", "1")):
    if env: os.environ["MASHGPU_PACKED_UNPACK"] = env
    else: os.environ.pop("MASHGPU_PACKED_UNPACK", None)
    print("stage h2h", tag, flush=True)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        h, n_ = eng.sketch_host_packed_raw(packed, mask, ng * L, off, p)[:2]
        dd = time.perf_counter() - t0
        best = dd if best is None else min(best, dd)
    res[f"h2h_packed_{tag}_bp_s"] = ng * L / best
    res[f"h2h_packed_{tag}_same"] = bool(np.array_equal(h.view(np.int64), sk[0].cpu().numpy()))
print(json.dumps(res))
