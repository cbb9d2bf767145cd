#!/usr/bin/env python3
"""ONE workload of bench.py on its own, for rocprofv3 passes (tools/profile_round3.sh): the kernels of
a leg then carry no launches of another leg under the same name, so `profiles/r03_kernel_stats_<leg>.csv`
reproduces that leg's pass time by itself.

  python tools/prof_leg.py --leg c3|c5|random|identical|clades|sketch|screen [--steps 2] [--n ...]

Every leg warms up once (index / prefix images are built there) and then runs --steps timed steps;
the number of launches per kernel = (1 + steps) x launches per pass unless a kernel only runs cold."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mash_amd import abi
from workloads import synth_torch

ap = argparse.ArgumentParser()
ap.add_argument("--leg", required=True)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--n", type=int, default=100000)
ap.add_argument("--n-genomes", type=int, default=10000)
ap.add_argument("--n-reads", type=int, default=10_000_000)
ap.add_argument("--cold", action="store_true", help="compare legs: every step from an invalidated table (the per-table job)")
args = ap.parse_args()
torch.cuda.init()
dev = torch.device("cuda", 0)
eng = abi.MashGpu(0, stream=torch.cuda.current_stream().cuda_stream)
S, K = 1000, 21
res = {"leg": args.leg, "steps": args.steps}


def triangle(h, nh, ln, n, s):
    torch.cuda.synchronize()
    t = eng.table_wrap(h.data_ptr(), nh.data_ptr(), ln.data_ptr(), n, s, keep=(h, nh, ln))
    pairs = n * (n - 1) // 2
    out = torch.empty((pairs, 2), dtype=torch.int32, device=dev)
    eng.compare_tri_dev(t, 0, n, out.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if args.cold:
            t.invalidate()
        eng.compare_tri_dev(t, 0, n, out.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    res.update({"pairs": pairs, "cold": args.cold, "ms_per_step": dt * 1e3, "pairs_per_s": pairs / dt,
                "checksum": [int(out[:, 0].sum(dtype=torch.int64).item()), int(out[:, 1].sum(dtype=torch.int64).item())]})
    t.free()


n = args.n
if args.leg == "c3":
    triangle(*synth_torch.clustered_sketch_table(n, S, clusters=max(1, n // 100), device=dev), n, S)
elif args.leg == "c5":
    triangle(*synth_torch.clustered_sketch_table(n, 10000, clusters=max(1, n // 100), pool=15000, private=4000, device=dev, block=2000), n, 10000)
elif args.leg == "random":
    triangle(*synth_torch.random_sketch_table(n, S, device=dev), n, S)
elif args.leg == "identical":
    triangle(*synth_torch.identical_sketch_table(n, S, device=dev), n, S)
elif args.leg == "clades":
    triangle(*synth_torch.clade_sketch_table(n, S, device=dev), n, S)
elif args.leg == "one_clade":
    n = min(n, 32768)
    triangle(*synth_torch.clade_sketch_table(n, S, clade=n, device=dev), n, S)
elif args.leg == "one_species":
    n = min(n, 32768)
    triangle(*synth_torch.species_sketch_table(n, S, device=dev), n, S)
elif args.leg == "sketch":
    ng, L = args.n_genomes, 1_000_000
    bases = synth_torch.synthetic_genomes(0, ng, L, device=dev)
    off = np.arange(ng + 1, dtype=np.uint64) * np.uint64(L)
    sk = torch.empty((ng, S), dtype=torch.int64, device=dev)
    nh = torch.empty(ng, dtype=torch.int32, device=dev)
    p = eng.params(k=K, s=S)
    torch.cuda.synchronize()
    eng.sketch_dev(bases.data_ptr(), ng * L, off, p, sk.data_ptr(), nh.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.sketch_dev(bases.data_ptr(), ng * L, off, p, sk.data_ptr(), nh.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    res.update({"bases": ng * L, "ms_per_step": dt * 1e3, "bp_per_s": ng * L / dt})
elif args.leg == "screen":
    from mash_amd import screen_dist
    RL, NSRC, GL = 150, 1000, 1_000_000
    p = eng.params(k=K, s=S)
    genomes = synth_torch.synthetic_genomes(0, NSRC, GL, device=dev, stride=40000)
    gh = torch.empty((NSRC, S), dtype=torch.int64, device=dev)
    gn = torch.empty(NSRC, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng.sketch_dev(genomes.data_ptr(), NSRC * GL, np.arange(NSRC + 1, dtype=np.uint64) * np.uint64(GL), p, gh.data_ptr(), gn.data_ptr())
    fh, fn, _ = synth_torch.clustered_sketch_table(n - NSRC, S, clusters=max(1, (n - NSRC) // 100), device=dev)
    db_h = torch.cat([gh, fh], 0).contiguous()
    db_n = torch.cat([gn, fn], 0).contiguous()
    db_l = torch.full((n,), GL, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    db = eng.table_wrap(db_h.data_ptr(), db_n.data_ptr(), db_l.data_ptr(), n, S, keep=(db_h, db_n, db_l))
    nb = 4
    per = (args.n_reads + nb - 1) // nb
    batches = [synth_torch.synthetic_reads(genomes, min(per, args.n_reads - b * per), RL, seed=7000 + b) for b in range(nb)]
    handles = [(b.data_ptr(), int(b.numel()), b) for b in batches]
    t_create = time.perf_counter()
    sc = eng.screen_open(db, p)
    torch.cuda.synchronize()
    res["create_ms"] = (time.perf_counter() - t_create) * 1e3
    res["key_bound"] = sc.tier_note()
    phase = {"reset": 0.0, "add": 0.0, "finish_sparse": 0.0}

    def step(timed):
        t = time.perf_counter()
        sc.reset()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for h in handles:
            sc.add_dev(h[0], h[1])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hits, mix, _ = sc.finish_sparse()
        t3 = time.perf_counter()
        if timed:
            phase["reset"] += t1 - t; phase["add"] += t2 - t1; phase["finish_sparse"] += t3 - t2
        return hits

    step(False)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hits = step(True)
    dt = (time.perf_counter() - t0) / args.steps
    res.update({"reads": args.n_reads, "ms_per_step": dt * 1e3, "reads_per_s": args.n_reads / dt, "hits": int(len(hits)),
                "phase_ms": {k: v * 1e3 / args.steps for k, v in phase.items()}})
    sc.close()
else:
    raise SystemExit("unknown leg")
print(json.dumps(res))
