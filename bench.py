#!/usr/bin/env python3
"""bench.py — headline benchmark of the Mash hot path on MI355X.

Metric (BASELINE.json): pairwise Mash distances/sec at s=1000, k=21 (64-bit hashes).
Workload: BASELINE config 3 — `mash triangle`, all-vs-all on N=100 000 pre-built
clustered synthetic sketches: 4.99995e9 pairs per step.  The table
(800 MB) is resident in HBM on every rank before the timed region; a step = one full
pass over the lower triangle, row-block sharded (equal-area blocks) across the ranks,
each rank writing {numer, denom} for its rows into its own HBM buffer (8 B/pair).
Total work is fixed as N grows ("scaling": "strong").  The only exchange is one RCCL
broadcast of the table from rank 0 before the timed region (reported separately).

Secondary (same JSON line, key "sketch"): BASELINE config 2 — sketch 10 000 synthetic
1 Mbp genomes, k=21 s=1000, reported as sketched bp/sec.

Also reported: `roofline` for the dominant kernel (compare_tiled_kernel), measured live
with HIP events on the launch stream inside libmashgpu (mg_prof_*), and `cpu_baseline`
= the reference's own compareSketches (oracle/_ref, kind "reference") or the C port
(kind "port") timed on this box's host cores on a bounded sample of the same table
(rank 0, N=1 only).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--n-sketches 100000] [--n-genomes 10000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
S = 1000
K = 21


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-sketches", type=int, default=100_000)
    ap.add_argument("--n-genomes", type=int, default=10_000)
    ap.add_argument("--genome-len", type=int, default=1_000_000)
    ap.add_argument("--no-sketch", action="store_true", help="skip the secondary sketch measurement")
    ap.add_argument("--no-screen", action="store_true", help="skip the tertiary screen measurement (config 4)")
    ap.add_argument("--n-reads", type=int, default=10_000_000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--dry-cpu", action="store_true",
                    help="plumbing test only (CI without GPUs): gloo + CPU tensors, no kernels, output marked dry")
    return ap.parse_args()


def cpu_baseline_compare(table_np, nhash_np, lengths_np, budget_s):
    """Reference compareSketches (incl. p-value) on the host cores, bounded sample:
    triangle rows of the first M sketches of the SAME table, M sized for ~budget_s."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle
    cores = min(os.cpu_count() or 1, 16)
    use_ref = pyoracle.ref_available()
    orc = pyoracle.Oracle(ref=use_ref)
    kspace = 4.0 ** K

    def run(m):
        from mash_amd import shard
        b = shard.equal_area_row_blocks(m, cores)
        sub = (table_np[:m], nhash_np[:m], lengths_np[:m])
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:      # ctypes releases the GIL: real parallelism
            list(ex.map(lambda g: orc.triangle(sub[0], sub[1], sub[2], b[g], b[g + 1], K, kspace, stats=True),
                        range(cores)))
        return time.perf_counter() - t0

    m = 400
    dt = run(m)
    rate = (m * (m - 1) / 2) / dt
    m2 = int(min(len(table_np), max(m, (2 * rate * budget_s) ** 0.5)))
    dt2 = run(m2)
    pairs = m2 * (m2 - 1) // 2
    return {"value": pairs / dt2, "unit": "pairs/s", "cores": cores,
            "kind": "reference" if use_ref else "port",
            "sample": f"triangle (compareSketches incl. p-value) on the first {m2} of the same sketches, "
                      f"{pairs} pairs, {cores} threads, {dt2:.1f} s"}


def cpu_baseline_sketch(budget_s):
    from oracle import pyoracle
    from mash_amd import synth
    use_ref = pyoracle.ref_available()
    orc = pyoracle.Oracle(ref=use_ref)
    p = orc.params(k=K, s=S)
    g = bytes(synth.synthetic_genome(0, 1_000_000))
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s:
        orc.sketch_records([g], p)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n * 1e6 / dt, "unit": "bp/s", "cores": 1, "kind": "reference" if use_ref else "port",
            "sample": f"{n} x 1 Mbp synthetic genome, addMinHashes+MinHashHeap, 1 thread, {dt:.1f} s"}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from mash_amd import abi, shard, synth_torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dry = args.dry_cpu
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    class _DryEngine:                      # no kernels: only the control flow around them is exercised
        def table_wrap(self, *a, **k):
            return self
        def compare_tri_dev(self, table, rb, re, out_ptr):
            pass
        def prof_enable(self, on=True):
            pass
        def prof_reset(self):
            pass
        def prof_avg_ms(self, name):
            return 0.0, 0
        def free(self):
            pass
        def close(self):
            pass

    eng = _DryEngine() if dry else abi.MashGpu(local_rank, stream=torch.cuda.current_stream().cuda_stream)

    # ------------------------------------------------------------------ table
    n = args.n_sketches
    if rank == 0:
        hashes, nhash, lengths = synth_torch.clustered_sketch_table(n, S, clusters=max(1, n // 100), device=dev)
    else:
        hashes = torch.empty((n, S), dtype=torch.int64, device=dev)
        nhash = torch.empty(n, dtype=torch.int32, device=dev)
        lengths = torch.empty(n, dtype=torch.int64, device=dev)
    bcast_ms = 0.0
    if world > 1:
        barrier()
        t0 = time.perf_counter()
        dist.broadcast(hashes, 0)          # the one exchange step: sketch table over xGMI
        dist.broadcast(nhash, 0)
        dist.broadcast(lengths, 0)
        barrier()
        bcast_ms = (time.perf_counter() - t0) * 1e3
    table = eng.table_wrap(hashes.data_ptr(), nhash.data_ptr(), lengths.data_ptr(), n, S,
                           keep=(hashes, nhash, lengths))

    blocks = shard.equal_area_row_blocks(n, world)
    rb, re = blocks[rank], blocks[rank + 1]
    my_pairs = shard.tri_pairs(rb, re)
    total_pairs = n * (n - 1) // 2
    out = torch.empty((max(my_pairs, 1), 2), dtype=torch.int32, device=dev)

    def step():
        eng.compare_tri_dev(table, rb, re, out.data_ptr())

    if not dry:
        torch.cuda.synchronize()       # table generation (torch stream) -> library stream
    for _ in range(args.warmup):
        step()
    eng.prof_enable(True)
    eng.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms, launches = eng.prof_avg_ms("compare")
    eng.prof_enable(False)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    value = total_pairs * args.steps / dt

    # cheap sanity on the produced output (outside the timed region): denom == s, numer <= s
    chk = out[: min(my_pairs, 1_000_000)]
    if not dry:
        assert int(chk[:, 1].min()) == S and int(chk[:, 0].max()) <= S, "compare output failed sanity check"

    # roofline of the dominant kernel on this rank: algorithmic bytes = pairs * (2*s*8 + 8)
    bytes_per_pair = 2 * S * 8 + 8
    achieved = (my_pairs * bytes_per_pair / (kern_ms * 1e-3)) / 1e9 if kern_ms > 0 else 0.0
    traffic = None
    units = None
    pmc_path = os.path.join(ROOT, "profiles", "compare_pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
            traffic = pmc.get("hbm_bytes_per_launch")
            units = pmc.get("units")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "compare_merged_kernel", "kernel_ms": round(kern_ms, 3), "launches": launches,
                "algorithmic_bytes_per_pair": bytes_per_pair,
                "measured_hbm_frac": (round(traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                      if traffic and kern_ms > 0 and n == 100_000 and world == 1 else None),
                "units": units,           # PMC pass over this command: VALU / SALU issue, LDS activity
                "note": "achieved/frac use the mandated no-reuse model of SURVEY.md §8d (2*s*8+8 B per pair); every "
                        "sketch is re-used ~1000x from LDS/L2, so frac exceeds 1. traffic = PMC-measured HBM bytes per "
                        "launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/compare_pmc_latest.json); measured_hbm_frac = "
                        "traffic / kernel time / peak"}

    result = {
        "metric": "pairwise Mash distances/sec (s=1000)",
        "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"mash triangle all-vs-all, {n} clustered synthetic sketches, k={K} s={S}, "
                               f"{total_pairs} pairs/step, row-block sharded x{world}",
                   "n_sketches": n, "sketch_size": S, "kmer": K, "hash_bits": 64,
                   "parallelism": f"rowblock{world}", "table_broadcast_ms": round(bcast_ms, 2)},
        "roofline": roofline,
    }
    if dry:
        result["dry"] = True               # plumbing test: NOT a measurement
        result["rank_blocks"] = blocks

    # ------------------------------------------------------------------ cpu baseline (rank 0, N=1)
    if rank == 0 and world == 1 and not args.no_cpu:
        m = min(n, 6000)
        result["cpu_baseline"] = cpu_baseline_compare(
            hashes[:m].cpu().numpy().view(np.uint64), nhash[:m].cpu().numpy().astype(np.uint32),
            lengths[:m].cpu().numpy().astype(np.uint64), args.cpu_seconds)

    # ------------------------------------------------------------------ secondary: sketching (config 2)
    del out
    if not args.no_sketch and not dry:
        g_blocks = shard.even_blocks(args.n_genomes, world)
        g0, g1 = g_blocks[rank], g_blocks[rank + 1]
        ng, L = g1 - g0, args.genome_len
        bases = synth_torch.synthetic_genomes(g0, g1, L, device=dev)
        off = np.arange(ng + 1, dtype=np.uint64) * np.uint64(L)
        sk_hashes = torch.empty((max(ng, 1), S), dtype=torch.int64, device=dev)
        sk_nhash = torch.empty(max(ng, 1), dtype=torch.int32, device=dev)
        p = eng.params(k=K, s=S)

        def sk_step():
            eng.sketch_dev(bases.data_ptr(), ng * L, off, p, sk_hashes.data_ptr(), sk_nhash.data_ptr())

        torch.cuda.synchronize()
        sk_step()
        eng.prof_enable(True)
        eng.prof_reset()
        barrier()
        t0 = time.perf_counter()
        sk_steps = max(2, args.steps)
        for _ in range(sk_steps):
            sk_step()
        barrier()
        sdt = time.perf_counter() - t0
        sk_ms, sk_launches = eng.prof_avg_ms("sketch")
        eng.prof_enable(False)
        tmax = torch.tensor([sdt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        sdt = float(tmax.item())
        assert int(sk_nhash.min()) == S, "sketch output failed sanity check"
        sk_bytes = ng * (L + 8 * S)                   # 1 B/base in + 8*s B per sketch out
        sk_traffic = None
        sk_units = None
        sk_pmc = os.path.join(ROOT, "profiles", "sketch_pmc_latest.json")
        if os.path.exists(sk_pmc) and world == 1 and args.n_genomes == 10_000 and L == 1_000_000:
            try:
                pmc = json.load(open(sk_pmc))
                sk_traffic = pmc.get("hbm_bytes_per_launch")
                sk_units = pmc.get("units")
            except Exception:
                sk_traffic = None
        sk_ach = sk_bytes / (sk_ms * 1e-3) / 1e9 if sk_ms > 0 else 0.0
        sketch = {"metric": "sketched bp/sec (k=21, s=1000)", "value": args.n_genomes * L * sk_steps / sdt,
                  "unit": "bp/s", "ms_per_step": sdt * 1e3 / sk_steps, "steps": sk_steps,
                  "config": {"workload": f"sketch {args.n_genomes} synthetic {L} bp genomes, k={K} s={S}, "
                                         f"ASCII bases resident in HBM, sharded x{world}"},
                  "roofline": {"bound": "hbm", "achieved": round(sk_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(sk_ach / HBM_PEAK_GBS, 4), "traffic": sk_traffic,
                               "kernel": "sketch_chunks_kernel<21,0,256,false>", "kernel_ms": round(sk_ms, 3),
                               "launches": sk_launches, "units": sk_units,
                               "note": "integer-ALU bound: PMC (units) shows the VALU issuing practically every "
                                       "cycle at ~150 VALU instructions per k-mer (10 64-bit multiplies on 32-bit "
                                       "halves), HBM traffic = algorithmic bytes; see DESIGN.md 4.2"}}
        if rank == 0 and world == 1 and not args.no_cpu:
            sketch["cpu_baseline"] = cpu_baseline_sketch(min(args.cpu_seconds, 6.0))
        result["sketch"] = sketch

    # ------------------------------------------------------------------ tertiary: screen (config 4)
    # 10^7 x 150 bp reads (0.5 % errors, both strands) against the first 10^5 - 10^3 rows of the
    # C3 table + the real sketches of the 10^3 genomes the reads come from.  Reads are sharded
    # by batch; the one collective is the all-reduce of the observation counters (RCCL) plus
    # an all-gather of the per-rank mixture sketches (mash_amd/screen_dist.py).  A step is the
    # whole job: table build, every batch, counters gathered + exchanged.
    if not args.no_screen and not dry:
        import gc
        bases = sk_hashes = sk_nhash = None             # release the sketch workload
        gc.collect()
        torch.cuda.empty_cache()
        from mash_amd import screen_dist
        scr = {"metric": "screened reads/sec (150 bp, k=21, s=1000, 100k-sketch database)", "unit": "reads/s"}
        ok = torch.ones(1, dtype=torch.int32, device=dev)
        try:
            RL, NSRC, GL = 150, 1000, 1_000_000
            p = eng.params(k=K, s=S)
            genomes = synth_torch.synthetic_genomes(0, NSRC, GL, device=dev, stride=40000)
            gh = torch.empty((NSRC, S), dtype=torch.int64, device=dev)
            gn = torch.empty(NSRC, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()                 # torch stream -> library stream
            eng.sketch_dev(genomes.data_ptr(), NSRC * GL, np.arange(NSRC + 1, dtype=np.uint64) * np.uint64(GL), p,
                           gh.data_ptr(), gn.data_ptr())
            rest = max(0, min(n, 100_000) - NSRC)
            db_h = torch.cat([gh, hashes[:rest]], 0).contiguous()
            db_n = torch.cat([gn, nhash[:rest]], 0).contiguous()
            db_l = torch.full((NSRC + rest,), GL, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            db = eng.table_wrap(db_h.data_ptr(), db_n.data_ptr(), db_l.data_ptr(), NSRC + rest, S, keep=(db_h, db_n, db_l))
            nb = max(4, world)                        # >= one batch per rank; few, large batches
            per_batch = (args.n_reads + nb - 1) // nb
            mine = screen_dist.shard_batches(nb, rank, world)
            batches = [synth_torch.synthetic_reads(genomes, min(per_batch, args.n_reads - b * per_batch), RL, seed=7000 + b)
                       for b in mine]
            del genomes
            torch.cuda.synchronize()
            local = screen_dist.gpu_local_screen(eng, db, p)
            handles = [(b.data_ptr(), int(b.numel()), b) for b in batches]

            def scr_step():
                # a local failure must not leave the other ranks alone in the collective
                try:
                    counts, mix = local(handles)
                except Exception as e:
                    scr["error"] = repr(e)
                    ok.zero_()
                    counts = torch.zeros((NSRC + rest) * S, dtype=torch.int32, device=dev)
                    mix = np.zeros(0, dtype=np.uint64)
                return screen_dist.exchange(counts, mix, S)
        except Exception as e:                      # keep every rank in step for the collectives below
            ok.zero_()
            scr["error"] = repr(e)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            counts, mix = scr_step()
            barrier()
            t0 = time.perf_counter()
            scr_steps = max(2, args.steps)
            for _ in range(scr_steps):
                counts, mix = scr_step()
            barrier()
            qdt = time.perf_counter() - t0
            tmax = torch.tensor([qdt], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            qdt = float(tmax.item())
        if int(ok.item()) == 1:
            shared = (counts.view(NSRC + rest, S)[:NSRC] > 0).sum(1).float().mean().item()
            assert 500 < shared < 900 and len(mix) == S, f"screen output failed sanity check (shared {shared}, mix {len(mix)})"
            scr.update({"value": args.n_reads * scr_steps / qdt, "ms_per_step": qdt * 1e3 / scr_steps, "steps": scr_steps,
                        "bp_per_s": args.n_reads * RL * scr_steps / qdt,
                        "config": {"workload": f"mash screen: {args.n_reads} synthetic {RL} bp reads (0.5% errors) vs "
                                               f"{NSRC + rest} sketches ({(NSRC + rest) * S} keys), reads resident in HBM, "
                                               f"batch-sharded x{world}, counters all-reduced",
                                   "mean_shared_hashes_of_sampled_genomes": round(shared, 1)},
                        "roofline": {"bound": "hbm", "achieved": round(args.n_reads * (RL + 1) * scr_steps / qdt / 1e9, 1),
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(args.n_reads * (RL + 1) * scr_steps / qdt / 1e9 / HBM_PEAK_GBS, 4),
                                     "traffic": None, "kernel": "sketch_chunks_kernel<21,0,256> (fused table probe)",
                                     "note": "1 B/base streamed once; integer-ALU bound like sketching (one murmur per "
                                             "k-mer), table probes filtered by the largest key; whole-step time, "
                                             "includes table build, counter gather and the exchange"}})
            db.free()
        elif "error" not in scr:
            scr["error"] = "failed on another rank"
        result["screen"] = scr

    if rank == 0:
        print(json.dumps(result))
    table.free()
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
