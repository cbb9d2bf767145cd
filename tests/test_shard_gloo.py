"""N>1 path on CPU: world_size-2 gloo processes run the same sharding + table
broadcast + per-rank slice logic bench.py uses on RCCL; the per-rank compute is the
oracle here (no GPU), so this checks decomposition, ordering and reassembly."""
import os
import socket
import sys

import numpy as np
import pytest

from mash_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_equal_area_blocks_partition_and_balance():
    for n in (2, 3, 17, 1000, 100000):
        for w in (1, 2, 3, 4, 8):
            b = shard.equal_area_row_blocks(n, w)
            assert b[0] == 0 and b[-1] == n and len(b) == w + 1
            assert all(b[i] <= b[i + 1] for i in range(w))
            pairs = [shard.tri_pairs(b[i], b[i + 1]) for i in range(w)]
            assert sum(pairs) == n * (n - 1) // 2
            if n >= 1000:
                assert max(pairs) - min(pairs) <= 2 * n        # within ~two rows of each other
    assert shard.even_blocks(10, 4) == [0, 3, 6, 8, 10]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from workloads import synth
    from oracle import pyoracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, s = 120, 200
    if rank == 0:
        table, nhash, lengths = synth.clustered_sketches(n, s, clusters=4, seed=2, pool=300, private=80)
        tt = torch.from_numpy(table.view(np.int64))
        tn = torch.from_numpy(nhash.astype(np.int32))
    else:
        tt = torch.empty((n, s), dtype=torch.int64)
        tn = torch.empty(n, dtype=torch.int32)
    dist.broadcast(tt, 0)          # the one exchange step (RCCL broadcast over xGMI on the GPU box)
    dist.broadcast(tn, 0)
    table = tt.numpy().view(np.uint64)
    nhash = tn.numpy().astype(np.uint32)
    b = shard.equal_area_row_blocks(n, world)
    orc = pyoracle.Oracle()
    numer, denom, _, _ = orc.triangle(table, nhash, np.ones(n, np.uint64), b[rank], b[rank + 1], 21, 4.0 ** 21)
    dist.barrier()
    q.put((rank, b[rank], b[rank + 1], numer, denom))
    dist.destroy_process_group()


def test_two_rank_triangle_reassembles(oracle):
    import torch.multiprocessing as mp
    from workloads import synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, sk = 120, 200
    table, nhash, lengths = synth.clustered_sketches(n, sk, clusters=4, seed=2, pool=300, private=80)
    numer, denom, _, _ = oracle.triangle(table, nhash, np.ones(n, np.uint64), 0, n, 21, 4.0 ** 21)
    got_n = np.concatenate([r[3] for r in res])
    got_d = np.concatenate([r[4] for r in res])
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n
    assert np.array_equal(got_n, numer) and np.array_equal(got_d, denom)


def test_bench_multirank_plumbing_dry(tmp_path):
    """bench.py's N>1 control flow (env parsing, process group, table broadcast, equal-area
    sharding, barrier + MAX-reduce timing, single JSON line from rank 0) under torchrun with
    gloo and 2 ranks; kernels are stubbed (--dry-cpu), the line is marked dry."""
    import json
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-cpu", "--n-sketches", "300", "--no-cpu",
           "--detail", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                # only rank 0 prints; the headline is the LAST line
    assert r.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 4096
    h = json.loads(lines[0])
    assert h["dry"] is True and h["n_gpus"] == 2 and h["steps"] == 2 and h["scaling"] == "strong"
    assert h["rank_blocks"][0] == 0 and h["rank_blocks"][-1] == 300 and len(h["rank_blocks"]) == 3
    assert h["config"]["parallelism"] == "rowblock2" and h["config"]["table_broadcast_ms"] > 0
    d = json.loads(open(tmp_path / "detail.json").read())      # everything else: the detail file
    # the blocks were re-cut after the measured step: a row's cost in pairs came out of the all-reduce, and with it the
    # first block is smaller than its equal-area size
    from mash_amd import shard
    assert d["config"]["row_weight_pairs"] > 0 and d["config"]["rank_row_blocks"] == d["rank_blocks"]
    assert d["rank_blocks"][1] < shard.equal_area_row_blocks(300, 2)[1]


# ------------------------------------------------------------------ read-sharded screen

def _screen_case():
    """small mixture + query sketches; expected counts by direct hashing with the oracle"""
    from workloads import synth
    from oracle import pyoracle
    orc = pyoracle.Oracle()
    rng = np.random.default_rng(5)
    k, s = 21, 60
    genomes = [synth._rand_dna(rng, 3000) for _ in range(3)]
    p = orc.params(k=k, s=s)
    db = [orc.sketch_records([g], p)[0] for g in genomes]
    reads = []
    for _ in range(240):
        g = genomes[int(rng.integers(0, 2))]
        st = int(rng.integers(0, 3000 - 100))
        reads.append(g[st:st + 100])
    batches = [reads[i:i + 30] for i in range(0, len(reads), 30)]         # 8 batches
    return orc, k, s, db, batches


def _oracle_local_screen(orc, k, s, db):
    """what one rank computes on its share (the GPU does this through mg_screen_*)"""
    import torch

    def run(my_batches):
        seen = {}
        for b in my_batches:
            for r in b:
                h, c, _, _, _ = orc.sketch_records([r], orc.params(k=k, s=100000))
                for hv, cv in zip(h, c):
                    seen[int(hv)] = seen.get(int(hv), 0) + int(cv)
        counts = np.array([[seen.get(int(x), 0) for x in row] for row in db], dtype=np.int32)
        mix = np.array(sorted(seen), dtype=np.uint64)[:s]
        return torch.from_numpy(counts.reshape(-1)), mix
    return run


def _screen_worker(rank, world, port, q, sparse_below=0.05):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from mash_amd import screen_dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc, k, s, db, batches = _screen_case()
    counts, mix = screen_dist.screen_sharded(_oracle_local_screen(orc, k, s, db), batches, s, sparse_below=sparse_below)
    q.put((rank, counts.numpy().copy(), mix.copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sparse_below", [0.0, 1.0])
def test_two_rank_screen_allreduce_and_mixture_merge(oracle, sparse_below):
    """counts summed over ranks (dense all-reduce / sparse index-count all-gather) and the merged
    mixture equal the single-process result"""
    import torch.multiprocessing as mp
    from mash_amd import screen_dist
    assert screen_dist.shard_batches(5, 1, 2) == [1, 3]
    a = np.array([1, 5, 9], np.uint64); b = np.array([2, 5, 7, 11], np.uint64)
    assert list(screen_dist.merge_mixtures([a, b], 4)) == [1, 2, 5, 7]
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_screen_worker, args=(r, 2, port, q, sparse_below)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    orc, k, s, db, batches = _screen_case()
    want_counts, want_mix = _oracle_local_screen(orc, k, s, db)(batches)
    for r in res:
        assert np.array_equal(r[1], want_counts.numpy())
        assert np.array_equal(r[2], want_mix)
    assert want_counts.numpy().reshape(3, -1)[2].sum() <= want_counts.numpy().reshape(3, -1)[0].sum()

