"""Read-sharded `mash screen` across ranks (BASELINE config 4).

Every rank holds the same query-sketch table `db` and screens ITS share of the mixture
(the reference already cuts the mixture into independent ~1 MiB chunks,
CommandScreen.cpp:192,224-249).  Two things are exchanged at the end, and nothing before:

  * the per-hash observation counts, u32[db_rows * s]: summed over ranks -- the one
    collective of the data path (RCCL over xGMI on GPUs; gloo on CPU tests): an all-gather of
    the non-zero (index, count) lists while they are sparse, a dense all-reduce otherwise.
    It replaces the shared atomic `hashCounts` map of the reference (CommandScreen.h:131);
  * each rank's bottom-s sketch of its share of the mixture: all-gathered (s u64 per rank)
    and merged -- bottom-s of the union of bottom-s sets is the bottom-s of the whole
    mixture, which is what the reference's heap merge computes (CommandScreen.cpp:288-302).

`local_screen` abstracts the device work so the exchange logic can run under gloo without a
GPU: it returns (counts tensor on `device`, u32 stored as int32/int64; mixture numpy u64).
"""
import numpy as np
import torch
import torch.distributed as dist

HASH_PAD = np.uint64(0xFFFFFFFFFFFFFFFF)


def shard_batches(n_batches, rank, world):
    """batch indices of this rank: round-robin, like the reference hands chunks to threads"""
    return list(range(rank, n_batches, world))


def merge_mixtures(mixes, s):
    """bottom-s distinct of the union of ascending distinct u64 arrays"""
    if not mixes:
        return np.zeros(0, dtype=np.uint64)
    u = np.unique(np.concatenate([np.asarray(m, dtype=np.uint64) for m in mixes]))
    return u[:s]


def sum_counts(counts, group=None, sparse_below=0.05):
    """Sum the observation counters over ranks, in place.  A mixture touches few of a large
    database's hashes (config 4: 7e5 of 1e8 counters per rank are non-zero), so when every
    rank's non-zero share is below `sparse_below` the ranks all-gather their (index, count)
    lists -- megabytes -- instead of all-reducing the dense vector (400 MB at config 4);
    otherwise one dense all-reduce."""
    world = dist.get_world_size(group)
    idx = torch.nonzero(counts).squeeze(1)
    nnz = torch.tensor([idx.numel()], dtype=torch.int64, device=counts.device)
    dist.all_reduce(nnz, op=dist.ReduceOp.MAX, group=group)
    m = int(nnz.item())
    if m > sparse_below * counts.numel():
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
        return counts
    pay = torch.zeros((2, m + 1), dtype=torch.int64, device=counts.device)       # row 0: indices, row 1: counts
    pay[0, : idx.numel()] = idx
    pay[1, : idx.numel()] = counts[idx].to(torch.int64)
    pay[0, m] = idx.numel()
    parts = [torch.empty_like(pay) for _ in range(world)]
    dist.all_gather(parts, pay, group=group)
    counts.zero_()
    for p in parts:
        k = int(p[0, m].item())
        if k:
            counts.index_add_(0, p[0, :k], p[1, :k].to(counts.dtype))
    return counts


def exchange(counts, mix, s, group=None, sparse_below=0.05):
    """sum the counts over ranks in place and merge the mixtures of all ranks.
    counts: integer tensor on the collective's device; mix: numpy u64 (<= s, ascending)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return counts, np.asarray(mix, dtype=np.uint64)[:s]
    sum_counts(counts, group, sparse_below)
    # fixed-size payload: s hashes (padded) + the count, as int64 bit patterns
    pay = np.full(s + 1, HASH_PAD, dtype=np.uint64)
    pay[: len(mix)] = mix
    pay[s] = len(mix)
    mine = torch.from_numpy(pay.view(np.int64).copy()).to(counts.device)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    mixes = []
    for g in gathered:
        a = g.cpu().numpy().view(np.uint64)
        mixes.append(a[: int(a[s])])
    return counts, merge_mixtures(mixes, s)


def screen_sharded(local_screen, batches, s, group=None, sparse_below=0.05):
    """local_screen(list of this rank's batches) -> (counts tensor, mixture u64 array).
    Returns (summed counts tensor, merged mixture) on every rank."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = [batches[i] for i in shard_batches(len(batches), rank, world)]
    counts, mix = local_screen(mine)
    return exchange(counts, mix, s, group, sparse_below)


def gpu_local_screen(eng, db, p, resident=False):
    """local_screen over libmashgpu: batches are lists of record bytes (host) or
    (device_ptr, nbytes, keepalive) tuples; counts stay on the GPU for the collective.
    resident: ONE mg_screen for all calls (the database's key table is built once, every call starts with
    mg_screen_reset); run.close() frees it."""
    state = {"sc": None}

    def run(my_batches):
        if resident:
            if state["sc"] is None:
                state["sc"] = eng.screen_open(db, p)
            else:
                state["sc"].reset()
            sc = state["sc"]
        else:
            sc = eng.screen_open(db, p)
        try:
            for b in my_batches:
                if isinstance(b, tuple):
                    sc.add_dev(b[0], b[1])
                else:
                    sc.add_records(b)
            counts = torch.empty(db.rows * db.sketch_size, dtype=torch.int32, device="cuda")
            sc.counts_dev(counts.data_ptr())
            _, mix, _ = sc.finish(want_counts=False, want_distinct=False)
        finally:
            if not resident:
                sc.close()
        return counts, mix

    def close():
        if state["sc"] is not None:
            state["sc"].close()
            state["sc"] = None
    run.close = close
    return run
