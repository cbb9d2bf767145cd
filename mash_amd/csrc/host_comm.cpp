// host_comm.cpp -- several GPUs: communicator, replicated / row-sharded tables, sharded compare and sketch calls
#include "host_internal.h"

/* ------------------------------------------------- several GPUs: communicator, replicated tables, row-block sharding */

// every pair is independent, so the all-pairs matrix is cut into row blocks,
// one per GPU, against a sketch table that is resident on every GPU.  The one exchange is the
// BROADCAST of that table from GPU 0 (RCCL over xGMI); the compare data path has no collective.
// Two shapes of the same thing:
//   local : one process drives every GPU (the `mash` CLI): a context per device, ncclCommInitAll;
//   rank  : one process per GPU (bench.py under torchrun): ncclCommInitRank on an id the caller
//           hands round (128 bytes, any transport).
int comm_fail(mg_comm *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else (void)fail(nullptr, code, msg);
    return code;
}

#define NCCL_TRY(c, call)                                                             \
    do {                                                                              \
        ncclResult_t r__ = (call);                                                    \
        if (r__ != ncclSuccess) return comm_fail((c), MG_ERR_HIP, std::string(#call) + ": " + ncclGetErrorString(r__)); \
    } while (0)

int mg_comm_create_local(const int *devices, int n, mg_comm **out)
{
    if (!out || !devices || n < 1) return comm_fail(nullptr, MG_ERR_INVALID, "mg_comm_create_local: bad argument");
    mg_comm *c = new mg_comm;
    c->local = true;
    c->nranks = n;
    bool distinct = true;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++) distinct = distinct && devices[i] != devices[j];
    for (int i = 0; i < n; i++) {
        mg_ctx *x = nullptr;
        const int rc = mg_ctx_create(devices[i], &x);
        if (rc != MG_OK) { mg_comm_destroy(c); return rc; }          // g_create_error holds the text
        c->ctxs.push_back(x);
    }
    // RCCL needs distinct devices; a list that repeats a device (tests on a one-GPU box: two
    // contexts on one device) exchanges by plain device copies instead.  One device needs nothing,
    // unless MASHGPU_COMM_FORCE_RCCL asks for the one-rank communicator (tests of the call path).
    if (distinct && (n > 1 || getenv("MASHGPU_COMM_FORCE_RCCL"))) {        // (a process-wide test knob: the communicator creates its contexts itself)
        c->comms.resize((size_t)n);
        const ncclResult_t r = ncclCommInitAll(c->comms.data(), n, devices);
        if (r != ncclSuccess) {
            c->comms.clear();
            const std::string msg = std::string("ncclCommInitAll: ") + ncclGetErrorString(r);
            mg_comm_destroy(c);
            return comm_fail(nullptr, MG_ERR_HIP, msg);
        }
    }
    *out = c;
    return MG_OK;
}

int mg_comm_unique_id(void *id_out, size_t id_bytes)
{
    if (!id_out || id_bytes < sizeof(ncclUniqueId)) return comm_fail(nullptr, MG_ERR_INVALID, "mg_comm_unique_id: buffer too small (128 bytes)");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return comm_fail(nullptr, MG_ERR_HIP, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    memcpy(id_out, &id, sizeof id);
    return MG_OK;
}

int mg_comm_create_rank(mg_ctx *ctx, const void *id, size_t id_bytes, int nranks, int rank, mg_comm **out)
{
    if (!ctx || !out || !id || id_bytes < sizeof(ncclUniqueId) || nranks < 1 || rank < 0 || rank >= nranks)
        return comm_fail(nullptr, MG_ERR_INVALID, "mg_comm_create_rank: bad argument");
    if (hipSetDevice(ctx->device) != hipSuccess) return comm_fail(nullptr, MG_ERR_HIP, "mg_comm_create_rank: hipSetDevice failed");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t nc;
    const ncclResult_t r = ncclCommInitRank(&nc, nranks, uid, rank);
    if (r != ncclSuccess) return comm_fail(nullptr, MG_ERR_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    mg_comm *c = new mg_comm;
    c->local = false;
    c->nranks = nranks;
    c->rank = rank;
    c->ctxs.push_back(ctx);
    c->comms.push_back(nc);
    *out = c;
    return MG_OK;
}

void mg_comm_destroy(mg_comm *c)
{
    if (!c) return;
    for (size_t i = 0; i < c->comms.size(); i++) {
        hipSetDevice(c->ctxs[i]->device);
        ncclCommDestroy(c->comms[i]);
    }
    if (c->local) for (mg_ctx *x : c->ctxs) mg_ctx_destroy(x);
    delete c;
}

int mg_comm_size(const mg_comm *c) { return c ? c->nranks : 0; }
int mg_comm_rank(const mg_comm *c) { return c ? c->rank : -1; }
int mg_comm_uses_rccl(const mg_comm *c) { return c && !c->comms.empty() ? 1 : 0; }
mg_ctx *mg_comm_ctx(mg_comm *c, int i) { return c && i >= 0 && (size_t)i < c->ctxs.size() ? c->ctxs[(size_t)i] : nullptr; }
const char *mg_comm_last_error(mg_comm *c) { return c ? c->err.c_str() : mg_last_error(nullptr); }

// Equal-AREA row blocks of the lower triangle (row i holds i pairs): block g of G over rows
// [row_begin, row_end) starts where g/G of the pairs lie behind -- boundaries go with sqrt(g/G).
void mg_shard_tri_rows(uint64_t row_begin, uint64_t row_end, int nranks, int rank, uint64_t *b_out, uint64_t *e_out)
{
    auto boundary = [&](int g) -> uint64_t {
        if (g <= 0) return row_begin;
        if (g >= nranks) return row_end;
        const long double total = (long double)tri_pairs(row_begin, row_end);
        const long double want = total * g / nranks + (long double)tri_pairs(0, row_begin);
        uint64_t r = (uint64_t)((1.0L + sqrtl(1.0L + 8.0L * want)) * 0.5L);
        if (r < row_begin) r = row_begin;
        if (r > row_end) r = row_end;
        while (r > row_begin && (long double)tri_pairs(0, r) > want) r--;
        while (r < row_end && (long double)tri_pairs(0, r + 1) <= want) r++;
        return r;
    };
    if (b_out) *b_out = boundary(rank);
    if (e_out) *e_out = boundary(rank + 1);
}

// The same with a cost per ROW on top of the cost per pair: row i costs i + row_weight pair-units.  The inverted-index
// engine fills 8 bytes per pair but discovers and merges per row (C3: a row costs what 60 000 pairs cost), so equal
// areas give the first block -- the short rows, a third of all rows at 8 blocks -- far more than its share.
// row_weight 0 = mg_shard_tri_rows.
void mg_shard_tri_rows_weighted(uint64_t row_begin, uint64_t row_end, int nranks, int rank, double row_weight, uint64_t *b_out,
                                uint64_t *e_out)
{
    if (!(row_weight > 0)) { mg_shard_tri_rows(row_begin, row_end, nranks, rank, b_out, e_out); return; }
    const long double w = (long double)row_weight;
    auto cost_below = [&](uint64_t r) -> long double { return (long double)tri_pairs(0, r) + w * (long double)r; };   // rows [0, r)
    auto boundary = [&](int g) -> uint64_t {
        if (g <= 0) return row_begin;
        if (g >= nranks) return row_end;
        const long double lo = cost_below(row_begin), want = lo + (cost_below(row_end) - lo) * g / nranks;
        // r(r - 1)/2 + w r = want  ->  r = (1/2 - w) + sqrt((w - 1/2)^2 + 2 want)
        const long double h = w - 0.5L;
        long double rr = -h + sqrtl(h * h + 2.0L * want);
        uint64_t r = rr <= (long double)row_begin ? row_begin : rr >= (long double)row_end ? row_end : (uint64_t)rr;
        while (r > row_begin && cost_below(r) > want) r--;
        while (r < row_end && cost_below(r + 1) <= want) r++;
        return r;
    };
    if (b_out) *b_out = boundary(rank);
    if (e_out) *e_out = boundary(rank + 1);
}

// ... and with a cost per row of the table BELOW a block's end: a rank derives what it needs -- the inverted index above all --
// from the view of the rows below its block's end (host_compare.cpp: tri_view), so its block [lo, hi) costs
//     pairs(lo, hi) + row_weight (hi - lo) + prefix_weight hi
// pair-units.  The blocks of equal cost are found by bisection on that cost: for a cost T the blocks are laid one after the other
// (each as long as T allows); the smallest T whose G blocks reach row_end is the answer.  Every rank computes all boundaries.
void mg_shard_tri_rows_costed(uint64_t row_begin, uint64_t row_end, int nranks, int rank, double row_weight, double prefix_weight,
                              uint64_t *b_out, uint64_t *e_out)
{
    if (!(prefix_weight > 0)) { mg_shard_tri_rows_weighted(row_begin, row_end, nranks, rank, row_weight, b_out, e_out); return; }
    if (nranks < 1) nranks = 1;
    const long double w = row_weight > 0 ? (long double)row_weight : 0.0L, v = (long double)prefix_weight;
    auto cost = [&](uint64_t lo, uint64_t hi) -> long double {
        return (long double)tri_pairs(lo, hi) + w * (long double)(hi - lo) + v * (long double)hi;
    };
    std::vector<uint64_t> b((size_t)nranks + 1, row_end);
    auto lay = [&](long double T) -> bool {                // blocks of cost <= T, one after the other; true: they reach row_end
        uint64_t lo = row_begin;
        b[0] = row_begin;
        for (int g = 0; g < nranks; g++) {
            uint64_t a = lo, e = row_end;                    // the largest hi in [lo, row_end] with cost(lo, hi) <= T (hi = lo: an empty block)
            if (cost(lo, e) <= T) a = e;
            else while (e - a > 1) { const uint64_t m = a + (e - a) / 2; if (cost(lo, m) <= T) a = m; else e = m; }
            if (a > lo && cost(lo, a) > T) a = lo;
            b[(size_t)g + 1] = a;
            lo = a;
        }
        return lo >= row_end;
    };
    long double t_lo = 0.0L, t_hi = cost(row_begin, row_end);
    for (int it = 0; it < 200 && t_hi - t_lo > 0.5L; it++) {
        const long double mid = (t_lo + t_hi) * 0.5L;
        if (lay(mid)) t_hi = mid; else t_lo = mid;
    }
    (void)lay(t_hi);
    b[(size_t)nranks] = row_end;
    if (rank < 0) rank = 0;
    if (rank >= nranks) rank = nranks - 1;
    if (b_out) *b_out = b[(size_t)rank];
    if (e_out) *e_out = b[(size_t)rank + 1];
}

void mg_shard_rows(uint64_t row_begin, uint64_t row_end, int nranks, int rank, uint64_t *b_out, uint64_t *e_out)
{
    const uint64_t n = row_end > row_begin ? row_end - row_begin : 0;
    if (b_out) *b_out = row_begin + n * (uint64_t)rank / (uint64_t)nranks;
    if (e_out) *e_out = row_begin + n * (uint64_t)(rank + 1) / (uint64_t)nranks;
}

// src (root's buffers, device memory of context `root`) -> dst buffers on every context; count bytes
static int comm_broadcast_bytes(mg_comm *c, int root, const std::vector<void *> &bufs, size_t bytes)
{
    if (bytes == 0) return MG_OK;
    const size_t n = c->ctxs.size();
    if (!c->comms.empty()) {
        NCCL_TRY(c, ncclGroupStart());
        for (size_t i = 0; i < n; i++) {
            const ncclResult_t r = ncclBroadcast(bufs[(size_t)root], bufs[i], bytes, ncclUint8, root, c->comms[i], c->ctxs[i]->stream);
            if (r != ncclSuccess) { ncclGroupEnd(); return comm_fail(c, MG_ERR_HIP, std::string("ncclBroadcast: ") + ncclGetErrorString(r)); }
        }
        NCCL_TRY(c, ncclGroupEnd());
    } else {
        for (size_t i = 0; i < n; i++) {
            if ((int)i == root || bufs[i] == bufs[(size_t)root]) continue;
            if (hipMemcpyPeerAsync(bufs[i], c->ctxs[i]->device, bufs[(size_t)root], c->ctxs[(size_t)root]->device, bytes,
                                   c->ctxs[(size_t)root]->stream) != hipSuccess)
                return comm_fail(c, MG_ERR_HIP, "table broadcast: device copy failed");
        }
    }
    return MG_OK;
}

int comm_sync_all(mg_comm *c)
{
    for (mg_ctx *x : c->ctxs) {
        if (hipSetDevice(x->device) != hipSuccess || hipStreamSynchronize(x->stream) != hipSuccess)
            return comm_fail(c, MG_ERR_HIP, "communicator: stream synchronisation failed");
    }
    return MG_OK;
}

int mg_dtable_upload(mg_comm *c, const uint64_t *hashes, const uint32_t *nhash, const uint64_t *lengths, uint64_t n,
                     uint64_t s, mg_dtable **out)
{
    if (!c || !c->local || !out) return comm_fail(c, MG_ERR_INVALID, "mg_dtable_upload: needs a local communicator");
    mg_dtable *d = new mg_dtable;
    d->comm = c;
    d->n = n;
    d->s = s;
    mg_table *t0 = nullptr;
    int rc = mg_table_upload(c->ctxs[0], hashes, nhash, lengths, n, s, &t0);   // host -> GPU 0
    if (rc != MG_OK) { c->err = c->ctxs[0]->err; delete d; return rc; }
    d->t.push_back(t0);
    const size_t G = c->ctxs.size();
    std::vector<void *> bh{(void *)t0->hashes}, bn{(void *)t0->nhash}, bl{(void *)t0->lengths};
    for (size_t i = 1; i < G; i++) {
        mg_ctx *x = c->ctxs[i];
        void *ph = nullptr, *pn = nullptr, *pl = nullptr;
        if (hipSetDevice(x->device) != hipSuccess || hipMalloc(&ph, std::max<uint64_t>(n * s, 1) * 8) != hipSuccess ||
            hipMalloc(&pn, std::max<uint64_t>(n, 1) * 4) != hipSuccess || hipMalloc(&pl, std::max<uint64_t>(n, 1) * 8) != hipSuccess) {
            mg_dtable_free(d);
            return comm_fail(c, MG_ERR_NOMEM, "mg_dtable_upload: device allocation failed");
        }
        mg_table *t = new mg_table;
        t->ctx = x; t->hashes = (const uint64_t *)ph; t->nhash = (const uint32_t *)pn; t->lengths = (const uint64_t *)pl;
        t->n = n; t->s = s; t->owns = true;
        d->t.push_back(t);
        bh.push_back(ph); bn.push_back(pn); bl.push_back(pl);
    }
    // GPU 0 -> every GPU: the one exchange of the all-pairs job
    rc = comm_broadcast_bytes(c, 0, bh, n * s * 8);
    if (rc == MG_OK) rc = comm_broadcast_bytes(c, 0, bn, n * 4);
    if (rc == MG_OK) rc = comm_broadcast_bytes(c, 0, bl, n * 8);
    if (rc == MG_OK) rc = comm_sync_all(c);
    if (rc != MG_OK) { mg_dtable_free(d); return rc; }
    *out = d;
    return MG_OK;
}

void mg_dtable_free(mg_dtable *d)
{
    if (!d) return;
    for (auto &v : d->views) mg_table_free(v.t);
    for (mg_table *t : d->t) mg_table_free(t);
    delete d;
}

// The LARGER side of a rect job need not be replicated: every device gets a block of consecutive rows
// (host -> each GPU its own rows, no exchange).  Such a table is the reference side of
// mg_compare_rect_*_sharded_host, which then splits the job by reference rows ("broadcast
// the smaller side, shard the larger side by rows").
int mg_dtable_upload_rows(mg_comm *c, const uint64_t *hashes, const uint32_t *nhash, const uint64_t *lengths, uint64_t n,
                          uint64_t s, mg_dtable **out)
{
    if (!c || !c->local || !out) return comm_fail(c, MG_ERR_INVALID, "mg_dtable_upload_rows: needs a local communicator");
    if (!hashes || !nhash || s == 0) return comm_fail(c, MG_ERR_INVALID, "mg_dtable_upload_rows: bad argument");
    mg_dtable *d = new mg_dtable;
    d->comm = c;
    d->by_rows = true;
    d->n = n;
    d->s = s;
    const int G = (int)c->ctxs.size();
    d->row0.resize((size_t)G + 1);
    for (int g = 0; g < G; g++) {
        uint64_t lo, hi;
        mg_shard_rows(0, n, G, g, &lo, &hi);
        d->row0[(size_t)g] = lo;
        d->row0[(size_t)g + 1] = hi;
    }
    d->t.assign((size_t)G, nullptr);
    std::vector<int> rcs((size_t)G, MG_OK);
    std::vector<std::thread> th;
    auto up = [&](int g) {
        const uint64_t lo = d->row0[(size_t)g], hi = d->row0[(size_t)g + 1];
        // (an empty block still gets a table: one padding row, zero rows visible)
        rcs[(size_t)g] = mg_table_upload(c->ctxs[(size_t)g], hashes + lo * s, nhash + lo, lengths ? lengths + lo : nullptr, hi - lo, s, &d->t[(size_t)g]);
    };
    for (int g = 0; g < G; g++) {
        if (G == 1) up(g); else th.emplace_back(up, g);
    }
    for (auto &t : th) t.join();
    for (int g = 0; g < G; g++)
        if (rcs[(size_t)g] != MG_OK) { c->err = c->ctxs[(size_t)g]->err; const int rc = rcs[(size_t)g]; mg_dtable_free(d); return rc; }
    *out = d;
    return MG_OK;
}

mg_table *mg_dtable_local(mg_dtable *d, int i) { return d && i >= 0 && (size_t)i < d->t.size() ? d->t[(size_t)i] : nullptr; }

// rank mode: the root's table -> a table on every rank (the root gets a non-owning alias of `src`)
int mg_table_broadcast(mg_comm *c, const mg_table *src, int root, uint64_t n, uint64_t s, mg_table **out)
{
    if (!c || c->local || !out || root < 0 || root >= c->nranks) return comm_fail(c, MG_ERR_INVALID, "mg_table_broadcast: needs a rank communicator");
    mg_ctx *x = c->ctxs[0];
    if (c->rank == root && (!src || src->n != n || src->s != s || !src->lengths))
        return comm_fail(c, MG_ERR_INVALID, "mg_table_broadcast: the root must pass the table (with lengths) and its true size");
    if (hipSetDevice(x->device) != hipSuccess) return comm_fail(c, MG_ERR_HIP, "hipSetDevice failed");
    mg_table *t = new mg_table;
    t->ctx = x; t->n = n; t->s = s;
    if (c->rank == root) {
        t->hashes = src->hashes; t->nhash = src->nhash; t->lengths = src->lengths; t->owns = false;
    } else {
        void *ph = nullptr, *pn = nullptr, *pl = nullptr;
        if (hipMalloc(&ph, std::max<uint64_t>(n * s, 1) * 8) != hipSuccess || hipMalloc(&pn, std::max<uint64_t>(n, 1) * 4) != hipSuccess ||
            hipMalloc(&pl, std::max<uint64_t>(n, 1) * 8) != hipSuccess) {
            delete t;
            return comm_fail(c, MG_ERR_NOMEM, "mg_table_broadcast: device allocation failed");
        }
        t->hashes = (const uint64_t *)ph; t->nhash = (const uint32_t *)pn; t->lengths = (const uint64_t *)pl; t->owns = true;
    }
    ncclResult_t r = ncclGroupStart();
    if (r == ncclSuccess) r = ncclBroadcast(t->hashes, (void *)t->hashes, n * s * 8, ncclUint8, root, c->comms[0], x->stream);
    if (r == ncclSuccess) r = ncclBroadcast(t->nhash, (void *)t->nhash, n * 4, ncclUint8, root, c->comms[0], x->stream);
    if (r == ncclSuccess) r = ncclBroadcast(t->lengths, (void *)t->lengths, n * 8, ncclUint8, root, c->comms[0], x->stream);
    const ncclResult_t r2 = ncclGroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess || hipStreamSynchronize(x->stream) != hipSuccess) {
        mg_table_free(t);
        return comm_fail(c, MG_ERR_HIP, std::string("mg_table_broadcast: ") + (r != ncclSuccess ? ncclGetErrorString(r) : "stream error"));
    }
    *out = t;
    return MG_OK;
}

// rank mode: element-wise sum of a u32 device buffer over all ranks (the counter exchange of a read-sharded screen)
int mg_comm_allreduce_u32_sum(mg_comm *c, uint32_t *buf_dev, uint64_t count)
{
    if (!c || c->local || (!buf_dev && count)) return comm_fail(c, MG_ERR_INVALID, "mg_comm_allreduce_u32_sum: needs a rank communicator");
    if (count == 0) return MG_OK;
    mg_ctx *x = c->ctxs[0];
    if (hipSetDevice(x->device) != hipSuccess) return comm_fail(c, MG_ERR_HIP, "hipSetDevice failed");
    NCCL_TRY(c, ncclAllReduce(buf_dev, buf_dev, count, ncclUint32, ncclSum, c->comms[0], x->stream));
    if (hipStreamSynchronize(x->stream) != hipSuccess) return comm_fail(c, MG_ERR_HIP, "mg_comm_allreduce_u32_sum: stream error");
    return MG_OK;
}

// local mode: rows [rb, re) cut into one block per GPU, every GPU driven by its own host thread;
// `fn(g, ctx, table replica(s), block begin, block end, pairs before the block)` does one block
// What a row of a triangle job costs beyond its pairs, in pairs (mg_shard_tri_rows_weighted): jobs large enough for the
// inverted-index engine fill per pair but discover and merge per row -- 60 s is C3's measured ratio (bench.py measures
// it per table; here a constant has to do: an all-random table has a third of it, clades seven times as much).
// MASHGPU_SHARD_ROW_WEIGHT overrides (0: equal areas).
static double tri_row_weight(const mg_ctx *ctx, uint64_t rb, uint64_t re, uint64_t s)
{
    if (const char *e = ctx_opt(ctx, "MASHGPU_SHARD_ROW_WEIGHT")) return atof(e);
    // only where the inverted-index engine takes the blocks (the tile engine costs per pair -- with a row weight
    // a few thousand rows were cut almost evenly by rows and the last device got ten times the first one's pairs) ...
    if (tri_pairs(rb, re) < 4000000ull) return 0.0;
    if (const char *e = ctx_opt(ctx, "MASHGPU_COMPARE_SPARSE")) { if (atoi(e) == 0) return 0.0; }
    if (const char *e = ctx_opt(ctx, "MASHGPU_COMPARE_KERNEL")) { if (strcmp(e, "sparse") != 0) return 0.0; }
    // ... and never more than a mean row's pairs: a row cannot cost more than it holds
    const double mean_row = (double)tri_pairs(rb, re) / (double)std::max<uint64_t>(re - rb, 1);
    return std::min(60.0 * (double)s, mean_row);
}

template <class F>
static int sharded_blocks(mg_comm *c, uint64_t rb, uint64_t re, bool triangle, uint64_t ncols, F fn, double row_weight = 0.0)
{
    const int G = (int)c->ctxs.size();
    std::vector<uint64_t> b((size_t)G + 1);
    for (int g = 0; g <= G; g++) {
        uint64_t lo, hi;
        if (triangle) mg_shard_tri_rows_weighted(rb, re, G, std::min(g, G - 1), row_weight, &lo, &hi);
        else mg_shard_rows(rb, re, G, std::min(g, G - 1), &lo, &hi);
        b[(size_t)g] = g < G ? lo : hi;
    }
    std::vector<int> rcs((size_t)G, MG_OK);
    std::vector<std::thread> th;
    for (int g = 0; g < G; g++) {
        const uint64_t lo = b[(size_t)g], hi = b[(size_t)g + 1];
        const uint64_t before = triangle ? tri_pairs(rb, lo) : (lo - rb) * ncols;
        if (lo >= hi) continue;
        if (G == 1) rcs[0] = fn(0, lo, hi, before);
        else th.emplace_back([&, g, lo, hi, before]() { rcs[(size_t)g] = fn(g, lo, hi, before); });
    }
    for (auto &t : th) t.join();
    for (int g = 0; g < G; g++)
        if (rcs[(size_t)g] != MG_OK) { c->err = c->ctxs[(size_t)g]->err; return rcs[(size_t)g]; }
    return MG_OK;
}

int dtable_check(mg_comm *c, const mg_dtable *t, const char *who, bool rows_ok)
{
    if (!c || !c->local || !t || t->comm != c || t->t.size() != c->ctxs.size())
        return comm_fail(c, MG_ERR_INVALID, std::string(who) + ": needs a local communicator and tables uploaded through it");
    if (t->by_rows && !rows_ok)
        return comm_fail(c, MG_ERR_INVALID, std::string(who) + ": a row-sharded table (mg_dtable_upload_rows) can only be the reference side of a rect job");
    return MG_OK;
}

// ---- rect jobs split by REFERENCE rows: device g compares every query with its block of
// reference rows -- its own rows of a row-sharded table, or a view of its replica's rows [lo, hi) -- and
// the blocks are put back into the reference's query-major order on the host.
static int ref_block(mg_comm *c, const mg_dtable *ref, size_t g, const mg_table **tab, uint64_t *lo_out, uint64_t *hi_out)
{
    const size_t G = c->ctxs.size();
    if (ref->by_rows) {
        *tab = ref->t[g];
        *lo_out = ref->row0[g];
        *hi_out = ref->row0[g + 1];
        return MG_OK;
    }
    uint64_t lo, hi;
    mg_shard_rows(0, ref->t[0]->n, (int)G, (int)g, &lo, &hi);
    *lo_out = lo;
    *hi_out = hi;
    std::lock_guard<std::mutex> lk(ref->views_mu);
    for (auto &v : ref->views)
        if (v.g == g && v.lo == lo && v.hi == hi) { *tab = v.t; return MG_OK; }
    const mg_table *full = ref->t[g];
    mg_table *view = nullptr;
    const int rc = mg_table_wrap_dev(c->ctxs[g], full->hashes + lo * full->s, full->nhash + lo, full->lengths ? full->lengths + lo : nullptr,
                                     hi - lo, full->s, &view);
    if (rc != MG_OK) return rc;
    ref->views.push_back({g, lo, hi, view});
    *tab = view;
    return MG_OK;
}

// dense outputs (mg_counts / mg_pair): call(g, ref block, query replica, q0, q1, out) fills (q1 - q0) x block rows
template <class T, class Call>
static int rect_by_ref_rows(mg_comm *c, const mg_dtable *ref, const mg_dtable *qry, uint64_t q_begin, uint64_t q_end, T *out_host, Call call)
{
    const size_t G = c->ctxs.size();
    const uint64_t nref = ref->by_rows ? ref->n : ref->t[0]->n;
    std::vector<int> rcs(G, MG_OK);
    std::vector<std::thread> th;
    auto work = [&](size_t g) {
        const mg_table *blk = nullptr;
        uint64_t lo = 0, hi = 0;
        int rc = ref_block(c, ref, g, &blk, &lo, &hi);
        if (rc != MG_OK || lo >= hi) { rcs[g] = rc; return; }
        const uint64_t w = hi - lo;
        // queries in blocks that bound the staging buffer (256 MiB)
        const uint64_t qstep = std::max<uint64_t>(1, (256ull << 20) / (w * sizeof(T)));
        std::vector<T> tmp;
        for (uint64_t q0 = q_begin; q0 < q_end && rc == MG_OK; q0 += qstep) {
            const uint64_t q1 = std::min(q_end, q0 + qstep);
            tmp.resize((q1 - q0) * w);
            rc = call(g, blk, qry->t[g], q0, q1, tmp.data());
            for (uint64_t q = q0; q < q1 && rc == MG_OK; q++)
                memcpy(out_host + (q - q_begin) * nref + lo, tmp.data() + (q - q0) * w, w * sizeof(T));
        }
        rcs[g] = rc;
    };
    for (size_t g = 0; g < G; g++) {
        if (G == 1) work(g); else th.emplace_back(work, g);
    }
    for (auto &t : th) t.join();
    for (size_t g = 0; g < G; g++)
        if (rcs[g] != MG_OK) { c->err = c->ctxs[g]->err; return rcs[g]; }
    return MG_OK;
}

// which side of a rect job is cut: the reference rows when that table is row-sharded or the larger side
static bool rect_split_refs(const mg_ctx *ctx, const mg_dtable *ref, uint64_t nq)
{
    if (ref->by_rows) return true;
    if (ctx_opt(ctx, "MASHGPU_RECT_SPLIT")) return strcmp(ctx_opt(ctx, "MASHGPU_RECT_SPLIT"), "refs") == 0;
    return ref->t.size() > 1 && ref->t[0]->n > nq;
}

int mg_compare_tri_sharded_host(mg_comm *c, const mg_dtable *t, uint64_t row_begin, uint64_t row_end, mg_counts *out_host)
{
    int rc = dtable_check(c, t, "mg_compare_tri_sharded_host");
    if (rc != MG_OK) return rc;
    if (row_end > t->t[0]->n) row_end = t->t[0]->n;
    if (row_begin >= row_end) return MG_OK;
    return sharded_blocks(c, row_begin, row_end, true, 0, [&](int g, uint64_t lo, uint64_t hi, uint64_t before) {
        return mg_compare_tri_host(c->ctxs[(size_t)g], t->t[(size_t)g], lo, hi, out_host + before);
    }, tri_row_weight(c->ctxs[0], row_begin, row_end, t->t[0]->s));
}

int mg_compare_rect_sharded_host(mg_comm *c, const mg_dtable *ref, const mg_dtable *qry, uint64_t q_begin, uint64_t q_end,
                                 mg_counts *out_host)
{
    int rc = dtable_check(c, ref, "mg_compare_rect_sharded_host", true);
    if (rc == MG_OK) rc = dtable_check(c, qry, "mg_compare_rect_sharded_host");
    if (rc != MG_OK) return rc;
    if (q_end > qry->t[0]->n) q_end = qry->t[0]->n;
    if (q_begin >= q_end) return MG_OK;
    if (rect_split_refs(c->ctxs[0], ref, q_end - q_begin))
        return rect_by_ref_rows<mg_counts>(c, ref, qry, q_begin, q_end, out_host,
                                           [&](size_t g, const mg_table *blk, const mg_table *q, uint64_t q0, uint64_t q1, mg_counts *o) {
            return mg_compare_rect_host(c->ctxs[g], blk, q, q0, q1, o);
        });
    const uint64_t nref = ref->t[0]->n;
    return sharded_blocks(c, q_begin, q_end, false, nref, [&](int g, uint64_t lo, uint64_t hi, uint64_t before) {
        return mg_compare_rect_host(c->ctxs[(size_t)g], ref->t[(size_t)g], qry->t[(size_t)g], lo, hi, out_host + before);
    });
}

int mg_compare_tri_pairs_sharded_host(mg_comm *c, const mg_dtable *t, uint64_t row_begin, uint64_t row_end, int kmer_size,
                                      double kmer_space, double max_distance, double max_p_value, mg_pair *out_host)
{
    int rc = dtable_check(c, t, "mg_compare_tri_pairs_sharded_host");
    if (rc != MG_OK) return rc;
    if (row_end > t->t[0]->n) row_end = t->t[0]->n;
    if (row_begin >= row_end) return MG_OK;
    return sharded_blocks(c, row_begin, row_end, true, 0, [&](int g, uint64_t lo, uint64_t hi, uint64_t before) {
        return mg_compare_tri_pairs_host(c->ctxs[(size_t)g], t->t[(size_t)g], lo, hi, kmer_size, kmer_space, max_distance,
                                         max_p_value, out_host + before);
    }, tri_row_weight(c->ctxs[0], row_begin, row_end, t->t[0]->s));
}

int mg_compare_rect_pairs_sharded_host(mg_comm *c, const mg_dtable *ref, const mg_dtable *qry, uint64_t q_begin, uint64_t q_end,
                                       int kmer_size, double kmer_space, double max_distance, double max_p_value,
                                       mg_pair *out_host)
{
    int rc = dtable_check(c, ref, "mg_compare_rect_pairs_sharded_host", true);
    if (rc == MG_OK) rc = dtable_check(c, qry, "mg_compare_rect_pairs_sharded_host");
    if (rc != MG_OK) return rc;
    if (q_end > qry->t[0]->n) q_end = qry->t[0]->n;
    if (q_begin >= q_end) return MG_OK;
    if (rect_split_refs(c->ctxs[0], ref, q_end - q_begin))
        return rect_by_ref_rows<mg_pair>(c, ref, qry, q_begin, q_end, out_host,
                                         [&](size_t g, const mg_table *blk, const mg_table *q, uint64_t q0, uint64_t q1, mg_pair *o) {
            return mg_compare_rect_pairs_host(c->ctxs[g], blk, q, q0, q1, kmer_size, kmer_space, max_distance, max_p_value, o);
        });
    const uint64_t nref = ref->t[0]->n;
    return sharded_blocks(c, q_begin, q_end, false, nref, [&](int g, uint64_t lo, uint64_t hi, uint64_t before) {
        return mg_compare_rect_pairs_host(c->ctxs[(size_t)g], ref->t[(size_t)g], qry->t[(size_t)g], lo, hi, kmer_size, kmer_space,
                                          max_distance, max_p_value, out_host + before);
    });
}

// survivors of both filters: every GPU collects its block's list, the lists are joined in block (= reference) order
template <class Call>
static int sharded_results(mg_comm *c, uint64_t rb, uint64_t re, bool triangle, uint64_t ncols, mg_result *out_host,
                           uint64_t capacity, uint64_t *count_out, Call call, double row_weight = 0.0)
{
    const size_t G = c->ctxs.size();
    std::vector<std::vector<mg_result>> part(G);
    const int rc = sharded_blocks(c, rb, re, triangle, ncols, [&](int g, uint64_t lo, uint64_t hi, uint64_t) {
        std::vector<mg_result> &v = part[(size_t)g];
        v.resize(1u << 16);
        uint64_t n = 0;
        int r = call(g, lo, hi, v.data(), (uint64_t)v.size(), &n);
        if (r == MG_ERR_NOMEM && n > v.size()) {
            v.resize(n);
            r = call(g, lo, hi, v.data(), (uint64_t)v.size(), &n);
        }
        v.resize(r == MG_OK ? n : 0);
        return r;
    }, row_weight);
    if (rc != MG_OK) return rc;
    uint64_t total = 0;
    for (auto &v : part) total += v.size();
    *count_out = total;
    if (total > capacity) return comm_fail(c, MG_ERR_NOMEM, "compare: more passing pairs than `capacity` (see *count_out)");
    uint64_t at = 0;
    for (auto &v : part) {
        if (!v.empty()) memcpy(out_host + at, v.data(), v.size() * sizeof(mg_result));
        at += v.size();
    }
    return MG_OK;
}

int mg_compare_tri_results_sharded_host(mg_comm *c, const mg_dtable *t, uint64_t row_begin, uint64_t row_end, int kmer_size,
                                        double kmer_space, double max_distance, double max_p_value, mg_result *out_host,
                                        uint64_t capacity, uint64_t *count_out)
{
    int rc = dtable_check(c, t, "mg_compare_tri_results_sharded_host");
    if (rc != MG_OK) return rc;
    if (!count_out || (!out_host && capacity)) return comm_fail(c, MG_ERR_INVALID, "mg_compare_tri_results_sharded_host: NULL argument");
    *count_out = 0;
    if (row_end > t->t[0]->n) row_end = t->t[0]->n;
    if (row_begin >= row_end) return MG_OK;
    return sharded_results(c, row_begin, row_end, true, 0, out_host, capacity, count_out,
                           [&](int g, uint64_t lo, uint64_t hi, mg_result *o, uint64_t cap, uint64_t *n) {
        return mg_compare_tri_results_host(c->ctxs[(size_t)g], t->t[(size_t)g], lo, hi, kmer_size, kmer_space, max_distance,
                                           max_p_value, o, cap, n);
    }, 10.0 * tri_row_weight(c->ctxs[0], row_begin, row_end, t->t[0]->s));      // (thresholded: no matrix is filled, the cost is nearly all per row)
}

int mg_compare_rect_results_sharded_host(mg_comm *c, const mg_dtable *ref, const mg_dtable *qry, uint64_t q_begin,
                                         uint64_t q_end, int kmer_size, double kmer_space, double max_distance,
                                         double max_p_value, mg_result *out_host, uint64_t capacity, uint64_t *count_out)
{
    int rc = dtable_check(c, ref, "mg_compare_rect_results_sharded_host", true);
    if (rc == MG_OK) rc = dtable_check(c, qry, "mg_compare_rect_results_sharded_host");
    if (rc != MG_OK) return rc;
    if (!count_out || (!out_host && capacity)) return comm_fail(c, MG_ERR_INVALID, "mg_compare_rect_results_sharded_host: NULL argument");
    *count_out = 0;
    if (q_end > qry->t[0]->n) q_end = qry->t[0]->n;
    if (q_begin >= q_end) return MG_OK;
    if (rect_split_refs(c->ctxs[0], ref, q_end - q_begin)) {
        // every device lists the survivors of its reference block (query major, columns relative to the block);
        // the reference order is query major over ALL references: per query, the blocks' runs in block order
        const size_t G = c->ctxs.size();
        std::vector<std::vector<mg_result>> part(G);
        std::vector<uint64_t> lo_of(G, 0);
        std::vector<int> rcs(G, MG_OK);
        std::vector<std::thread> th;
        auto work = [&](size_t g) {
            const mg_table *blk = nullptr;
            uint64_t lo = 0, hi = 0;
            int r = ref_block(c, ref, g, &blk, &lo, &hi);
            lo_of[g] = lo;
            if (r != MG_OK || lo >= hi) { rcs[g] = r; return; }
            std::vector<mg_result> &v = part[g];
            v.resize(1u << 16);
            uint64_t n = 0;
            r = mg_compare_rect_results_host(c->ctxs[g], blk, qry->t[g], q_begin, q_end, kmer_size, kmer_space, max_distance, max_p_value,
                                             v.data(), (uint64_t)v.size(), &n);
            if (r == MG_ERR_NOMEM && n > v.size()) {
                v.resize(n);
                r = mg_compare_rect_results_host(c->ctxs[g], blk, qry->t[g], q_begin, q_end, kmer_size, kmer_space, max_distance, max_p_value,
                                                 v.data(), (uint64_t)v.size(), &n);
            }
            v.resize(r == MG_OK ? n : 0);
            rcs[g] = r;
        };
        for (size_t g = 0; g < G; g++) {
            if (G == 1) work(g); else th.emplace_back(work, g);
        }
        for (auto &t : th) t.join();
        for (size_t g = 0; g < G; g++)
            if (rcs[g] != MG_OK) { c->err = c->ctxs[g]->err; return rcs[g]; }
        uint64_t total = 0;
        for (auto &v : part) total += v.size();
        *count_out = total;
        if (total > capacity) return comm_fail(c, MG_ERR_NOMEM, "compare: more passing pairs than `capacity` (see *count_out)");
        std::vector<size_t> cur(G, 0);
        uint64_t at = 0;
        for (uint64_t q = q_begin; q < q_end; q++)                 // (rows of the results are query indices)
            for (size_t g = 0; g < G; g++) {
                std::vector<mg_result> &v = part[g];
                while (cur[g] < v.size() && v[cur[g]].row == q) {
                    mg_result r = v[cur[g]++];
                    r.col += (uint32_t)lo_of[g];
                    out_host[at++] = r;
                }
            }
        return MG_OK;
    }
    return sharded_results(c, q_begin, q_end, false, ref->t[0]->n, out_host, capacity, count_out,
                           [&](int g, uint64_t lo, uint64_t hi, mg_result *o, uint64_t cap, uint64_t *n) {
        return mg_compare_rect_results_host(c->ctxs[(size_t)g], ref->t[(size_t)g], qry->t[(size_t)g], lo, hi, kmer_size, kmer_space,
                                            max_distance, max_p_value, o, cap, n);
    });
}

/* Sketching on every GPU of a local communicator (independent units, no collective; the
 * reference fans its files / records out to its -p threads, Sketch.cpp:211,354, and consumes the
 * results in submission order, ThreadPool.hxx:127-167): the sketches are cut into one block of
 * consecutive sketches per device, balanced by BYTES, one host thread per device runs mg_sketch_host on
 * its block, and every block writes its own rows of the outputs -- input order by construction. */
int mg_sketch_sharded_host(mg_comm *c, const mg_params *p, const uint8_t *bases, uint64_t nbases, const uint64_t *sketch_off,
                           uint64_t nsketch, uint64_t *hashes_out, uint32_t *nhash_out, uint32_t *counts_out)
{
    if (!c || !c->local) return comm_fail(c, MG_ERR_INVALID, "mg_sketch_sharded_host: needs a local communicator");
    if (!p || !sketch_off || !hashes_out || !nhash_out || (!bases && nbases)) return comm_fail(c, MG_ERR_INVALID, "mg_sketch_sharded_host: NULL argument");
    if (nsketch == 0) return MG_OK;
    const size_t G = c->ctxs.size();
    const uint64_t s = p->sketch_size;
    // block boundaries: sketch k goes to the device whose share of the bytes its first byte falls in
    std::vector<uint64_t> b(G + 1, nsketch);
    b[0] = 0;
    const uint64_t total = sketch_off[nsketch] - sketch_off[0];
    for (size_t g = 1; g < G; g++) {
        const uint64_t want = sketch_off[0] + (uint64_t)((unsigned __int128)total * g / G);
        b[g] = (uint64_t)(std::lower_bound(sketch_off, sketch_off + nsketch, want) - sketch_off);
        if (b[g] < b[g - 1]) b[g] = b[g - 1];
    }
    std::vector<int> rcs(G, MG_OK);
    std::vector<std::thread> th;
    auto work = [&](size_t g) {
        const uint64_t k0 = b[g], k1 = b[g + 1];
        if (k0 >= k1) return;
        const uint64_t base = sketch_off[k0];
        std::vector<uint64_t> off(k1 - k0 + 1);
        for (uint64_t k = k0; k <= k1; k++) off[k - k0] = sketch_off[k] - base;
        rcs[g] = mg_sketch_host(c->ctxs[g], p, bases + base, off.back(), off.data(), k1 - k0, hashes_out + k0 * s, nhash_out + k0,
                                counts_out ? counts_out + k0 * s : nullptr);
    };
    for (size_t g = 0; g < G; g++) {
        if (G == 1) work(g); else th.emplace_back(work, g);
    }
    for (auto &t : th) t.join();
    for (size_t g = 0; g < G; g++)
        if (rcs[g] != MG_OK) { c->err = c->ctxs[g]->err; return rcs[g]; }
    return MG_OK;
}

