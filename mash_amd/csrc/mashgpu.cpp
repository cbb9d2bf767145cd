// mashgpu.cpp — the C ABI (include/mashgpu.h) over the gfx950 kernels.
// Host-side orchestration only: work lists, device buffers, launches, error strings.
#include "host_internal.h"

namespace {

thread_local std::string g_create_error;

// contexts that exist: a table freed after its context (interpreter teardown after a failed test) must not touch it
std::mutex g_live_mu;
std::vector<const void *> g_live_ctx;

}  // namespace

bool ctx_is_live(const void *c)
{
    std::lock_guard<std::mutex> lk(g_live_mu);
    return std::find(g_live_ctx.begin(), g_live_ctx.end(), c) != g_live_ctx.end();
}

// Scratch of the latency-sensitive entry points (one genome sketched, a few queries compared)
// comes from a per-context cache: hipMalloc + hipFree cost tens of microseconds each and a call
// makes a dozen of them.  Blocks up to 64 MiB are kept (256 MiB in total) and reused by later
// calls; everything runs on ctx->stream, so a block handed back while work on it is still
// queued is only ever touched again by work queued behind it.
void ctx_trim(mg_ctx *ctx)
{
    for (auto &b : ctx->blk_free) hipFree(b.p);
    ctx->blk_free.clear();
    ctx->blk_cached = 0;
    for (auto &b : ctx->big_free) hipFree(b.p);
    ctx->big_free.clear();
    ctx->big_cached = 0;
}

constexpr size_t CTX_SMALL_BLOCK = (size_t)64 << 20;

hipError_t ctx_malloc(mg_ctx *ctx, void **out, size_t bytes)
{
    bytes = (std::max<size_t>(bytes, 1) + 255) & ~size_t(255);
    if (bytes > CTX_SMALL_BLOCK) {
        // large block: best fit among the blocks handed back, at most a quarter larger than asked for
        size_t pick = SIZE_MAX;
        for (size_t i = 0; i < ctx->big_free.size(); i++) {
            const size_t b = ctx->big_free[i].bytes;
            if (b >= bytes && b <= bytes + bytes / 4 && (pick == SIZE_MAX || b < ctx->big_free[pick].bytes)) pick = i;
        }
        if (pick != SIZE_MAX) {
            *out = ctx->big_free[pick].p;
            ctx->blk_live.push_back(ctx->big_free[pick]);
            ctx->big_cached -= ctx->big_free[pick].bytes;
            ctx->big_free.erase(ctx->big_free.begin() + (long)pick);
            return hipSuccess;
        }
        hipError_t e = hipMalloc(out, bytes);
        if (e != hipSuccess && (!ctx->big_free.empty() || !ctx->blk_free.empty())) {
            (void)hipGetLastError();
            hipStreamSynchronize(ctx->stream);
            ctx_trim(ctx);
            e = hipMalloc(out, bytes);
        }
        if (e == hipSuccess) ctx->blk_live.push_back({*out, bytes});
        return e;
    }
    size_t best = SIZE_MAX;
    for (size_t i = 0; i < ctx->blk_free.size(); i++) {
        const size_t b = ctx->blk_free[i].bytes;
        if (b >= bytes && b <= 2 * bytes + (64u << 10) && (best == SIZE_MAX || b < ctx->blk_free[best].bytes)) best = i;
    }
    if (best != SIZE_MAX) {
        *out = ctx->blk_free[best].p;
        ctx->blk_live.push_back(ctx->blk_free[best]);
        ctx->blk_cached -= ctx->blk_free[best].bytes;
        ctx->blk_free.erase(ctx->blk_free.begin() + (long)best);
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess && (!ctx->blk_free.empty() || !ctx->big_free.empty())) {          // give the caches back and retry
        (void)hipGetLastError();
        hipStreamSynchronize(ctx->stream);
        ctx_trim(ctx);
        e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess) ctx->blk_live.push_back({*out, bytes});
    return e;
}

void ctx_free(mg_ctx *ctx, void *p)
{
    if (!p) return;
    for (size_t i = 0; i < ctx->blk_live.size(); i++) {
        if (ctx->blk_live[i].p != p) continue;
        const mg_ctx::Block b = ctx->blk_live[i];
        ctx->blk_live.erase(ctx->blk_live.begin() + (long)i);
        if (b.bytes <= CTX_SMALL_BLOCK && ctx->blk_cached + b.bytes <= (256u << 20) && ctx->blk_free.size() < 64) {
            ctx->blk_free.push_back(b);
            ctx->blk_cached += b.bytes;
        } else if (b.bytes > CTX_SMALL_BLOCK && ctx->big_cached + b.bytes <= ctx->big_limit && ctx->big_free.size() < 64) {
            ctx->big_free.push_back(b);
            ctx->big_cached += b.bytes;
        } else {
            hipFree(p);
        }
        return;
    }
    hipFree(p);                                                // not ours: plain allocation
}

int fail(mg_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

// a knob: the context's own setting, else the environment's (nullptr: not set)
void *ctx_pinned(mg_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->pin_cap) return ctx->pin;
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    ctx->pin = nullptr;
    ctx->pin_cap = 0;
    const size_t cap = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&ctx->pin, cap, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        ctx->pin = nullptr;
        return nullptr;
    }
    ctx->pin_cap = cap;
    return ctx->pin;
}

const char *ctx_opt(const mg_ctx *ctx, const char *name)
{
    if (ctx) {
        const auto it = ctx->options.find(name);
        if (it != ctx->options.end()) return it->second.c_str();
    }
    return getenv(name);
}

void prof_begin(mg_ctx *ctx, std::vector<ProfRec> &v, hipStream_t stream)
{
    if (!ctx->prof) return;
    ProfRec r;
    hipEventCreate(&r.a);
    hipEventCreate(&r.b);
    hipEventRecord(r.a, stream ? stream : ctx->stream);
    v.push_back(r);
}

void prof_end(mg_ctx *ctx, std::vector<ProfRec> &v, hipStream_t stream)
{
    if (!ctx->prof || v.empty()) return;
    hipEventRecord(v.back().b, stream ? stream : ctx->stream);
}


int mg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int mg_ctx_create(int device, mg_ctx **out)
{
    if (!out) return fail(nullptr, MG_ERR_INVALID, "mg_ctx_create: out is NULL");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, MG_ERR_HIP, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, MG_ERR_INVALID, "mg_ctx_create: bad device index");
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, MG_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    mg_ctx *c = new mg_ctx;
    c->device = device;
    // a BLOCKING stream: implicitly ordered with the legacy default stream, so a caller that prepares
    // inputs or clears outputs on stream 0 (torch's default) needs no event between that and our work
    e = hipStreamCreateWithFlags(&c->stream, hipStreamDefault);
    if (e != hipSuccess) { delete c; return fail(nullptr, MG_ERR_HIP, "hipStreamCreate failed"); }
    c->own_stream = true;
    int cus = 0;                        // (one attribute, not hipGetDeviceProperties: that call fills a page of fields)
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) c->cu_count = cus;
    {
        // the pool of large blocks may hold up to 40 % of the device's memory (the index of an s = 10 000 table of 10^5 rows is
        // 40 GB with its scratch: a pool smaller than a table's blocks frees and allocates them again for every table)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) c->big_limit = std::max<size_t>(c->big_limit, total_b / 10 * 4);
        else (void)hipGetLastError();
    }
    { std::lock_guard<std::mutex> lk(g_live_mu); g_live_ctx.push_back(c); }
    *out = c;
    return MG_OK;
}

void mg_ctx_destroy(mg_ctx *ctx)
{
    if (!ctx) return;
    { std::lock_guard<std::mutex> lk(g_live_mu); g_live_ctx.erase(std::remove(g_live_ctx.begin(), g_live_ctx.end(), (const void *)ctx), g_live_ctx.end()); }
    mg_prof_reset(ctx);
    for (auto &c : ctx->cost_clk) { if (c.a) hipEventDestroy(c.a); if (c.b) hipEventDestroy(c.b); }
    for (auto &sl : ctx->slots) {
        if (sl.dev) hipFree(sl.dev);
        if (sl.host) hipHostFree(sl.host);
        if (sl.done) hipEventDestroy(sl.done);
    }
    ctx_trim(ctx);
    for (auto &b : ctx->blk_live) hipFree(b.p);
    if (ctx->pin) hipHostFree(ctx->pin);
    if (ctx->aux) { hipStreamDestroy(ctx->aux); hipEventDestroy(ctx->aux_go); hipEventDestroy(ctx->aux_done); hipFree(ctx->aux_ctr); ctx->aux = nullptr; }
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *mg_last_error(mg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mg_ctx_trim(mg_ctx *ctx)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx_trim(ctx);
    return MG_OK;
}

int mg_ctx_set_stream(mg_ctx *ctx, void *hip_stream)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (ctx->own_stream && ctx->stream) { hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
    if (hip_stream == nullptr) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamDefault));   // blocking, see mg_ctx_create
        ctx->own_stream = true;
    } else {
        ctx->stream = (hipStream_t)hip_stream;
    }
    return MG_OK;
}

int mg_ctx_set_option(mg_ctx *ctx, const char *name, const char *value)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!name || strncmp(name, "MASHGPU_", 8) != 0) return fail(ctx, MG_ERR_INVALID, "mg_ctx_set_option: knob names start with MASHGPU_");
    if (value) ctx->options[name] = value; else ctx->options.erase(name);
    return MG_OK;
}

int mg_ctx_set_async(mg_ctx *ctx, int on)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    ctx->async = on != 0;
    return MG_OK;
}

int mg_ctx_synchronize(mg_ctx *ctx)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MG_OK;
}

int mg_ctx_cu_count(mg_ctx *ctx) { return ctx ? ctx->cu_count : 0; }

int mg_params_init(mg_params *p, int kmer_size, uint64_t sketch_size, uint32_t seed,
                   const char *alphabet, int noncanonical, int preserve_case)
{
    if (!p || !alphabet || kmer_size < 1 || kmer_size > 32 || sketch_size < 1) return MG_ERR_INVALID;
    memset(p, 0, sizeof *p);
    p->kmer_size = kmer_size;
    p->sketch_size = sketch_size;
    p->seed = seed;
    p->noncanonical = noncanonical ? 1 : 0;
    p->preserve_case = preserve_case ? 1 : 0;
    p->min_copies = 1;
    p->target_cov = 0.0;
    p->bloom_bytes = 0;
    for (const char *c = alphabet; *c; c++) {            // Sketch.cpp:1113-1125
        char u = *c;
        if (!preserve_case && u > 96 && u < 123) u -= 32;
        p->alphabet[(unsigned char)u] = 1;
    }
    for (int i = 0; i < 256; i++) p->alphabet_size += p->alphabet[i] ? 1 : 0;
    p->use64 = pow((double)p->alphabet_size, (double)kmer_size) > pow(2.0, 32.0);   // :1136
    return MG_OK;
}

/* ------------------------------------------------------------------ tables */

int mg_table_upload(mg_ctx *ctx, const uint64_t *hashes, const uint32_t *nhash, const uint64_t *lengths,
                    uint64_t n, uint64_t s, mg_table **out)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!hashes || !nhash || !out || s == 0) return fail(ctx, MG_ERR_INVALID, "mg_table_upload: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf<uint64_t> dh(ctx), dl(ctx);                     // (from the context's pool: the next table of this shape takes the same blocks)
    DevBuf<uint32_t> dn(ctx);
    if (dh.alloc(n * s) != hipSuccess || dn.alloc(n) != hipSuccess || dl.alloc(n) != hipSuccess)
        return fail(ctx, MG_ERR_NOMEM, "mg_table_upload: device allocation failed");
    HIP_TRY(ctx, hipMemcpyAsync(dh, hashes, n * s * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dn, nhash, n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (lengths) HIP_TRY(ctx, hipMemcpyAsync(dl, lengths, n * 8, hipMemcpyHostToDevice, ctx->stream));
    else HIP_TRY(ctx, hipMemsetAsync(dl, 0, n * 8, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    mg_table *t = new mg_table;
    t->ctx = ctx; t->hashes = dh.release(); t->nhash = dn.release(); t->lengths = dl.release(); t->n = n; t->s = s; t->owns = true;
    *out = t;
    return MG_OK;
}

int mg_table_wrap_dev(mg_ctx *ctx, const uint64_t *hashes_dev, const uint32_t *nhash_dev,
                      const uint64_t *lengths_dev, uint64_t n, uint64_t s, mg_table **out)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!hashes_dev || !nhash_dev || !out || s == 0) return fail(ctx, MG_ERR_INVALID, "mg_table_wrap_dev: bad argument");
    mg_table *t = new mg_table;
    t->ctx = ctx; t->hashes = hashes_dev; t->nhash = nhash_dev; t->lengths = lengths_dev;
    t->n = n; t->s = s; t->owns = false;
    *out = t;
    return MG_OK;
}

// everything the compare path derived from the table's contents goes back to the context (large blocks to its
// pool, in stream order: whatever is still queued on them runs before anything queued later reuses them)
static void table_drop_derived(mg_table *t)
{
    mg_ctx *ctx = t->ctx;
    for (mg_table *v : t->prefix_views) { table_drop_derived(v); delete v; }
    t->prefix_views.clear();
    if (!t->pfx.empty() || !t->win.empty() || !t->sparse.empty()) hipSetDevice(ctx->device);
    for (auto &im : t->pfx) ctx_free(ctx, im.second);
    t->pfx.clear();
    for (auto &w : t->win) ctx_free(ctx, w.dev);
    t->win.clear();
    for (mg_table::Sparse *sp : t->sparse) {
        for (auto &pl : sp->plans) {
            if (pl.order) ctx_free(ctx, pl.order);
            if (pl.dtiles) ctx_free(ctx, pl.dtiles);
        }
        for (void *q : {(void *)sp->off, (void *)sp->keys_sorted, (void *)sp->gend, (void *)sp->sorted_rows,
                        (void *)sp->pos_img, (void *)sp->code_img, (void *)sp->short_rows, (void *)sp->short_cnt, (void *)sp->cand,
                        (void *)sp->res, (void *)sp->seg_base, (void *)sp->seg_cnt, (void *)sp->chunks, (void *)sp->chunk_inc,
                        sp->scan_temp, (void *)sp->counters, (void *)sp->rep, (void *)sp->cls_of, (void *)sp->cls_off,
                        (void *)sp->cls_rows, (void *)sp->cls_first, (void *)sp->order, (void *)sp->dgroups, (void *)sp->grp_of,
                        (void *)sp->ulist, (void *)sp->upos, (void *)sp->gdata, (void *)sp->ext, (void *)sp->inv, (void *)sp->phashes})
            if (q) ctx_free(ctx, q);
        for (void *q : sp->jn.bufs)
            if (q) ctx_free(ctx, q);
        delete sp;
    }
    t->sparse.clear();
    t->cls.clear();
    t->last.clear();
    t->nh.clear();
    t->have_max = false;
}

const mg_table *table_prefix_view(const mg_table *t, uint64_t n)
{
    if (n >= t->n) return t;
    for (mg_table *v : t->prefix_views)
        if (v->n == n) return v;
    if (t->prefix_views.size() >= 4) {
        table_drop_derived(t->prefix_views.front());
        delete t->prefix_views.front();
        t->prefix_views.erase(t->prefix_views.begin());
    }
    mg_table *v = new mg_table;
    v->ctx = t->ctx;
    v->hashes = t->hashes;
    v->nhash = t->nhash;
    v->lengths = t->lengths;
    v->n = n;
    v->s = t->s;
    v->owns = false;
    t->prefix_views.push_back(v);
    return v;
}

void mg_table_free(mg_table *t)
{
    if (!t) return;
    if (!ctx_is_live(t->ctx)) { delete t; return; }        // the context is gone (and its device memory with it): only the handle is left
    {
        std::lock_guard<std::recursive_mutex> lk(t->ctx->mu);
        table_drop_derived(t);
        if (t->owns) {
            hipSetDevice(t->ctx->device);
            ctx_free(t->ctx, (void *)t->hashes);
            ctx_free(t->ctx, (void *)t->nhash);
            ctx_free(t->ctx, (void *)t->lengths);
        }
    }
    delete t;
}

int mg_table_invalidate(mg_table *t)
{
    if (!t || !ctx_is_live(t->ctx)) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(t->ctx->mu);
    table_drop_derived(t);
    return MG_OK;
}

uint64_t mg_table_rows(const mg_table *t) { return t ? t->n : 0; }
uint64_t mg_table_sketch_size(const mg_table *t) { return t ? t->s : 0; }

/* ------------------------------------------------------------------ profiling */

int mg_prof_enable(mg_ctx *ctx, int on)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    ctx->prof = on != 0;
    return MG_OK;
}

void mg_prof_reset(mg_ctx *ctx)
{
    if (!ctx) return;
    for (auto *v : {&ctx->prof_compare, &ctx->prof_sketch, &ctx->prof_fill, &ctx->prof_discover, &ctx->prof_merge, &ctx->prof_index, &ctx->prof_dense, &ctx->prof_join, &ctx->prof_fill_aside}) {
        for (auto &r : *v) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
        v->clear();
    }
}

double mg_prof_avg_ms(mg_ctx *ctx, const char *name, uint64_t *launches_out)
{
    if (launches_out) *launches_out = 0;
    if (!ctx || !name) return 0.0;
    std::vector<ProfRec> *v = nullptr;
    if (strcmp(name, "compare") == 0) v = &ctx->prof_compare;
    else if (strcmp(name, "sketch") == 0) v = &ctx->prof_sketch;
    else if (strcmp(name, "compare_fill") == 0) v = &ctx->prof_fill;
    else if (strcmp(name, "compare_fill_aside") == 0) v = &ctx->prof_fill_aside;
    else if (strcmp(name, "compare_discover") == 0) v = &ctx->prof_discover;
    else if (strcmp(name, "compare_merge") == 0) v = &ctx->prof_merge;
    else if (strcmp(name, "compare_index") == 0) v = &ctx->prof_index;
    else if (strcmp(name, "compare_dense") == 0) v = &ctx->prof_dense;
    else if (strcmp(name, "compare_join") == 0) v = &ctx->prof_join;
    if (!v || v->empty()) return 0.0;
    hipStreamSynchronize(ctx->stream);
    double tot = 0.0;
    uint64_t n = 0;
    for (auto &r : *v) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { tot += ms; n++; }
    }
    if (launches_out) *launches_out = n;
    return n ? tot / (double)n : 0.0;
}

