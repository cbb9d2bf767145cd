// mashgpu.cpp — the C ABI (include/mashgpu.h) over the gfx950 kernels.
// Host-side orchestration only: work lists, device buffers, launches, error strings.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <memory>
#include <iterator>
#include <map>
#include <mutex>
#include <queue>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/mashgpu.h"
#include "compare_internal.h"
#include "finish_internal.h"
#include "pvalue.h"
#include "screen_internal.h"
#include "sketch_internal.h"

namespace {

thread_local std::string g_create_error;

// contexts that exist: a table freed after its context (interpreter teardown after a failed test) must not touch it
std::mutex g_live_mu;
std::vector<const void *> g_live_ctx;
bool ctx_is_live(const void *c)
{
    std::lock_guard<std::mutex> lk(g_live_mu);
    return std::find(g_live_ctx.begin(), g_live_ctx.end(), c) != g_live_ctx.end();
}

struct ProfRec { hipEvent_t a, b; };

}  // namespace

struct mg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int cu_count = 0;
    std::string err;
    bool prof = false;
    std::vector<ProfRec> prof_compare, prof_sketch;
    // phases of the inverted-index compare engine (compare_sparse.hip), each its own kernel
    std::vector<ProfRec> prof_fill, prof_discover, prof_merge, prof_index, prof_dense;
    // its fill runs on a stream of its own beside discover + merge (HBM-write bound vs latency bound)
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // Entry points lock the context: any number of host threads may drive one context, one call at
    // a time (SURVEY 8b "thread-safe per ctx"); recursive because entry points call each other.
    std::recursive_mutex mu;
    // mg_ctx_set_async: compare *_dev calls return once their work is queued on `stream`
    bool async = false;
    // tile lists of the compare launches: a ring of {device buffer, pinned staging}; a slot is taken
    // again only after the launches that read it are done (its event), so calls need not end in a
    // stream synchronisation for the list's sake
    struct TileSlot { void *dev = nullptr, *host = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool pending = false; };
    TileSlot slots[4];
    unsigned slot_next = 0;
    // small device blocks handed back by finished calls (ctx_malloc / ctx_free)
    struct Block { void *p; size_t bytes; };
    std::vector<Block> blk_free, blk_live;
    size_t blk_cached = 0;
    // large blocks (the inverted index of a table, candidate lists: hundreds of MB each) handed back by
    // mg_table_free / mg_table_invalidate: a hipMalloc of 3 GB costs milliseconds, the next table of the same
    // shape takes the very same blocks.  Bounded by big_limit; dropped when any allocation fails; mg_ctx_trim.
    std::vector<Block> big_free;
    size_t big_cached = 0, big_limit = (size_t)48 << 30;
    // mg_ctx_set_option: tuning and test knobs of this context (name -> value); a knob that is not set here is looked
    // up in the environment under the same name
    std::map<std::string, std::string> options;
};

struct mg_table {
    mg_ctx *ctx = nullptr;
    const uint64_t *hashes = nullptr;
    const uint32_t *nhash = nullptr;
    const uint64_t *lengths = nullptr;
    uint64_t n = 0, s = 0;
    bool owns = false;
    // lazily built by the compare path (cached across calls; the table is immutable)
    mutable bool have_max = false;
    mutable uint64_t maxval = 0;
    mutable std::vector<std::pair<int, uint32_t *>> pfx;   // u32 prefix images, one per shift in use
    mutable std::vector<uint8_t> cls;     // density class per row (host copy, see table_classes)
    mutable std::vector<uint64_t> last;   // largest hash per row (host copy)
    mutable std::vector<uint32_t> nh;     // hashes per row (host copy)
    // window offsets of the large-sketch compare path (see table_windows), cached per geometry
    struct Windows { int shr; uint32_t delta, nwin, s; uint32_t *dev; std::vector<uint32_t> host; };
    mutable std::vector<Windows> win;
    // inverted index of the compare path's sparse engine (see table_sparse_index), one per sketch size in use
    struct Sparse {
        uint32_t s = 0;                    // sketch size the index covers (the first min(nhash, s) hashes of a row)
        bool usable = false;               // false: outside the engine's reach (reason in `why`), the tile engine is used
        std::string why;
        uint32_t E = 0, G = 0, rs = 0;     // entries, distinct values, row stride of the images
        uint64_t shared = 0;               // sum over values of (copies choose 2): pairs x shared hashes
        uint32_t max_group = 0;            // copies of the most frequent value
        double build_ms = 0;
        uint32_t *off = nullptr;           // [n + 1] compact entry offsets (device)
        std::vector<uint32_t> off_host;
        uint64_t *keys_sorted = nullptr;   // [E] the values in sorted order (rect queries are located in them)
        uint32_t *gend = nullptr;          // [E] at the first sorted position of a value: one past its last
        uint32_t *sorted_rows = nullptr;   // [E] row of every sorted position
        uint32_t *code_img = nullptr;      // [n * rs + 64] 2 x (first sorted position of the entry's value), padding 0xFFFFFFFF
        uint32_t *pos_img = nullptr;       // [n * rs] the entry's own sorted position
        uint32_t *order = nullptr;         // [n] rows in visiting order (see sp_row_key_kernel); nullptr: table order
        // dense groups (compare_dense.hip): runs of consecutive near-identical rows whose inner pairs are bit-mask arithmetic;
        // the index's runs are clipped for their rows, so discovery only sees partners outside a row's group
        std::vector<mg::DenseGroup> dgroups_host;
        mg::DenseGroup *dgroups = nullptr;
        uint32_t *grp_of = nullptr;        // [n] group of a row, 0xFFFFFFFF: none
        uint32_t *ulist = nullptr, *upos = nullptr;        // the groups' universes: values and the positions of their leaders
        unsigned long long *gdata = nullptr, *xm = nullptr; // mask blocks; per row and word three masks of the extras' offsets
        uint16_t *ext = nullptr;
        uint32_t dn_wmax = 0, dn_xs = 0;
        bool dn_lists = false;             // (test knob) every word resolved from the extras' lists instead of their masks
        // The index may be built on the table in ANOTHER ROW ORDER (rows that belong together next to each other, so that
        // they form dense groups whatever the order of the collection; compare_dense.hip: dense_cluster_rows): `clustered`
        // says this variant was asked for, inv != nullptr that the order differs -- index row a is table row inv[a], the
        // index reads the reordered copy `phashes`, and every kernel that writes results maps rows back.  Only the plain
        // full-triangle job uses it (row ranges, rect and list jobs address table rows and take the other variant).
        bool clustered = false;
        uint32_t *inv = nullptr;
        uint64_t *phashes = nullptr;
        // identical rows: rep[row] = first row of its class (nullptr: the table has no copies), classes of >= 2 rows
        uint32_t *rep = nullptr, *cls_of = nullptr, *cls_off = nullptr, *cls_rows = nullptr, *cls_first = nullptr;
        uint32_t cls_members = 0;          // rows in classes of two and more
        uint64_t cls_pairs = 0;            // pairs inside those classes (full triangle)
        uint64_t copies = 0;               // rows that are a copy of an earlier row
        uint64_t runs_dropped = 0;         // entries whose run was a copy of another run of the same row (sp_run_dedupe_kernel)
        uint32_t one_class = 0;            // != 0: EVERY row is a copy of row 0, which has this many hashes (every pair is {c, c})
        bool has_empty = false;            // some row has no hash at all
        uint32_t *short_rows = nullptr, *short_cnt = nullptr;    // rows with fewer than s hashes (ascending) and their counts
        std::vector<uint32_t> short_rows_host;
        // what a (rows, range) job costs, learned by a counting pass the first time it is seen
        struct Plan { const void *rows; uint64_t rb, re; bool triangle; uint64_t cand, shared; bool use; uint32_t *order;
                      mg::DenseTile *dtiles; uint32_t ndtiles, dtile_rows; uint64_t dense_pairs; };
        std::vector<Plan> plans;
        uint2 *cand = nullptr, *res = nullptr;    // candidate list and the candidates' results, grown on demand
        uint64_t cand_cap = 0;
        unsigned long long *counters = nullptr;   // [4] device
        // per row of a launch: its segment of the candidate list, its merge work items (+ scan scratch)
        unsigned long long *seg_base = nullptr;
        uint32_t *seg_cnt = nullptr, *chunks = nullptr, *chunk_inc = nullptr;
        void *scan_temp = nullptr;
        size_t scan_temp_bytes = 0;
        uint64_t seg_rows = 0;
    };
    mutable std::vector<Sparse *> sparse;
};

#define HIP_TRY(ctx, call)                                                           \
    do {                                                                             \
        hipError_t e__ = (call);                                                     \
        if (e__ != hipSuccess) {                                                     \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);        \
            return MG_ERR_HIP;                                                       \
        }                                                                            \
    } while (0)

// Scratch of the latency-sensitive entry points (one genome sketched, a few queries compared)
// comes from a per-context cache: hipMalloc + hipFree cost tens of microseconds each and a call
// makes a dozen of them.  Blocks up to 64 MiB are kept (256 MiB in total) and reused by later
// calls; everything runs on ctx->stream, so a block handed back while work on it is still
// queued is only ever touched again by work queued behind it.
static void ctx_trim(mg_ctx *ctx)
{
    for (auto &b : ctx->blk_free) hipFree(b.p);
    ctx->blk_free.clear();
    ctx->blk_cached = 0;
    for (auto &b : ctx->big_free) hipFree(b.p);
    ctx->big_free.clear();
    ctx->big_cached = 0;
}

constexpr size_t CTX_SMALL_BLOCK = (size_t)64 << 20;

static hipError_t ctx_malloc(mg_ctx *ctx, void **out, size_t bytes)
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

static void ctx_free(mg_ctx *ctx, void *p)
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

// device allocation released on every exit path; with a context it comes from the context's
// block cache, without one hipFree synchronises with the device
template <class T>
struct DevBuf {
    T *p = nullptr;
    mg_ctx *owner = nullptr;
    DevBuf() = default;
    explicit DevBuf(mg_ctx *ctx) : owner(ctx) {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) { if (owner) ctx_free(owner, p); else hipFree(p); } }
    hipError_t alloc(uint64_t count)
    {
        const size_t bytes = std::max<uint64_t>(count, 1) * sizeof(T);
        return owner ? ctx_malloc(owner, reinterpret_cast<void **>(&p), bytes) : hipMalloc(&p, bytes);
    }
    T *release() { T *q = p; p = nullptr; return q; }
    operator T *() const { return p; }
};

static int fail(mg_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

// a knob: the context's own setting, else the environment's (nullptr: not set)
static const char *ctx_opt(const mg_ctx *ctx, const char *name)
{
    if (ctx) {
        const auto it = ctx->options.find(name);
        if (it != ctx->options.end()) return it->second.c_str();
    }
    return getenv(name);
}

static void prof_begin(mg_ctx *ctx, std::vector<ProfRec> &v, hipStream_t stream = nullptr)
{
    if (!ctx->prof) return;
    ProfRec r;
    hipEventCreate(&r.a);
    hipEventCreate(&r.b);
    hipEventRecord(r.a, stream ? stream : ctx->stream);
    v.push_back(r);
}

static void prof_end(mg_ctx *ctx, std::vector<ProfRec> &v, hipStream_t stream = nullptr)
{
    if (!ctx->prof || v.empty()) return;
    hipEventRecord(v.back().b, stream ? stream : ctx->stream);
}

extern "C" {

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
    for (auto &sl : ctx->slots) {
        if (sl.dev) hipFree(sl.dev);
        if (sl.host) hipHostFree(sl.host);
        if (sl.done) hipEventDestroy(sl.done);
    }
    ctx_trim(ctx);
    for (auto &b : ctx->blk_live) hipFree(b.p);
    if (ctx->aux) hipStreamDestroy(ctx->aux);
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
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

/* ------------------------------------------------------------------ sketching */

static bool alphabet_is_dna(const mg_params *p)
{
    if (p->alphabet_size != 4) return false;
    return p->alphabet['A'] && p->alphabet['C'] && p->alphabet['G'] && p->alphabet['T'];
}

// table probe fused into the sketch pass (mash screen)
struct ProbeHook {
    const unsigned long long *keys;
    uint32_t *obs;
    uint64_t mask, key_max;
    uint32_t *touched;                   // see SketchArgs::probe_touched
    unsigned long long *ntouched;
    uint64_t touched_cap;
    uint64_t tier;                       // == key_max: one tier
    const uint32_t *bits;
    uint64_t bits_scale;
};

// Work decomposition of one sketching call: chunks of k-mer start positions (one workgroup each)
// and, for sketches cut into several chunks, the merges that finish them.
struct SketchPlan {
    std::vector<mg::SketchWork> work;
    std::vector<mg::MergeWork> merges;      // [final merges ..., first level of the two-level merges ...]
    size_t nfinal = 0;                      // merges[0, nfinal) write sketches, the rest write pool slots
    uint64_t nslots = 0;                    // pool slots (one per chunk of a multi-chunk sketch)
};

static int plan_sketch_work(mg_ctx *ctx, const mg_params *p, const uint64_t *sketch_off, uint64_t nsketch, uint64_t nbases,
                            int nt, SketchPlan *plan)
{
    const uint64_t k = (uint64_t)p->kmer_size;
    const uint64_t tile = mg::sketch_tile(nt);
    uint64_t total_pos = 0;
    for (uint64_t i = 0; i < nsketch; i++) {
        if (sketch_off[i + 1] < sketch_off[i] || sketch_off[i + 1] > nbases)
            return fail(ctx, MG_ERR_INVALID, "mg_sketch: sketch_off not monotone / out of range");
        const uint64_t len = sketch_off[i + 1] - sketch_off[i];
        if (len >= k) total_pos += len - k + 1;
    }
    uint64_t target_items = 2048;
    if (const char *e = ctx_opt(ctx, "MASHGPU_SKETCH_ITEMS")) target_items = strtoull(e, nullptr, 10);
    if (target_items < 1) target_items = 1;
    uint64_t chunk = (total_pos + target_items - 1) / target_items;
    uint64_t min_chunk = 4 * tile;
    if (const char *e = ctx_opt(ctx, "MASHGPU_SKETCH_MIN_CHUNK")) min_chunk = strtoull(e, nullptr, 10);
    if (chunk < min_chunk) chunk = min_chunk;
    chunk = (chunk + tile - 1) / tile * tile;

    std::vector<mg::MergeWork> level1;
    uint64_t nslots = 0;
    for (uint64_t i = 0; i < nsketch; i++) {
        const uint64_t b = sketch_off[i], e = sketch_off[i + 1];
        const uint64_t len = e - b;
        if (len < k) continue;
        const uint64_t npos = len - k + 1;
        const uint64_t nch = (npos + chunk - 1) / chunk;
        if (nch > 0xFFFFFFFFull || nslots + nch > 0xFFFFFFFFull) return fail(ctx, MG_ERR_INVALID, "mg_sketch: too many chunks");
        if (nch > 1) {
            // many chunks: groups of G slots are merged in parallel into their first slot, then one
            // workgroup merges the group results
            const uint64_t G = 32;
            if (nch > 2 * G) {
                const uint64_t ngroups = (nch + G - 1) / G;
                for (uint64_t g = 0; g < ngroups; g++)
                    level1.push_back(mg::MergeWork{(uint32_t)i, (uint32_t)(nslots + g * G), (uint32_t)std::min(G, nch - g * G), 1, 1});
                plan->merges.push_back(mg::MergeWork{(uint32_t)i, (uint32_t)nslots, (uint32_t)ngroups, (uint32_t)G, 0});
            } else {
                plan->merges.push_back(mg::MergeWork{(uint32_t)i, (uint32_t)nslots, (uint32_t)nch, 1, 0});
            }
        }
        for (uint64_t c = 0; c < nch; c++) {
            mg::SketchWork w;
            w.begin = b + c * chunk;
            w.end = b + std::min(npos, (c + 1) * chunk);
            w.limit = e;
            w.sketch = (uint32_t)i;
            w.slot = nch > 1 ? (uint32_t)(nslots + c) : 0u;
            w.nchunks = (uint32_t)nch;
            w._pad = 0;
            plan->work.push_back(w);
        }
        if (nch > 1) nslots += nch;
    }
    plan->nfinal = plan->merges.size();
    plan->merges.insert(plan->merges.end(), level1.begin(), level1.end());
    plan->nslots = nslots;
    return MG_OK;
}

// What one sketching call holds on the device (released when the call returns) and the launch
// arguments built over it.
struct SketchRun {
    mg_ctx *ctx;
    const mg_params *p;
    int mode = 0, nt = 0;
    uint32_t cap = 0;
    uint64_t s = 0, nsketch = 0;
    SketchPlan plan;
    mg::SketchArgs a;
    DevBuf<mg::SketchWork> d_work;
    DevBuf<mg::MergeWork> d_merge;
    DevBuf<uint8_t> d_alpha;
    DevBuf<uint64_t> d_pool, d_gT, d_seed;
    DevBuf<uint32_t> d_pool_n;
    std::vector<uint64_t> seeds;            // per sketch, ~0 = not seeded (empty: no seeding at all)
    SketchRun(mg_ctx *c, const mg_params *pp)
        : ctx(c), p(pp), d_work(c), d_merge(c), d_alpha(c), d_pool(c), d_gT(c), d_seed(c), d_pool_n(c) {}
};

// merges[0, nfinal) are final, [nfinal, nfinal + nlevel1) first level: the first level runs first
static int launch_merges(SketchRun &r, const mg::MergeWork *d_list, size_t nfinal, size_t nlevel1, uint64_t *hashes_out_dev,
                         uint32_t *nhash_out_dev)
{
    if (nfinal + nlevel1 == 0) return MG_OK;
    mg::MergeArgs m;
    m.pool = r.d_pool;
    m.pool_n = r.d_pool_n;
    m.hashes_out = hashes_out_dev;
    m.nhash_out = nhash_out_dev;
    m.sketch_size = (uint32_t)r.s;
    m.cap = r.cap;
    if (nlevel1) {
        m.work = d_list + nfinal;
        HIP_TRY(r.ctx, mg::launch_merge_chunks(r.nt, m, (uint32_t)nlevel1, r.ctx->stream));
    }
    m.work = d_list;
    HIP_TRY(r.ctx, mg::launch_merge_chunks(r.nt, m, (uint32_t)nfinal, r.ctx->stream));
    return MG_OK;
}

// Seeded thresholds (sketch.hip, SelState::T0): a sketch of L k-mers is started with the threshold
// 3 s/L of the hash range instead of discovering it (the discovery sorts the candidate buffer about
// ten times per chunk: 15 % of a 1 Mbp genome, most of the latency of a small call).  Sketches that
// end with fewer than s hashes below their seed are run again without one (rerun_short_sketches), so
// the result never depends on it.
static int seed_thresholds(SketchRun &r, const uint64_t *sketch_off)
{
    r.a.seed_T = nullptr;
    if (ctx_opt(r.ctx, "MASHGPU_SKETCH_NO_SEED")) return MG_OK;
    const uint64_t k = (uint64_t)r.p->kmer_size;
    const double kmer_space = std::pow((double)std::max<uint32_t>(r.p->alphabet_size, 2), (double)k) / (r.p->noncanonical ? 1.0 : 2.0);
    std::vector<uint64_t> seeds(r.nsketch, ~0ull);
    bool any = false;
    double factor = 3.0;                                            // expected hashes below the seed, in units of s
    if (const char *e = ctx_opt(r.ctx, "MASHGPU_SKETCH_SEED_FACTOR")) factor = std::max(1.0, atof(e));
    for (uint64_t i = 0; i < r.nsketch; i++) {
        const uint64_t len = sketch_off[i + 1] - sketch_off[i];
        if (len < k) continue;
        const double npos = (double)(len - k + 1);
        const double frac = factor * (double)r.s / npos;
        if (frac >= 0.25) continue;                                 // short input: nothing to gain
        if (kmer_space < 64.0 * npos) continue;                     // few possible k-mers: distinct << L, the guess would miss
        seeds[i] = (uint64_t)(frac * (r.p->use64 ? 18446744073709551616.0 : 4294967296.0));
        any = true;
    }
    if (!any) return MG_OK;
    HIP_TRY(r.ctx, r.d_seed.alloc(r.nsketch));
    HIP_TRY(r.ctx, hipMemcpyAsync(r.d_seed, seeds.data(), r.nsketch * 8, hipMemcpyHostToDevice, r.ctx->stream));
    r.a.seed_T = r.d_seed;
    r.seeds.swap(seeds);
    return MG_OK;
}

// Second, unseeded run of the seeded sketches that came out short (also of those that simply have
// fewer than s distinct k-mers: their second run gives the same list).
static int rerun_short_sketches(SketchRun &r, uint64_t *hashes_out_dev, uint32_t *nhash_out_dev)
{
    if (r.seeds.empty()) return MG_OK;
    mg_ctx *ctx = r.ctx;
    std::vector<uint32_t> nh(r.nsketch);
    HIP_TRY(ctx, hipMemcpyAsync(nh.data(), nhash_out_dev, r.nsketch * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint8_t> again(r.nsketch, 0);
    bool any = false;
    for (uint64_t i = 0; i < r.nsketch; i++)
        if (r.seeds[i] != ~0ull && nh[i] < r.s) { again[i] = 1; any = true; }
    if (!any) return MG_OK;
    const SketchPlan &plan = r.plan;
    std::vector<mg::SketchWork> work;
    for (const mg::SketchWork &w : plan.work) if (again[w.sketch]) work.push_back(w);
    std::vector<mg::MergeWork> fin, lvl1;
    for (size_t q = 0; q < plan.merges.size(); q++)
        if (again[plan.merges[q].sketch]) (q < plan.nfinal ? fin : lvl1).push_back(plan.merges[q]);
    uint32_t prev = 0xFFFFFFFFu;
    for (const mg::SketchWork &w : work) {             // a sketch's chunks are consecutive, slots ascending
        if (w.nchunks > 1 && w.sketch != prev) {
            HIP_TRY(ctx, hipMemsetAsync(r.d_pool_n + w.slot, 0, (size_t)w.nchunks * 4, ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(r.d_gT + w.sketch, 0xFF, 8, ctx->stream));
        }
        prev = w.sketch;
    }
    DevBuf<mg::SketchWork> d_work(ctx);
    DevBuf<mg::MergeWork> d_merge(ctx);
    HIP_TRY(ctx, d_work.alloc(work.size()));
    HIP_TRY(ctx, hipMemcpyAsync(d_work, work.data(), work.size() * sizeof(mg::SketchWork), hipMemcpyHostToDevice, ctx->stream));
    mg::SketchArgs a = r.a;
    a.work = d_work;
    a.seed_T = nullptr;
    a.probe_keys = nullptr;                            // every k-mer was already looked up by the first run
    a.probe_obs = nullptr;
    HIP_TRY(ctx, mg::launch_sketch_chunks(r.p->kmer_size, r.mode, r.nt, a, (uint32_t)work.size(), ctx->stream));
    if (!fin.empty()) {
        std::vector<mg::MergeWork> both = fin;
        both.insert(both.end(), lvl1.begin(), lvl1.end());
        HIP_TRY(ctx, d_merge.alloc(both.size()));
        HIP_TRY(ctx, hipMemcpyAsync(d_merge, both.data(), both.size() * sizeof(mg::MergeWork), hipMemcpyHostToDevice, ctx->stream));
        const int rc = launch_merges(r, d_merge, fin.size(), lvl1.size(), hashes_out_dev, nhash_out_dev);
        if (rc != MG_OK) return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the lists above go out of scope
    return MG_OK;
}

// Multiplicities: re-stream every chunk against the finished sketches (count_chunks_kernel).
static int count_multiplicities(SketchRun &r, const uint8_t *bases_dev, const uint64_t *hashes_dev, const uint32_t *nhash_dev,
                                uint32_t *counts_out_dev, uint32_t min_copies)
{
    mg_ctx *ctx = r.ctx;
    const uint64_t s = r.s, nsketch = r.nsketch;
    const std::vector<mg::SketchWork> &work = r.plan.work;
    DevBuf<unsigned long long> d_firstpos(ctx), d_tstar(ctx), d_pos2(ctx);
    DevBuf<uint32_t> d_fix(ctx);
    DevBuf<mg::SketchWork> d_work2(ctx);
    HIP_TRY(ctx, d_firstpos.alloc(nsketch * s));
    HIP_TRY(ctx, hipMemsetAsync(d_firstpos, 0xFF, nsketch * s * 8, ctx->stream));
    HIP_TRY(ctx, d_tstar.alloc(nsketch));
    HIP_TRY(ctx, d_fix.alloc(nsketch));
    mg::CountArgs ca;
    ca.bases = bases_dev;
    ca.work = r.d_work;
    ca.alphabet = r.d_alpha;
    ca.hashes = hashes_dev;
    ca.nhash = nhash_dev;
    ca.counts = counts_out_dev;
    ca.firstpos = d_firstpos;
    ca.tstar = d_tstar;
    ca.sketch_size = (uint32_t)s;
    ca.seed = r.p->seed;
    ca.use64 = r.p->use64;
    ca.fold_case = r.p->preserve_case ? 0 : 1;
    ca.prevpos = nullptr;
    ca.phase = 0;
    HIP_TRY(ctx, mg::launch_count_chunks(r.p->kmer_size, r.mode, ca, (uint32_t)work.size(), ctx->stream));
    // minCov m: a hash is promoted at its m-th occurrence, so t* is the latest m-th occurrence:
    // walk from the first to the m-th position, one pass per step
    unsigned long long *pos_m = d_firstpos;
    if (min_copies > 1) {
        HIP_TRY(ctx, d_pos2.alloc(nsketch * s));
        unsigned long long *cur = d_pos2, *prv = d_firstpos;
        for (uint32_t j = 2; j <= min_copies; j++) {
            HIP_TRY(ctx, hipMemsetAsync(cur, 0xFF, nsketch * s * 8, ctx->stream));
            ca.firstpos = cur;
            ca.prevpos = prv;
            ca.phase = 2;
            HIP_TRY(ctx, mg::launch_count_chunks(r.p->kmer_size, r.mode, ca, (uint32_t)work.size(), ctx->stream));
            std::swap(cur, prv);
        }
        pos_m = prv;
        ca.prevpos = nullptr;
    }
    HIP_TRY(ctx, mg::launch_count_tstar(nhash_dev, counts_out_dev, pos_m, d_tstar, d_fix, (uint32_t)nsketch, (uint32_t)s, ctx->stream));
    std::vector<uint32_t> fix(nsketch);
    HIP_TRY(ctx, hipMemcpyAsync(fix.data(), d_fix, nsketch * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<mg::SketchWork> work2;
    for (const mg::SketchWork &w : work) if (fix[w.sketch]) work2.push_back(w);
    if (!work2.empty()) {
        // the reference stops counting its largest kept hash once the heap is full with it on top
        HIP_TRY(ctx, d_work2.alloc(work2.size()));
        HIP_TRY(ctx, hipMemcpyAsync(d_work2, work2.data(), work2.size() * sizeof(mg::SketchWork), hipMemcpyHostToDevice, ctx->stream));
        ca.work = d_work2;
        ca.phase = 1;
        HIP_TRY(ctx, mg::launch_count_chunks(r.p->kmer_size, r.mode, ca, (uint32_t)work2.size(), ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // work2 goes out of scope
    }
    return MG_OK;
}

static int sketch_dev_impl(mg_ctx *ctx, const mg_params *p, const uint8_t *bases_dev, uint64_t nbases,
                           const uint64_t *sketch_off, uint64_t nsketch, uint64_t *hashes_out_dev,
                           uint32_t *nhash_out_dev, uint32_t *counts_out_dev, const ProbeHook *probe);

int mg_sketch_dev(mg_ctx *ctx, const mg_params *p, const uint8_t *bases_dev, uint64_t nbases,
                  const uint64_t *sketch_off, uint64_t nsketch, uint64_t *hashes_out_dev,
                  uint32_t *nhash_out_dev, uint32_t *counts_out_dev)
{
    return sketch_dev_impl(ctx, p, bases_dev, nbases, sketch_off, nsketch, hashes_out_dev, nhash_out_dev,
                           counts_out_dev, nullptr);
}

// minCov >= 2: bottom-s of the hashes seen at least m times (see range_count_kernel).  One
// sketch at a time; `work` holds the chunks of all sketches, grouped by sketch.
static int sketch_min_copies(mg_ctx *ctx, const mg_params *p, int mode, const uint8_t *bases_dev,
                             const std::vector<mg::SketchWork> &work, const mg::SketchWork *d_work,
                             const uint8_t *d_alpha, uint64_t nsketch, uint64_t *hashes_out_dev,
                             uint32_t *nhash_out_dev)
{
    const uint64_t s = p->sketch_size;
    const uint64_t hash_max = p->use64 ? 0xFFFFFFFFFFFFFFFEull : 0xFFFFFFFFull;
    const uint64_t max_expect = 1ull << 24;                 // distinct hashes aimed at per round (table: 4x slots)
    unsigned long long *d_keys = nullptr, *d_out = nullptr, *d_outn = nullptr;
    uint32_t *d_cnts = nullptr, *d_ovf = nullptr;
    uint64_t slots_cap = 0, out_cap = 0;
    int rc = MG_OK;
    auto release = [&]() {
        hipStreamSynchronize(ctx->stream);
        for (void *q : {(void *)d_keys, (void *)d_out, (void *)d_outn, (void *)d_cnts, (void *)d_ovf})
            if (q) hipFree(q);
    };
    if (hipMalloc(&d_outn, 8) != hipSuccess || hipMalloc(&d_ovf, 4) != hipSuccess) {
        release();
        return fail(ctx, MG_ERR_NOMEM, "mg_sketch: allocation failed");
    }
    size_t w0 = 0;
    while (w0 < work.size() && rc == MG_OK) {
        const uint32_t sk = work[w0].sketch;
        size_t w1 = w0;
        uint64_t npos = 0;
        while (w1 < work.size() && work[w1].sketch == sk) { npos += work[w1].end - work[w1].begin; w1++; }
        std::vector<uint64_t> kept;                          // ascending across rounds
        uint64_t lo = 0;
        // m copies: most distinct hashes of a read set are singletons, plan for 64 s; m = 1: 2 s suffice
        uint64_t expect = std::max<uint64_t>((p->min_copies > 1 ? 64 : 2) * s, 1ull << 16);
        if (const char *e = ctx_opt(ctx, "MASHGPU_MINCOPIES_EXPECT")) expect = std::max<uint64_t>(1024, strtoull(e, nullptr, 10));  // test knob
        bool exhausted = false;
        while (kept.size() < s && !exhausted && rc == MG_OK) {
            // range [lo, hi] expected to hold <= `expect` distinct hashes (there are <= npos k-mers)
            const long double frac = npos <= expect ? 1.0L : (long double)expect / (long double)npos;
            const long double width = frac * ((long double)hash_max + 1.0L);
            uint64_t hi = hash_max;
            if (frac < 1.0L && width < (long double)(hash_max - lo)) hi = lo + (uint64_t)width;
            const uint64_t want = std::min<uint64_t>(expect, npos);
            uint64_t slots = 1024;
            while (slots < 4 * want) slots <<= 1;
            if (slots > slots_cap) {
                if (d_keys) hipFree(d_keys);
                if (d_cnts) hipFree(d_cnts);
                d_keys = nullptr; d_cnts = nullptr;
                if (hipMalloc(&d_keys, slots * 8) != hipSuccess || hipMalloc(&d_cnts, slots * 4) != hipSuccess) {
                    rc = fail(ctx, MG_ERR_NOMEM, "mg_sketch: allocation failed (min_copies table)");
                    break;
                }
                slots_cap = slots;
            }
            if (slots / 2 > out_cap) {
                if (d_out) hipFree(d_out);
                d_out = nullptr;
                if (hipMalloc(&d_out, slots / 2 * 8) != hipSuccess) {
                    rc = fail(ctx, MG_ERR_NOMEM, "mg_sketch: allocation failed (min_copies list)");
                    break;
                }
                out_cap = slots / 2;
            }
            mg::RangeCountArgs ra;
            ra.bases = bases_dev;
            ra.work = d_work + w0;
            ra.alphabet = d_alpha;
            ra.keys = d_keys;
            ra.cnts = d_cnts;
            ra.overflow = d_ovf;
            ra.mask = slots - 1;
            ra.lo = lo; ra.hi = hi;
            ra.seed = p->seed;
            ra.use64 = p->use64;
            ra.fold_case = p->preserve_case ? 0 : 1;
            unsigned long long n_out = 0;
            uint32_t ovf = 0;
            hipError_t e = hipMemsetAsync(d_keys, 0xFF, slots * 8, ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(d_cnts, 0, slots * 4, ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(d_ovf, 0, 4, ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(d_outn, 0, 8, ctx->stream);
            if (e == hipSuccess) e = mg::launch_range_count(p->kmer_size, mode, ra, (uint32_t)(w1 - w0), ctx->stream);
            if (e == hipSuccess) e = mg::launch_range_extract(d_keys, d_cnts, slots, p->min_copies > 1 ? p->min_copies : 1, d_out, d_outn, out_cap, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&n_out, d_outn, 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&ovf, d_ovf, 4, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { rc = fail(ctx, MG_ERR_HIP, std::string("mg_sketch (min_copies): ") + hipGetErrorString(e)); break; }
            if (ovf || n_out > out_cap) {                    // more distinct hashes than planned: narrow the range
                if (expect <= 1024) { rc = fail(ctx, MG_ERR_HIP, "mg_sketch (min_copies): counting table overflow"); break; }
                expect /= 4;
                continue;
            }
            std::vector<uint64_t> got(n_out);
            if (n_out && hipMemcpy(got.data(), d_out, n_out * 8, hipMemcpyDeviceToHost) != hipSuccess) {
                rc = fail(ctx, MG_ERR_HIP, "mg_sketch (min_copies): D2H copy failed");
                break;
            }
            std::sort(got.begin(), got.end());
            for (uint64_t v : got) { if (kept.size() < s) kept.push_back(v); }
            if (hi >= hash_max) exhausted = true;
            else lo = hi + 1;
            if (expect < max_expect) expect *= 8;
        }
        if (rc != MG_OK) break;
        const uint32_t n = (uint32_t)kept.size();
        if (n && hipMemcpy(hashes_out_dev + (uint64_t)sk * s, kept.data(), (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(ctx, MG_ERR_HIP, "mg_sketch (min_copies): H2D copy failed");
        if (rc == MG_OK && hipMemcpy(nhash_out_dev + sk, &n, 4, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(ctx, MG_ERR_HIP, "mg_sketch (min_copies): H2D copy failed");
        w0 = w1;
    }
    (void)nsketch;
    release();
    return rc;
}

static int sketch_dev_impl(mg_ctx *ctx, const mg_params *p, const uint8_t *bases_dev, uint64_t nbases,
                           const uint64_t *sketch_off, uint64_t nsketch, uint64_t *hashes_out_dev,
                           uint32_t *nhash_out_dev, uint32_t *counts_out_dev, const ProbeHook *probe)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!p || !sketch_off || !hashes_out_dev || !nhash_out_dev || (!bases_dev && nbases))
        return fail(ctx, MG_ERR_INVALID, "mg_sketch: NULL argument");
    if (p->kmer_size < 1 || p->kmer_size > 32) return fail(ctx, MG_ERR_INVALID, "mg_sketch: k must be 1..32");
    if (counts_out_dev && !mg::count_supported(p->sketch_size))
        return fail(ctx, MG_ERR_UNSUPPORTED, "mg_sketch: sketch size too large for the multiplicity pass");
    if (p->target_cov > 0) return fail(ctx, MG_ERR_UNSUPPORTED, "mg_sketch: target_cov needs mg_sketch_reads_host");
    if (p->bloom_bytes) return fail(ctx, MG_ERR_UNSUPPORTED, "mg_sketch: bloom_bytes needs mg_sketch_reads_host");
    if (nsketch == 0) return MG_OK;
    if (nsketch > 0xFFFFFFFFull) return fail(ctx, MG_ERR_INVALID, "mg_sketch: too many sketches");
    if (((uintptr_t)bases_dev & 15) != 0) return fail(ctx, MG_ERR_INVALID, "mg_sketch: bases must be 16-byte aligned");
    const bool dna = alphabet_is_dna(p);
    if (!p->noncanonical && !dna)
        return fail(ctx, MG_ERR_UNSUPPORTED, "mg_sketch: canonical k-mers need the ACGT alphabet");
    int nt = 0;
    uint32_t cap = 0;
    // sketch sizes beyond the LDS selector (s > 12288) take the exact range-counting path that
    // also serves min_copies > 1: bottom-s distinct hashes via an open-addressing table in HBM
    const bool lds_selector = mg::sketch_geometry(p->sketch_size, &nt, &cap);
    if (!lds_selector) {
        if (probe) return fail(ctx, MG_ERR_UNSUPPORTED, "mg_screen: sketch size too large (max 12288)");
        nt = 256;                                          // chunk geometry only
        cap = 0;
    }
    const int mode = dna ? (p->noncanonical ? 1 : 0) : 2;
    const uint64_t s = p->sketch_size;
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    SketchRun run(ctx, p);
    run.mode = mode; run.nt = nt; run.cap = cap; run.s = s; run.nsketch = nsketch;
    int rc = plan_sketch_work(ctx, p, sketch_off, nsketch, nbases, nt, &run.plan);
    if (rc != MG_OK) return rc;
    const SketchPlan &plan = run.plan;

    // outputs default to "empty sketch"
    HIP_TRY(ctx, hipMemsetAsync(hashes_out_dev, 0xFF, nsketch * s * 8, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(nhash_out_dev, 0, nsketch * 4, ctx->stream));
    if (counts_out_dev) HIP_TRY(ctx, hipMemsetAsync(counts_out_dev, 0, nsketch * s * 4, ctx->stream));
    if (plan.work.empty()) {                               // (nothing long enough to hold a k-mer: empty sketches, complete on return)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return MG_OK;
    }

    HIP_TRY(ctx, run.d_work.alloc(plan.work.size()));
    HIP_TRY(ctx, hipMemcpyAsync(run.d_work, plan.work.data(), plan.work.size() * sizeof(mg::SketchWork), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, run.d_alpha.alloc(256));
    HIP_TRY(ctx, hipMemcpyAsync(run.d_alpha, p->alphabet, 256, hipMemcpyHostToDevice, ctx->stream));
    const uint32_t min_copies = p->min_copies > 1 ? p->min_copies : 1;
    const bool range_path = min_copies > 1 || !lds_selector;
    if (plan.nslots && !range_path) {
        HIP_TRY(ctx, run.d_pool.alloc(plan.nslots * s));
        HIP_TRY(ctx, run.d_pool_n.alloc(plan.nslots));
        HIP_TRY(ctx, hipMemsetAsync(run.d_pool_n, 0, plan.nslots * 4, ctx->stream));
        HIP_TRY(ctx, run.d_gT.alloc(nsketch));
        HIP_TRY(ctx, hipMemsetAsync(run.d_gT, 0xFF, nsketch * 8, ctx->stream));
        HIP_TRY(ctx, run.d_merge.alloc(plan.merges.size()));
        HIP_TRY(ctx, hipMemcpyAsync(run.d_merge, plan.merges.data(), plan.merges.size() * sizeof(mg::MergeWork), hipMemcpyHostToDevice, ctx->stream));
    }
    mg::SketchArgs &a = run.a;
    a.bases = bases_dev;
    a.work = run.d_work;
    a.alphabet = run.d_alpha;
    a.hashes_out = hashes_out_dev;
    a.nhash_out = nhash_out_dev;
    a.pool = run.d_pool;
    a.pool_n = run.d_pool_n;
    a.g_T = run.d_gT;
    a.sketch_size = (uint32_t)s;
    a.cap = cap;
    a.seed = p->seed;
    a.use64 = p->use64;
    a.fold_case = p->preserve_case ? 0 : 1;
    a.probe_keys = probe ? probe->keys : nullptr;
    a.probe_obs = probe ? probe->obs : nullptr;
    a.probe_mask = probe ? probe->mask : 0;
    a.probe_max = probe ? probe->key_max : 0;
    a.probe_touched = probe ? probe->touched : nullptr;
    a.probe_ntouched = probe ? probe->ntouched : nullptr;
    a.probe_touched_cap = probe ? probe->touched_cap : 0;
    a.probe_tier = probe ? probe->tier : 0;
    a.probe_bits = probe ? probe->bits : nullptr;
    a.probe_bits_scale = probe ? probe->bits_scale : 0;
    a.seed_T = nullptr;
    if (range_path) {
        // -m / s beyond the LDS selector: bottom-s of the hashes seen at least m times, by exact range counting
        if (probe) return fail(ctx, MG_ERR_UNSUPPORTED, "mg_screen: min_copies does not apply");
        rc = sketch_min_copies(ctx, p, mode, bases_dev, plan.work, run.d_work, run.d_alpha, nsketch, hashes_out_dev, nhash_out_dev);
        if (rc != MG_OK) return rc;
    } else {
        rc = seed_thresholds(run, sketch_off);
        if (rc != MG_OK) return rc;
        prof_begin(ctx, ctx->prof_sketch);
        HIP_TRY(ctx, mg::launch_sketch_chunks(p->kmer_size, mode, nt, a, (uint32_t)plan.work.size(), ctx->stream));
        prof_end(ctx, ctx->prof_sketch);
        rc = launch_merges(run, run.d_merge, plan.nfinal, plan.merges.size() - plan.nfinal, hashes_out_dev, nhash_out_dev);
        if (rc == MG_OK) rc = rerun_short_sketches(run, hashes_out_dev, nhash_out_dev);
        if (rc != MG_OK) return rc;
    }
    if (counts_out_dev) {
        rc = count_multiplicities(run, bases_dev, hashes_out_dev, nhash_out_dev, counts_out_dev, min_copies);
        if (rc != MG_OK) return rc;
    }
    // the call is synchronous: results are complete, and the work lists may go
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MG_OK;
}

int mg_sketch_host(mg_ctx *ctx, const mg_params *p, const uint8_t *bases, uint64_t nbases,
                   const uint64_t *sketch_off, uint64_t nsketch, uint64_t *hashes_out,
                   uint32_t *nhash_out, uint32_t *counts_out)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!p || !hashes_out || !nhash_out) return fail(ctx, MG_ERR_INVALID, "mg_sketch_host: NULL argument");
    if (nsketch == 0) return MG_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t s = p->sketch_size;
    DevBuf<uint8_t> d_bases(ctx);
    DevBuf<uint64_t> d_hashes(ctx);
    DevBuf<uint32_t> d_nhash(ctx), d_counts(ctx);
    if ((counts_out && d_counts.alloc(nsketch * s) != hipSuccess) || d_bases.alloc(nbases + 64) != hipSuccess ||
        d_hashes.alloc(nsketch * s) != hipSuccess || d_nhash.alloc(nsketch) != hipSuccess)
        return fail(ctx, MG_ERR_NOMEM, "mg_sketch_host: device allocation failed");
    if (hipMemcpyAsync(d_bases, bases, nbases, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return fail(ctx, MG_ERR_HIP, "mg_sketch_host: H2D copy failed");
    const int rc = mg_sketch_dev(ctx, p, d_bases, nbases, sketch_off, nsketch, d_hashes, d_nhash, d_counts);
    if (rc != MG_OK) return rc;
    if ((counts_out && hipMemcpyAsync(counts_out, d_counts, nsketch * s * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) ||
        hipMemcpyAsync(hashes_out, d_hashes, nsketch * s * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(nhash_out, d_nhash, nsketch * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return fail(ctx, MG_ERR_HIP, "mg_sketch_host: D2H copy failed");
    return MG_OK;
}

/* ------------------------------------------------- packed nucleotide input (ingest.hip, pack_bases.cpp) */

// One implementation behind mg_sketch_host_packed / mg_sketch_dev_packed: the sketches are taken in pieces of whole
// sketches (about 2^28 bases each); a piece's packed range is turned back into the bytes of the ASCII path
// (launch_unpack_bases) and handed to the ordinary sketch path.  Host input: the NEXT piece crosses PCIe on a
// stream of its own (a helper thread issues and awaits the copy) while this one is sketched.
static int sketch_packed_impl(mg_ctx *ctx, const mg_params *p, const uint8_t *packed, const uint8_t *mask, bool host_input,
                              uint64_t nbases, const uint64_t *sketch_off, uint64_t nsketch, uint64_t *d_hashes, uint32_t *d_nhash,
                              uint32_t *d_counts)
{
    const uint64_t s = p->sketch_size;
    uint64_t cap = 1ull << 28;
    if (const char *e = ctx_opt(ctx, "MASHGPU_PACKED_PIECE")) cap = std::max<uint64_t>(strtoull(e, nullptr, 10), 1);      // (test knob)
    for (uint64_t i = 0; i < nsketch; i++)
        if (sketch_off[i] > sketch_off[i + 1] || sketch_off[i + 1] > nbases) return fail(ctx, MG_ERR_INVALID, "mg_sketch_packed: sketch_off must ascend and end within nbases");
    struct Piece { uint64_t i0, i1, b0, b1; };
    std::vector<Piece> pieces;
    uint64_t longest = 0;
    for (uint64_t i0 = 0; i0 < nsketch;) {
        uint64_t i1 = i0 + 1;
        while (i1 < nsketch && sketch_off[i1 + 1] - sketch_off[i0] <= cap && i1 - i0 < (1ull << 24)) i1++;
        pieces.push_back({i0, i1, sketch_off[i0], sketch_off[i1]});
        longest = std::max(longest, sketch_off[i1] - sketch_off[i0]);
        i0 = i1;
    }
    DevBuf<uint8_t> d_ascii(ctx), d_pk[2] = {DevBuf<uint8_t>(ctx), DevBuf<uint8_t>(ctx)}, d_mk[2] = {DevBuf<uint8_t>(ctx), DevBuf<uint8_t>(ctx)};
    if (d_ascii.alloc(((longest + 15u) & ~15ull) + 64u) != hipSuccess) return fail(ctx, MG_ERR_NOMEM, "mg_sketch_packed: device allocation failed");
    hipStream_t copy_stream = nullptr;
    struct StreamGuard { hipStream_t *s; ~StreamGuard() { if (*s) hipStreamDestroy(*s); } } stream_guard{&copy_stream};
    if (host_input) {
        for (int k = 0; k < 2; k++)
            if (d_pk[k].alloc(longest / 4u + 32u) != hipSuccess || (mask && d_mk[k].alloc(longest / 8u + 32u) != hipSuccess))
                return fail(ctx, MG_ERR_NOMEM, "mg_sketch_packed: device allocation failed");
        HIP_TRY(ctx, hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    }
    // the piece's ranges in the two arrays, cut at 4-byte boundaries (the kernel's loads are dwords)
    auto pk_byte0 = [](const Piece &q) { return (q.b0 / 4u) & ~3ull; };
    auto mk_byte0 = [](const Piece &q) { return (q.b0 / 8u) & ~3ull; };
    const int device = ctx->device;
    auto copy_piece = [&, device](const Piece &q, int slot, hipError_t *err) {
        *err = hipSetDevice(device);
        const uint64_t pb0 = pk_byte0(q), pb1 = (q.b1 + 3u) / 4u, mb0 = mk_byte0(q), mb1 = (q.b1 + 7u) / 8u;
        if (*err == hipSuccess && pb1 > pb0) *err = hipMemcpyAsync(d_pk[slot], packed + pb0, pb1 - pb0, hipMemcpyHostToDevice, copy_stream);
        if (*err == hipSuccess && mask && mb1 > mb0) *err = hipMemcpyAsync(d_mk[slot], mask + mb0, mb1 - mb0, hipMemcpyHostToDevice, copy_stream);
        if (*err == hipSuccess) *err = hipStreamSynchronize(copy_stream);
    };
    hipError_t copy_err = hipSuccess;
    if (host_input && !pieces.empty()) copy_piece(pieces[0], 0, &copy_err);
    std::vector<uint64_t> off;
    for (size_t c = 0; c < pieces.size(); c++) {
        const Piece &q = pieces[c];
        if (copy_err != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("mg_sketch_packed: H2D copy failed: ") + hipGetErrorString(copy_err));
        std::thread next;
        hipError_t next_err = hipSuccess;
        if (host_input && c + 1 < pieces.size()) next = std::thread(copy_piece, std::cref(pieces[c + 1]), (int)((c + 1) & 1), &next_err);
        struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join{next};
        const uint64_t len = q.b1 - q.b0;
        const uint8_t *src_pk, *src_mk;
        uint32_t skip, mskip;
        if (host_input) {
            src_pk = d_pk[c & 1];
            src_mk = mask ? d_mk[c & 1].p : nullptr;
            skip = (uint32_t)(q.b0 - pk_byte0(q) * 4u);
            mskip = (uint32_t)(q.b0 - mk_byte0(q) * 8u);
        } else {
            src_pk = packed + pk_byte0(q);
            src_mk = mask ? mask + mk_byte0(q) : nullptr;
            skip = (uint32_t)(q.b0 - pk_byte0(q) * 4u);
            mskip = (uint32_t)(q.b0 - mk_byte0(q) * 8u);
        }
        HIP_TRY(ctx, mg::launch_unpack_bases(src_pk, src_mk, skip, mskip, len, d_ascii, ctx->stream));
        off.resize(q.i1 - q.i0 + 1);
        for (uint64_t i = q.i0; i <= q.i1; i++) off[i - q.i0] = sketch_off[i] - q.b0;
        const int rc = sketch_dev_impl(ctx, p, d_ascii, len, off.data(), q.i1 - q.i0, d_hashes + q.i0 * s, d_nhash + q.i0,
                                       d_counts ? d_counts + q.i0 * s : nullptr, nullptr);
        if (rc != MG_OK) return rc;
        if (next.joinable()) next.join();
        copy_err = next_err;
    }
    return MG_OK;
}

static int sketch_packed_check(mg_ctx *ctx, const mg_params *p, const void *packed, uint64_t nbases, const uint64_t *sketch_off,
                               const void *hashes_out, const void *nhash_out)
{
    if (!p || !sketch_off || !hashes_out || !nhash_out || (!packed && nbases)) return fail(ctx, MG_ERR_INVALID, "mg_sketch_packed: NULL argument");
    if (!alphabet_is_dna(p)) return fail(ctx, MG_ERR_UNSUPPORTED, "mg_sketch_packed: packed input is defined for the ACGT alphabet only");
    return MG_OK;
}

int mg_sketch_dev_packed(mg_ctx *ctx, const mg_params *p, const uint8_t *packed_dev, const uint8_t *invalid_mask_dev, uint64_t nbases,
                         const uint64_t *sketch_off_host, uint64_t nsketch, uint64_t *hashes_out_dev, uint32_t *nhash_out_dev,
                         uint32_t *counts_out_dev)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    int rc = sketch_packed_check(ctx, p, packed_dev, nbases, sketch_off_host, hashes_out_dev, nhash_out_dev);
    if (rc != MG_OK || nsketch == 0) return rc;
    if (((uintptr_t)packed_dev & 15u) || ((uintptr_t)invalid_mask_dev & 15u)) return fail(ctx, MG_ERR_INVALID, "mg_sketch_dev_packed: arrays must be 16-byte aligned");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return sketch_packed_impl(ctx, p, packed_dev, invalid_mask_dev, false, nbases, sketch_off_host, nsketch, hashes_out_dev, nhash_out_dev, counts_out_dev);
}

int mg_sketch_host_packed(mg_ctx *ctx, const mg_params *p, const uint8_t *packed, const uint8_t *invalid_mask, uint64_t nbases,
                          const uint64_t *sketch_off, uint64_t nsketch, uint64_t *hashes_out, uint32_t *nhash_out, uint32_t *counts_out)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    int rc = sketch_packed_check(ctx, p, packed, nbases, sketch_off, hashes_out, nhash_out);
    if (rc != MG_OK || nsketch == 0) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t s = p->sketch_size;
    DevBuf<uint64_t> d_hashes(ctx);
    DevBuf<uint32_t> d_nhash(ctx), d_counts(ctx);
    if ((counts_out && d_counts.alloc(nsketch * s) != hipSuccess) || d_hashes.alloc(nsketch * s) != hipSuccess || d_nhash.alloc(nsketch) != hipSuccess)
        return fail(ctx, MG_ERR_NOMEM, "mg_sketch_host_packed: device allocation failed");
    rc = sketch_packed_impl(ctx, p, packed, invalid_mask, true, nbases, sketch_off, nsketch, d_hashes, d_nhash, d_counts);
    if (rc != MG_OK) return rc;
    if ((counts_out && hipMemcpyAsync(counts_out, d_counts, nsketch * s * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) ||
        hipMemcpyAsync(hashes_out, d_hashes, nsketch * s * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(nhash_out, d_nhash, nsketch * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return fail(ctx, MG_ERR_HIP, "mg_sketch_host_packed: D2H copy failed");
    return MG_OK;
}

/* ------------------------------------------------- streamed ingest: segments in, sketches out */

// mg_sketch_host wants the whole batch as ONE host array: a caller that parses files has to
// concatenate them first (600 MB of memcpy for 12 000 small genomes) and the pageable H2D copy
// then runs while nothing else does.  A session instead takes the bytes as they are parsed:
// they are packed into a ring of two pinned staging buffers and leave for the device on a copy
// stream while the caller parses on; sketch boundaries are marked as they occur; mg_sketch_finish
// runs the kernels over what has arrived and hands the sketches back.  (The reference overlaps
// parsing and sketching the same way through its ThreadPool, ThreadPool.hxx:127-167.)
struct mg_sketch_session {
    mg_ctx *ctx = nullptr;
    mg_params p;
    uint8_t *d_bases = nullptr;
    uint64_t d_cap = 0, d_used = 0;
    uint8_t *stage[2] = {nullptr, nullptr};
    uint64_t stage_cap = 32ull << 20, fill = 0;
    int cur = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool ev_pending[2] = {false, false};
    hipStream_t copy_stream = nullptr;
    uint64_t window = 0;                      // bytes lent by mg_sketch_stage and not yet committed
    std::vector<uint64_t> off{0};
};

static int session_submit(mg_sketch_session *ss)
{
    mg_ctx *ctx = ss->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (ss->fill == 0) return MG_OK;
    if (ss->d_used + ss->fill + 64 > ss->d_cap) {
        // grow the device arena (copies what has arrived; rare: capacity doubles)
        uint64_t cap = std::max<uint64_t>(ss->d_cap * 2, 256ull << 20);
        while (cap < ss->d_used + ss->fill + 64) cap *= 2;
        uint8_t *nb = nullptr;
        if (hipMalloc(&nb, cap) != hipSuccess) return fail(ctx, MG_ERR_NOMEM, "mg_sketch_add: device allocation failed");
        if (ss->d_used)
            HIP_TRY(ctx, hipMemcpyAsync(nb, ss->d_bases, ss->d_used, hipMemcpyDeviceToDevice, ss->copy_stream));
        HIP_TRY(ctx, hipStreamSynchronize(ss->copy_stream));
        if (ss->d_bases) hipFree(ss->d_bases);
        ss->d_bases = nb;
        ss->d_cap = cap;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ss->d_bases + ss->d_used, ss->stage[ss->cur], ss->fill, hipMemcpyHostToDevice, ss->copy_stream));
    HIP_TRY(ctx, hipEventRecord(ss->ev[ss->cur], ss->copy_stream));
    ss->ev_pending[ss->cur] = true;
    ss->d_used += ss->fill;
    ss->fill = 0;
    ss->cur ^= 1;
    if (ss->ev_pending[ss->cur]) {                          // the other buffer's copy must have left before it is refilled
        HIP_TRY(ctx, hipEventSynchronize(ss->ev[ss->cur]));
        ss->ev_pending[ss->cur] = false;
    }
    return MG_OK;
}

int mg_sketch_begin(mg_ctx *ctx, const mg_params *p, mg_sketch_session **out)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!p || !out) return fail(ctx, MG_ERR_INVALID, "mg_sketch_begin: NULL argument");
    if (p->target_cov > 0) return fail(ctx, MG_ERR_UNSUPPORTED, "mg_sketch_begin: target_cov needs mg_sketch_reads_host");
    if (p->bloom_bytes) return fail(ctx, MG_ERR_UNSUPPORTED, "mg_sketch_begin: bloom_bytes needs mg_sketch_reads_host");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mg_sketch_session *ss = new mg_sketch_session;
    ss->ctx = ctx;
    ss->p = *p;
    if (const char *e = ctx_opt(ctx, "MASHGPU_STAGE_BYTES")) ss->stage_cap = std::max<uint64_t>(64, strtoull(e, nullptr, 10));   // test knob
    hipError_t e = hipStreamCreateWithFlags(&ss->copy_stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipHostMalloc((void **)&ss->stage[i], ss->stage_cap, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ss->ev[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        mg_sketch_session_free(ss);
        return fail(ctx, MG_ERR_HIP, std::string("mg_sketch_begin: ") + hipGetErrorString(e));
    }
    *out = ss;
    return MG_OK;
}

int mg_sketch_add(mg_sketch_session *ss, const uint8_t *bytes, uint64_t len)
{
    if (!ss) return MG_ERR_INVALID;
    if (!bytes && len) return fail(ss->ctx, MG_ERR_INVALID, "mg_sketch_add: NULL bytes");
    while (len) {
        const uint64_t n = std::min(len, ss->stage_cap - ss->fill);
        memcpy(ss->stage[ss->cur] + ss->fill, bytes, n);
        ss->fill += n;
        bytes += n;
        len -= n;
        if (ss->fill == ss->stage_cap) {
            const int rc = session_submit(ss);
            if (rc != MG_OK) return rc;
        }
    }
    return MG_OK;
}

uint64_t mg_sketch_stage_capacity(const mg_sketch_session *ss) { return ss ? ss->stage_cap : 0; }

int mg_sketch_stage(mg_sketch_session *ss, uint64_t len, uint8_t **window)
{
    if (!ss || !window) return MG_ERR_INVALID;
    if (len > ss->stage_cap) return fail(ss->ctx, MG_ERR_INVALID, "mg_sketch_stage: window larger than the staging buffer (use mg_sketch_add)");
    if (ss->fill + len > ss->stage_cap) {
        const int rc = session_submit(ss);
        if (rc != MG_OK) return rc;
    }
    ss->window = len;
    *window = ss->stage[ss->cur] + ss->fill;
    return MG_OK;
}

int mg_sketch_commit(mg_sketch_session *ss, uint64_t len)
{
    if (!ss) return MG_ERR_INVALID;
    if (len > ss->window) return fail(ss->ctx, MG_ERR_INVALID, "mg_sketch_commit: more bytes than the window that was lent");
    ss->window -= len;
    ss->fill += len;
    return ss->fill == ss->stage_cap ? session_submit(ss) : MG_OK;
}

int mg_sketch_end_sketch(mg_sketch_session *ss)
{
    if (!ss) return MG_ERR_INVALID;
    ss->off.push_back(ss->d_used + ss->fill);
    return MG_OK;
}

uint64_t mg_sketch_pending(const mg_sketch_session *ss) { return ss ? ss->off.size() - 1 : 0; }

int mg_sketch_finish(mg_sketch_session *ss, uint64_t *hashes_out, uint32_t *nhash_out, uint32_t *counts_out)
{
    if (!ss) return MG_ERR_INVALID;
    mg_ctx *ctx = ss->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    const uint64_t nsketch = ss->off.size() - 1;
    int rc = MG_OK;
    if (nsketch) {
        if (!hashes_out || !nhash_out) return fail(ctx, MG_ERR_INVALID, "mg_sketch_finish: NULL argument");
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        if (ss->fill == 0 && ss->d_used == 0) { uint8_t sep = MG_RECORD_SEP; rc = mg_sketch_add(ss, &sep, 1); }   // all sketches empty
        if (rc == MG_OK) rc = session_submit(ss);
        if (rc != MG_OK) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ss->copy_stream));
        ss->ev_pending[0] = ss->ev_pending[1] = false;
        const uint64_t s = ss->p.sketch_size;
        DevBuf<uint64_t> d_hashes(ctx);
        DevBuf<uint32_t> d_nhash(ctx), d_counts(ctx);
        if ((counts_out && d_counts.alloc(nsketch * s) != hipSuccess) || d_hashes.alloc(nsketch * s) != hipSuccess ||
            d_nhash.alloc(nsketch) != hipSuccess)
            return fail(ctx, MG_ERR_NOMEM, "mg_sketch_finish: device allocation failed");
        rc = mg_sketch_dev(ctx, &ss->p, ss->d_bases, ss->d_used, ss->off.data(), nsketch, d_hashes, d_nhash, d_counts);
        if (rc != MG_OK) return rc;
        if ((counts_out && hipMemcpyAsync(counts_out, d_counts, nsketch * s * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) ||
            hipMemcpyAsync(hashes_out, d_hashes, nsketch * s * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(nhash_out, d_nhash, nsketch * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            return fail(ctx, MG_ERR_HIP, "mg_sketch_finish: D2H copy failed");
    }
    ss->d_used = 0;                                       // the arena and the staging ring are kept for the next batch
    ss->fill = 0;
    ss->off.assign(1, 0);
    return MG_OK;
}

void mg_sketch_session_free(mg_sketch_session *ss)
{
    if (!ss) return;
    hipSetDevice(ss->ctx->device);
    if (ss->copy_stream) { hipStreamSynchronize(ss->copy_stream); hipStreamDestroy(ss->copy_stream); }
    for (int i = 0; i < 2; i++) {
        if (ss->stage[i]) hipHostFree(ss->stage[i]);
        if (ss->ev[i]) hipEventDestroy(ss->ev[i]);
    }
    if (ss->d_bases) hipFree(ss->d_bases);
    delete ss;
}

/* ------------------------------------------------- reads mode with early stop (-c) */

namespace {

// MinHashHeap::tryInsert (MinHashHeap.cpp:68-145) over explicit containers: kept hashes with
// counts, pending hashes (multiplicityMinimum > 1) and the pending max-queue that may hold hashes
// already erased from the pending set; with -b, the Bloom filter in front of the kept set.
//
// The filter (MinHashHeap.cpp:19-41: vendored bloom_filter.hpp with projected_element_count 1e9,
// false_positive_probability 0, maximum_size = bytes * 8): probability 0 makes
// compute_optimal_parameters (bloom_filter.hpp:107-155) pick one hash function and cast -inf to
// the table size, which x86-64 builds turn into 2^63 and the clamp into maximum_size -- ONE hash
// over bytes * 8 bits.  Salt :449-508 (salt_count 1), hash_ap :526-568 over the hash's 8 or 4
// bytes, bit = hash % table_size :443-447.
struct ReadsBloom {
    std::vector<uint8_t> bits;
    uint64_t nbits = 0;
    bool use64 = true;
    void init(uint64_t bytes, bool u64)
    {
        nbits = bytes * 8;
        use64 = u64;
        bits.assign((size_t)std::min<uint64_t>(bytes, 1ull << 29), 0);   // a 32-bit hash stays below bit 2^32
    }
    uint64_t bit_of(uint64_t hash) const
    {
        const uint64_t seed = 0xA5A5A5A55A5A5A5Aull * 0xA5A5A5A5ull + 1ull;     // random_seed_
        uint32_t h = 0xAAAAAAAAu * 0xAAAAAAAAu + (uint32_t)seed;                // the filter's only salt
        if (use64) {
            const uint32_t w0 = (uint32_t)hash, w1 = (uint32_t)(hash >> 32);
            h ^= (h << 7) ^ (w0 * (h >> 3)) ^ (~((h << 11) + (w1 ^ (h >> 5))));
        } else {
            h ^= ~((h << 11) + ((uint32_t)hash ^ (h >> 5)));
        }
        return (uint64_t)h % nbits;
    }
    bool test_and_set(uint64_t hash)                        // contains ? true : (insert, false)
    {
        const uint64_t b = bit_of(hash);
        const uint8_t m = (uint8_t)(1u << (b & 7));
        if (bits[b >> 3] & m) return true;
        bits[b >> 3] |= m;
        return false;
    }
};

struct ReadsHeap {
    uint64_t cap, mmin;
    std::map<uint64_t, uint32_t> kept;
    std::map<uint64_t, uint32_t> pending;
    std::priority_queue<uint64_t> pending_q;
    uint64_t msum = 0;                                       // multiplicitySum
    ReadsBloom bloom;                                        // nbits == 0: none

    ReadsHeap(uint64_t s, uint64_t m) : cap(s), mmin(m < 1 ? 1 : m) {}
    bool full() const { return kept.size() >= cap; }
    uint64_t top() const { return kept.rbegin()->first; }
    double multiplicity() const { return kept.empty() ? 0.0 : (double)msum / (double)kept.size(); }   // MinHashHeap.h:44

    void try_insert(uint64_t h)
    {
        if (!(kept.size() < cap || h < top())) return;       // :70-74
        auto it = kept.find(h);
        if (it != kept.end()) {                              // :120-124
            it->second++;
            msum++;
        } else if (bloom.nbits) {                            // :78-94
            if (bloom.test_and_set(h)) {
                kept.emplace(h, 2u);
                msum += 2;
            }
        } else {
            auto pit = pending.find(h);
            const uint64_t pc = pit == pending.end() ? 0 : pit->second;
            if (mmin == 1 || pc == mmin - 1) {               // :96-109
                kept.emplace(h, (uint32_t)mmin);
                msum += mmin;
                if (mmin > 1 && pit != pending.end()) pending.erase(pit);
            } else {                                         // :110-118
                if (pit == pending.end()) { pending_q.push(h); pending.emplace(h, 1u); }
                else pit->second++;
            }
        }
        if (kept.size() > cap) {                             // :126-144
            auto last = std::prev(kept.end());
            const uint64_t tv = last->first;
            msum -= last->second;
            kept.erase(last);
            while (!pending_q.empty() && tv < pending_q.top()) {
                pending.erase(pending_q.top());
                pending_q.pop();
            }
        }
    }
};

}  // namespace

// Reads mode as a SESSION: chunks of whole records in reading order; the heap (incl. the -m pending
// set) lives on the host between chunks, the device sees one chunk at a time, and with -c the caller
// stops reading its files the moment a chunk reports the target coverage.  Host and device memory are
// bounded by one chunk for EVERY reads option (-r, -m, -c, -b): what the reference's reader loop does
// (Sketch.cpp:1196-1270), where mg_sketch_host / mg_sketch_begin keep the whole read set in HBM.
struct mg_reads_session {
    mg_ctx *ctx = nullptr;
    mg_params p;
    int mode = 0;
    ReadsHeap heap;
    bool stopped = false;
    uint64_t used = 0;                  // records consumed when the stop occurred
    uint64_t records = 0;               // records (>= k) seen so far
    double shrink = 1.0;
    uint8_t *d_bases = nullptr;
    uint64_t d_cap = 0;
    uint8_t *d_alpha = nullptr;
    mg::HashEvent *d_ev = nullptr;
    unsigned long long *d_cnt = nullptr;
    std::vector<mg::HashEvent> ev;
    mg_reads_session(uint64_t s, uint64_t m) : heap(s, m) {}
};

static const uint64_t kReadsEventCap = 1ull << 23;           // events per pass (128 MiB)

int mg_reads_begin(mg_ctx *ctx, const mg_params *p, mg_reads_session **out)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!p || !out) return fail(ctx, MG_ERR_INVALID, "mg_reads_begin: NULL argument");
    if (p->kmer_size < 1 || p->kmer_size > 32) return fail(ctx, MG_ERR_INVALID, "mg_sketch: k must be 1..32");
    // (neither -c nor -b: plain reads mode, any min_copies, in constant memory -- nothing stops the reading)
    if (p->bloom_bytes && p->min_copies > 1) return fail(ctx, MG_ERR_INVALID, "mg_reads_begin: min_copies cannot be used with bloom_bytes");   // sketchParameterSetup.cpp:44-48
    if (p->bloom_bytes > (1ull << 60)) return fail(ctx, MG_ERR_INVALID, "mg_reads_begin: bloom_bytes out of range");
    const bool dna = alphabet_is_dna(p);
    if (!p->noncanonical && !dna) return fail(ctx, MG_ERR_UNSUPPORTED, "mg_sketch: canonical k-mers need the ACGT alphabet");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mg_reads_session *rs = new mg_reads_session(p->sketch_size, p->min_copies);
    rs->ctx = ctx;
    rs->p = *p;
    rs->mode = dna ? (p->noncanonical ? 1 : 0) : 2;
    if (p->bloom_bytes) {
        try { rs->heap.bloom.init(p->bloom_bytes, p->use64 != 0); }
        catch (const std::bad_alloc &) { delete rs; return fail(ctx, MG_ERR_NOMEM, "mg_reads_begin: the Bloom filter does not fit in host memory"); }
    }
    if (hipMalloc(&rs->d_alpha, 256) != hipSuccess || hipMalloc(&rs->d_ev, kReadsEventCap * sizeof(mg::HashEvent)) != hipSuccess ||
        hipMalloc(&rs->d_cnt, 8) != hipSuccess ||
        hipMemcpyAsync(rs->d_alpha, p->alphabet, 256, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        mg_reads_free(rs);
        return fail(ctx, MG_ERR_NOMEM, "mg_reads_begin: device allocation failed");
    }
    *out = rs;
    return MG_OK;
}

int mg_reads_add_host(mg_reads_session *rs, const uint8_t *bases, uint64_t nbases, int *stopped_out)
{
    if (!rs) return MG_ERR_INVALID;
    mg_ctx *ctx = rs->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (stopped_out) *stopped_out = rs->stopped ? 1 : 0;
    if (!bases && nbases) return fail(ctx, MG_ERR_INVALID, "mg_reads_add_host: NULL bases");
    if (rs->stopped || nbases == 0) return MG_OK;
    const mg_params *p = &rs->p;
    const uint64_t k = (uint64_t)p->kmer_size;
    // records of the chunk (kseq drops nothing inside a record, so separators are record ends)
    std::vector<uint64_t> rec_begin, rec_end;
    for (uint64_t b = 0; b < nbases;) {
        const void *q = memchr(bases + b, MG_RECORD_SEP, nbases - b);
        const uint64_t e = q ? (uint64_t)((const uint8_t *)q - bases) : nbases;
        if (e - b >= k) { rec_begin.push_back(b); rec_end.push_back(e); }   // shorter records are skipped (Sketch.cpp:1222-1226)
        b = e + 1;
    }
    if (rec_begin.empty()) return MG_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (nbases + 64 > rs->d_cap) {
        if (rs->d_bases) { hipStreamSynchronize(ctx->stream); hipFree(rs->d_bases); rs->d_bases = nullptr; }
        rs->d_cap = 0;
        const uint64_t cap = std::max<uint64_t>(nbases + 64, 1ull << 20);
        if (hipMalloc(&rs->d_bases, cap) != hipSuccess) return fail(ctx, MG_ERR_NOMEM, "mg_reads_add_host: device allocation failed");
        rs->d_cap = cap;
    }
    HIP_TRY(ctx, hipMemcpyAsync(rs->d_bases, bases, nbases, hipMemcpyHostToDevice, ctx->stream));

    ReadsHeap &heap = rs->heap;
    const bool cov = p->target_cov > 0;                     // without -c (a -b session) nothing stops the reading
    const double hash_space = p->use64 ? 18446744073709551616.0 : 4294967296.0;
    const uint64_t tile = mg::sketch_tile(256);
    std::vector<mg::HashEvent> &ev = rs->ev;
    // The unit of a launch is a PIECE: a range of k-mer start positions inside one record.  A record of any length (a
    // chromosome under -r, Sketch.cpp:1196-1270 has no limit) is cut into pieces of at most kPiece positions -- every
    // position yields at most one event, so a piece always fits the event buffer -- and the stop test of -c still
    // follows whole records only (Sketch.cpp:1258).
    constexpr uint64_t kPiece = kReadsEventCap / 4;
    struct Piece { uint64_t pb, pe; uint32_t rec; bool last; };
    std::vector<Piece> pieces;
    for (size_t r = 0; r < rec_begin.size(); r++) {
        const uint64_t p0 = rec_begin[r], p1 = rec_end[r] - k + 1;       // k-mer starts [p0, p1)
        for (uint64_t o = p0; o < p1; o += kPiece) pieces.push_back({o, std::min(p1, o + kPiece), (uint32_t)r, o + kPiece >= p1});
    }
    size_t r0 = 0;                                           // next piece
    size_t rr = 0;                                           // record the replay is in
    bool touched = false;                                    // ... and whether it changed the heap
    const uint64_t want_bytes = 2ull << 20;                  // while the heap is not full everything is an event
    while (r0 < pieces.size() && !rs->stopped) {
        // pieces [r0, r1): as many as are expected to stay within the event capacity
        const uint64_t bound = heap.full() ? heap.top() : 0xFFFFFFFFFFFFFFFFull;
        const double pass = heap.full() ? std::min(1.0, ((double)bound + 1.0) / hash_space) : 1.0;
        uint64_t budget = (uint64_t)std::min<double>((double)(1ull << 40), (double)(kReadsEventCap / 2) / std::max(pass, 1e-12));
        if (budget < want_bytes || !heap.full()) budget = want_bytes;
        budget = (uint64_t)std::max(1.0, (double)budget * rs->shrink);
        size_t r1 = r0;
        uint64_t bytes = 0;
        while (r1 < pieces.size() && (r1 == r0 || bytes + (pieces[r1].pe - pieces[r1].pb) <= budget)) {
            bytes += pieces[r1].pe - pieces[r1].pb;
            r1++;
        }
        // work items: k-mer start positions [b0, b0 + npos), none reading past the last piece's record
        const uint64_t b0 = pieces[r0].pb, b1 = rec_end[pieces[r1 - 1].rec];
        std::vector<mg::SketchWork> work;
        const uint64_t npos = pieces[r1 - 1].pe - b0;
        uint64_t chunk = (npos + 4095) / 4096;
        if (chunk < 2 * tile) chunk = 2 * tile;
        chunk = (chunk + tile - 1) / tile * tile;
        for (uint64_t o = 0; o < npos; o += chunk) {
            mg::SketchWork w;
            w.begin = b0 + o; w.end = b0 + std::min(npos, o + chunk); w.limit = b1;
            w.sketch = 0; w.slot = 0; w.nchunks = 1; w._pad = 0;
            work.push_back(w);
        }
        DevBuf<mg::SketchWork> d_work(ctx);
        if (d_work.alloc(work.size()) != hipSuccess) return fail(ctx, MG_ERR_NOMEM, "mg_reads_add_host: device allocation failed");
        mg::EventArgs ea;
        ea.bases = rs->d_bases; ea.work = d_work; ea.alphabet = rs->d_alpha; ea.out = rs->d_ev; ea.count = rs->d_cnt;
        ea.capacity = kReadsEventCap; ea.bound = bound; ea.seed = p->seed; ea.use64 = p->use64;
        ea.fold_case = p->preserve_case ? 0 : 1;
        unsigned long long n_ev = 0;
        hipError_t e = hipMemcpyAsync(d_work, work.data(), work.size() * sizeof(mg::SketchWork), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(rs->d_cnt, 0, 8, ctx->stream);
        if (e == hipSuccess) e = mg::launch_hash_events(p->kmer_size, rs->mode, ea, (uint32_t)work.size(), ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&n_ev, rs->d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("mg_reads_add_host: ") + hipGetErrorString(e));
        if (n_ev > kReadsEventCap) {                         // denser than expected: take fewer pieces
            if (r1 - r0 == 1) return fail(ctx, MG_ERR_HIP, "mg_reads_add_host: more events than k-mer positions in a piece");
            rs->shrink /= 4;
            continue;
        }
        rs->shrink = 1.0;
        ev.resize(n_ev);
        if (n_ev && hipMemcpy(ev.data(), rs->d_ev, n_ev * sizeof(mg::HashEvent), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(ctx, MG_ERR_HIP, "mg_reads_add_host: D2H copy failed");
        std::sort(ev.begin(), ev.end(), [](const mg::HashEvent &x, const mg::HashEvent &y) { return x.pos < y.pos; });
        // replay, record by record; the stop test follows every record that changed the heap
        auto close_records = [&](size_t upto) {               // records [rr, upto) are complete
            if (rr < upto) {
                if (cov && touched && heap.multiplicity() >= p->target_cov) { rs->stopped = true; rs->used = rs->records + rr + 1; }
                touched = false;
                rr = upto;
            }
        };
        for (size_t i = 0; i < ev.size() && !rs->stopped; i++) {
            size_t at = rr;
            while (ev[i].pos >= rec_end[at]) at++;           // the event's record
            close_records(at);
            if (rs->stopped) break;
            heap.try_insert(ev[i].hash);
            touched = true;
        }
        if (!rs->stopped) close_records(pieces[r1 - 1].last ? (size_t)pieces[r1 - 1].rec + 1 : (size_t)pieces[r1 - 1].rec);
        r0 = r1;
    }
    rs->records += rec_begin.size();
    if (stopped_out) *stopped_out = rs->stopped ? 1 : 0;
    return MG_OK;
}

int mg_reads_finish(mg_reads_session *rs, uint64_t *hashes_out, uint32_t *nhash_out, uint32_t *counts_out, uint64_t *records_used_out)
{
    if (!rs) return MG_ERR_INVALID;
    if (!hashes_out || !nhash_out) return fail(rs->ctx, MG_ERR_INVALID, "mg_reads_finish: NULL argument");
    const uint64_t s = rs->p.sketch_size;
    for (uint64_t i = 0; i < s; i++) hashes_out[i] = MG_HASH_PAD;
    if (counts_out) memset(counts_out, 0, s * 4);
    uint32_t n = 0;
    for (const auto &kv : rs->heap.kept) {
        hashes_out[n] = kv.first;
        if (counts_out) counts_out[n] = kv.second;
        n++;
    }
    *nhash_out = n;
    if (records_used_out) *records_used_out = rs->stopped ? rs->used : rs->records;
    return MG_OK;
}

int mg_reads_reset(mg_reads_session *rs)
{
    if (!rs) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(rs->ctx->mu);
    ReadsBloom bloom;
    std::swap(bloom, rs->heap.bloom);                       // keep the filter's memory, clear its bits
    std::fill(bloom.bits.begin(), bloom.bits.end(), 0);
    rs->heap = ReadsHeap(rs->p.sketch_size, rs->p.min_copies);
    std::swap(bloom, rs->heap.bloom);
    rs->stopped = false;
    rs->used = rs->records = 0;
    rs->shrink = 1.0;
    return MG_OK;
}

void mg_reads_free(mg_reads_session *rs)
{
    if (!rs) return;
    hipSetDevice(rs->ctx->device);
    hipStreamSynchronize(rs->ctx->stream);
    for (void *q : {(void *)rs->d_bases, (void *)rs->d_alpha, (void *)rs->d_ev, (void *)rs->d_cnt})
        if (q) hipFree(q);
    delete rs;
}

int mg_sketch_reads_host(mg_ctx *ctx, const mg_params *p, const uint8_t *bases, uint64_t nbases, uint64_t *hashes_out,
                         uint32_t *nhash_out, uint32_t *counts_out, uint64_t *records_used_out)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!p || !hashes_out || !nhash_out || (!bases && nbases)) return fail(ctx, MG_ERR_INVALID, "mg_sketch_reads_host: NULL argument");
    if (p->kmer_size < 1 || p->kmer_size > 32) return fail(ctx, MG_ERR_INVALID, "mg_sketch: k must be 1..32");
    if (!(p->target_cov > 0) && p->bloom_bytes == 0) {
        // records of the batch (shorter ones are skipped, Sketch.cpp:1222-1226): the "reads used" of a run without -c
        const uint64_t k = (uint64_t)p->kmer_size;
        uint64_t nrec = 0;
        for (uint64_t b = 0; b < nbases;) {
            const void *q = memchr(bases + b, MG_RECORD_SEP, nbases - b);
            const uint64_t e = q ? (uint64_t)((const uint8_t *)q - bases) : nbases;
            if (e - b >= k) nrec++;
            b = e + 1;
        }
        if (records_used_out) *records_used_out = nrec;
        mg_params q = *p;
        q.target_cov = 0;
        const uint64_t off[2] = {0, nbases};
        return mg_sketch_host(ctx, &q, bases, nbases, off, 1, hashes_out, nhash_out, counts_out);
    }
    // one chunk through the session
    mg_reads_session *rs = nullptr;
    int rc = mg_reads_begin(ctx, p, &rs);
    if (rc != MG_OK) return rc;
    rc = mg_reads_add_host(rs, bases, nbases, nullptr);
    if (rc == MG_OK) rc = mg_reads_finish(rs, hashes_out, nhash_out, counts_out, records_used_out);
    mg_reads_free(rs);
    return rc;
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
                        (void *)sp->ulist, (void *)sp->upos, (void *)sp->gdata, (void *)sp->xm, (void *)sp->ext, (void *)sp->inv, (void *)sp->phashes})
            if (q) ctx_free(ctx, q);
        delete sp;
    }
    t->sparse.clear();
    t->cls.clear();
    t->last.clear();
    t->nh.clear();
    t->have_max = false;
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

/* ------------------------------------------------------------------ comparing */

// largest hash of a table (device reduction, cached)
static int table_max(mg_ctx *ctx, const mg_table *t, uint64_t *out)
{
    if (!t->have_max) {
        unsigned long long *d = nullptr;
        HIP_TRY(ctx, hipMalloc(&d, 8));
        hipError_t e = hipMemsetAsync(d, 0, 8, ctx->stream);
        if (e == hipSuccess) e = mg::launch_table_max(t->hashes, t->nhash, t->n, t->s, d, ctx->stream);
        unsigned long long h = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        hipFree(d);
        if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("table max: ") + hipGetErrorString(e));
        t->maxval = h;
        t->have_max = true;
    }
    *out = t->maxval;
    return MG_OK;
}

// u32 prefix image of a table for shift `shr` (cached; a table mixing very different hash
// densities is compared class by class, each class through its own shift)
static int table_prefix(mg_ctx *ctx, const mg_table *t, int shr, const uint32_t **out)
{
    for (auto &im : t->pfx)
        if (im.first == shr) { *out = im.second; return MG_OK; }
    if (t->pfx.size() >= 24) {                             // keep the cache bounded
        hipStreamSynchronize(ctx->stream);
        hipFree(t->pfx.front().second);
        t->pfx.erase(t->pfx.begin());
    }
    const uint64_t ps = mg::compare_pfx_stride(t->s);
    uint32_t *img = nullptr;
    HIP_TRY(ctx, hipMalloc(&img, std::max<uint64_t>(t->n * ps * 4, 4)));
    hipError_t e = mg::launch_make_prefix(t->hashes, t->nhash, t->n, t->s, ps, (uint32_t)shr, img, ctx->stream);
    if (e != hipSuccess) { hipFree(img); return fail(ctx, MG_ERR_HIP, std::string("compare (prefix image): ") + hipGetErrorString(e)); }
    t->pfx.emplace_back(shr, img);
    *out = img;
    return MG_OK;
}

// density class of every row (bit length of the mean hash spacing), computed once per table
static int table_classes(mg_ctx *ctx, const mg_table *t)
{
    if (t->cls.size() == t->n) return MG_OK;
    uint8_t *d = nullptr;
    unsigned long long *dl = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, std::max<uint64_t>(t->n, 1)));
    if (hipMalloc(&dl, std::max<uint64_t>(t->n, 1) * 8) != hipSuccess) { hipFree(d); return fail(ctx, MG_ERR_NOMEM, "compare: allocation failed"); }
    std::vector<uint8_t> h(t->n);
    std::vector<uint64_t> hl(t->n);
    std::vector<uint32_t> hn(t->n);
    hipError_t e = mg::launch_row_classes(t->hashes, t->nhash, t->n, t->s, d, dl, ctx->stream);
    if (e == hipSuccess && t->n) e = hipMemcpyAsync(hn.data(), t->nhash, t->n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && t->n) e = hipMemcpyAsync(h.data(), d, t->n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && t->n) e = hipMemcpyAsync(hl.data(), dl, t->n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d);
    hipFree(dl);
    if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("compare (row classes): ") + hipGetErrorString(e));
    t->cls.swap(h);
    t->last.swap(hl);
    t->nh.swap(hn);
    return MG_OK;
}

// Window offsets of a table for the large-sketch compare path: for every row and every boundary
// w * delta (w < nwin; boundary nwin = end of the row) the index of the first hash whose prefix
// (shift shr) is at or above it, over the row's first min(nhash, s) hashes.  Device array of
// (nwin + 1) per row plus a host copy (the host sizes tiles from it); cached per geometry.
static int table_windows(mg_ctx *ctx, const mg_table *t, int shr, uint32_t delta, uint32_t nwin, uint32_t s,
                         const mg_table::Windows **out)
{
    for (auto &w : t->win)
        if (w.shr == shr && w.delta == delta && w.nwin == nwin && w.s == s) { *out = &w; return MG_OK; }
    if (t->win.size() >= 8) {
        hipStreamSynchronize(ctx->stream);
        hipFree(t->win.front().dev);
        t->win.erase(t->win.begin());
    }
    const uint32_t *img = nullptr;
    int rc = table_prefix(ctx, t, shr, &img);
    if (rc != MG_OK) return rc;
    mg_table::Windows w;
    w.shr = shr; w.delta = delta; w.nwin = nwin; w.s = s; w.dev = nullptr;
    const uint64_t count = t->n * (uint64_t)(nwin + 1);
    HIP_TRY(ctx, hipMalloc(&w.dev, std::max<uint64_t>(count, 1) * 4));
    w.host.resize(count);
    hipError_t e = mg::launch_window_offsets(img, mg::compare_pfx_stride(t->s), t->nhash, t->n, s, nwin, delta, w.dev, ctx->stream);
    if (e == hipSuccess && count) e = hipMemcpyAsync(w.host.data(), w.dev, count * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { hipFree(w.dev); return fail(ctx, MG_ERR_HIP, std::string("compare (window offsets): ") + hipGetErrorString(e)); }
    t->win.push_back(std::move(w));
    *out = &t->win.back();
    return MG_OK;
}

// Copies a tile list to the device through a slot of the context's staging ring (grown on demand);
// tiles_release marks the slot as in use until the launches queued so far are done.
static int stage_tiles(mg_ctx *ctx, const void *tiles, size_t bytes, void **dev_out, int *slot_out)
{
    const int si = (int)(ctx->slot_next++ % 4u);
    mg_ctx::TileSlot &sl = ctx->slots[si];
    if (!sl.done) HIP_TRY(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (sl.pending) {                                      // four launches behind at most
        HIP_TRY(ctx, hipEventSynchronize(sl.done));
        sl.pending = false;
    }
    if (bytes > sl.cap) {
        if (sl.dev) { hipFree(sl.dev); sl.dev = nullptr; }
        if (sl.host) { hipHostFree(sl.host); sl.host = nullptr; }
        sl.cap = 0;
        const size_t cap = std::max<size_t>(bytes + bytes / 2, 1u << 16);
        HIP_TRY(ctx, hipMalloc(&sl.dev, cap));
        HIP_TRY(ctx, hipHostMalloc(&sl.host, cap, hipHostMallocDefault));
        sl.cap = cap;
    }
    memcpy(sl.host, tiles, bytes);
    HIP_TRY(ctx, hipMemcpyAsync(sl.dev, sl.host, bytes, hipMemcpyHostToDevice, ctx->stream));
    *dev_out = sl.dev;
    *slot_out = si;
    return MG_OK;
}

static int tiles_release(mg_ctx *ctx, int slot)
{
    mg_ctx::TileSlot &sl = ctx->slots[slot];
    HIP_TRY(ctx, hipEventRecord(sl.done, ctx->stream));
    sl.pending = true;
    return MG_OK;
}

// Value windows of one density class (see run_compare_merged): the window width delta (prefix
// domain) is taken from the class's densest row so that it has `target` hashes per window, and the
// rows are then cut into tiles greedily by their ACTUAL window offsets: a tile takes rows (in
// order) while it has fewer than the kernel's row limit and its share of every window fits the
// tile table.  The plan stands only if no row has more hashes in a window than a tag can index --
// else a narrower second try, else no plan (the class uses plain tiles).
struct WindowPlan {
    const mg_table::Windows *rows = nullptr, *cols = nullptr;
    uint32_t delta = 0, nwin = 0;
    std::vector<std::pair<uint32_t, uint32_t>> groups;    // tiles: [first, last) positions in the class's row list
};

// Hashes of the densest row per window.  A pair of unrelated sketches is decided once the union of
// the two reaches s elements, i.e. after ~0.5 s hashes of either (0.537 s covers the spread of
// that point over a tile's pairs); cost per pair ~ windows until then x (hashes + fixed cost per
// window and column) / rows per tile, rows = what fits the tile table.
static double window_target(uint32_t s, uint32_t rows_max)
{
    const double need = 0.537 * (double)s, fixed = 150.0;
    const double row_cap = (double)mg::compare_window_row_entries() * 0.45, ecap = (double)mg::compare_window_entries();
    double best = 0, best_cost = 1e300;
    for (int m = 1; m <= 255; m++) {
        const double tw = std::ceil(need / m);
        if (tw > row_cap) continue;
        const double rows = std::min((double)rows_max, std::floor(0.97 * ecap / tw));
        if (rows < 1) continue;
        const double cost = m * (tw + fixed) / rows;
        if (cost < best_cost) { best_cost = cost; best = tw; }
        if (tw <= 64) break;
    }
    return best > 0 ? best : std::min(row_cap, need);
}

// `must`: the sketches are too large for plain tiles (s > 16 384), so a class that would be served
// by one window (few hashes, or none) still gets a plan -- of that single window.
static int plan_windows(mg_ctx *ctx, const mg_table *rows, const mg_table *cols, const std::vector<uint32_t> &list, int shr,
                        uint64_t xmax, uint32_t s, bool must, WindowPlan *out, uint32_t row_cap = 0)
{
    if (row_cap == 0) row_cap = mg::compare_window_row_entries();         // entries of one row a tile's tag can index
    const uint32_t Rw = mg::compare_window_rows(s);
    double target = window_target(s, Rw);                                    // entries of the densest row per window
    if (const char *e = ctx_opt(ctx, "MASHGPU_COMPARE_WIN_TARGET")) target = std::max(1.0, atof(e));
    // Hashes per unit of prefix of a row at the class's 10th percentile: rows at least that dense
    // (90 % of them) have `target` hashes or more in a window, so their pairs are decided where the
    // target says.  (Taken from the DENSEST row, typical rows fell a few per cent short of it and one
    // pair in six stayed open after the first window -- every column then ran twice.)  What a tile
    // holds is decided below from the actual offsets, whatever the density of its rows.
    double dens = 0;
    {
        std::vector<double> d;
        d.reserve(list.size());
        for (uint32_t i : list) {
            const uint64_t ni = std::min<uint64_t>(rows->nh[i], s);
            if (ni) d.push_back((double)ni / ((double)(rows->last[i] >> shr) + 1.0));
        }
        if (!d.empty()) {
            const size_t q = d.size() / 10;
            std::nth_element(d.begin(), d.begin() + (long)q, d.end());
            dens = d[q];
        }
    }
    if (dens <= 0 && !must) return MG_OK;
    for (int attempt = 0; attempt < 2; attempt++, target *= 0.7) {
        double dd = dens > 0 ? std::floor(target / dens) : (double)xmax + 1.0;
        if (dd < 1.0) return MG_OK;
        if (dd >= (double)xmax + 1.0) {                                      // one window: nothing to gain
            if (!must) return MG_OK;
            dd = (double)xmax + 1.0;
        }
        const uint32_t delta = (uint32_t)dd;
        const uint64_t nw = (xmax + delta) / delta;                          // ceil((xmax + 1) / delta)
        if ((nw < 2 && !must) || nw > 255) return MG_OK;
        const uint32_t nwin = (uint32_t)nw;
        const mg_table::Windows *cand = nullptr;
        int rc = table_windows(ctx, rows, shr, delta, nwin, s, &cand);
        if (rc != MG_OK) return rc;
        bool fits = true;
        std::vector<std::pair<uint32_t, uint32_t>> groups;
        std::vector<uint32_t> tot(nwin, 0);
        uint32_t g0 = 0;
        for (uint32_t k = 0; k < list.size() && fits; k++) {
            const uint32_t *o = &cand->host[(uint64_t)list[k] * (nwin + 1)];
            bool room = k - g0 < Rw;
            for (uint32_t w = 0; w < nwin; w++) {
                const uint32_t c = o[w + 1] - o[w];
                if (c > row_cap) fits = false;                               // the tag's index field
                if (tot[w] + c > mg::compare_window_entries()) room = false;
            }
            if (!room) {                                                     // row k opens the next tile
                groups.emplace_back(g0, k);
                g0 = k;
                std::fill(tot.begin(), tot.end(), 0u);
            }
            for (uint32_t w = 0; w < nwin; w++) tot[w] += o[w + 1] - o[w];
        }
        if (!fits) continue;
        if (g0 < list.size()) groups.emplace_back(g0, (uint32_t)list.size());
        const mg_table::Windows *wc = nullptr;
        rc = table_windows(ctx, cols, shr, delta, nwin, s, &wc);
        if (rc != MG_OK) return rc;
        out->rows = rows == cols ? wc : cand;
        out->cols = wc;
        out->delta = delta;
        out->nwin = nwin;
        out->groups.swap(groups);
        return MG_OK;
    }
    return MG_OK;
}

// The merged-rows engine (compare_merged.hip) over rows [row_begin, row_end): density classes,
// per-class prefix images, optional value windows, tile lists, launches.
// `windows_only`: s is beyond plain tiles; every class must get a window plan, else nothing is
// launched and kNoWindowPlan is returned (the caller falls back to the generic kernel).
static const int kNoWindowPlan = -1000;

static int run_compare_merged(mg_ctx *ctx, const mg_table *rows, const mg_table *cols, uint64_t row_begin, uint64_t row_end,
                              bool triangle, mg::CompareArgs &a, uint32_t R, uint64_t CC, uint64_t maxcols, bool windows_only)
{
    // Rows are grouped by hash DENSITY before they are cut into tiles of R: one linear
    // value -> bucket map per tile spreads the entries evenly only if its rows are equally
    // dense, and collections mix genomes of very different sizes (a virus sketch spans the
    // whole hash range, a bacterial one its bottom 1/5000).  Within a class rows keep their
    // order, so a tile's rows stay close together and the triangle's "columns below the
    // row" rule wastes little: a tile runs to its largest row.  Every class is compared
    // through its own 32-bit prefix image (value >> shr, shr from the class maximum, larger
    // values saturate): a prefix must resolve the values of the tile's rows, or equal
    // prefixes of different values send block after block down the exact path.
    int rc = table_classes(ctx, rows);
    if (rc != MG_OK) return rc;
    std::vector<std::vector<uint32_t>> by_class(65);
    for (uint64_t i = row_begin; i < row_end; i++) by_class[rows->cls[i] > 64 ? 64 : rows->cls[i]].push_back((uint32_t)i);
    // Large sketches: R*s <= ~16 000 leaves few rows per tile (2 at s = 10 000), and a probe
    // serves only that many pairs.  They are compared VALUE WINDOW by value window instead:
    // a launch handles the hashes of one prefix range, sized so that 16 rows' share of it fills
    // the tile table; a pair carries its match count from launch to launch in its output slot
    // and drops out once its union reaches s (see compare_merged.hip, WIN).
    // Smaller sketches use the same mode with TWO windows or so: unrelated pairs are decided by the
    // lower half of the hash range (the union of two sketches reaches s elements there), so the
    // first window holds ~0.54 s hashes of a row and 29 rows share a tile -- and a probe -- instead
    // of 16; the few pairs still open (related sketches) go on to the next window.
    // Measured (profiles/r02_engine_sweep.txt, s = 1000): the window engine wins from ~30 000 sketches
    // on (40 000: 16.9 vs 15.5e9 pairs/s; 70 000: 22.2 vs 17.2; 100 000: 26.5 vs 17.7) and loses below
    // (20 000: 10.2 vs 11.9; 10 000: 6.0 vs 7.5) -- more launches, each with its tail, and a table
    // build per tile and window -- so small jobs keep plain tiles; large sketches (s >= 1800) always
    // take windows (plain tiles would hold 8 rows or fewer).  With the round-2 kernel
    // (tools/small_n_profile.py) the crossover sits at ~23 000 sketches for s = 1000 (28 000: 16.4 vs
    // 14.8; 20 000: 12.1 vs 12.9) and at ~11 000 for s = 400 (20 000: 27.3 vs 21.7; 10 000: 16.2 vs
    // 16.6): rows x columns >= 1.4e8 up to s = 400, rising linearly to 5.5e8 at s = 1000.
    const double win_cross = a.s <= 400 ? 1.4e8 : a.s >= 1000 ? 5.5e8 : 1.4e8 + (a.s - 400.0) * (4.1e8 / 600.0);
    bool want_win = a.s >= 1800 || (a.s >= 200 && (double)(row_end - row_begin) * (double)maxcols >= win_cross);
    if (const char *e = ctx_opt(ctx, "MASHGPU_COMPARE_WINDOWS")) want_win = atoi(e) != 0;
    if (windows_only) want_win = true;
    const uint32_t R_plain = R;
    // A launch of few row tiles (a handful of queries against a large database, or a small
    // density class) would leave most CUs idle with full-length column chunks: cut the columns
    // finer until there are ~4 tiles per CU (a table build costs about as much as 100 columns,
    // so not below 256).  Every class is its own launch, so this is decided per class.
    const bool cc_forced = ctx_opt(ctx, "MASHGPU_COMPARE_COLS") != nullptr;
    auto chunk_for = [&](uint64_t nrt) -> uint64_t {
        // measured (profiles/r02_engine_sweep.txt): 2048 tiles pay from ~10 000 columns on (n = 10 000:
        // 4.1 -> 6.0e9 pairs/s windows, 6.3 -> 7.5e9 plain); below that the tiles get too short for their builds
        uint64_t min_tiles = maxcols >= 8192 ? 2048 : 512;  // (MASHGPU_COMPARE_MIN_TILES: tuning knob)
        if (const char *e = ctx_opt(ctx, "MASHGPU_COMPARE_MIN_TILES")) min_tiles = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
        if (cc_forced || nrt == 0 || nrt * ((maxcols + CC - 1) / CC) >= min_tiles) return CC;
        uint64_t chunks = (2 * min_tiles + nrt - 1) / nrt;
        const uint64_t most = std::max<uint64_t>(1, maxcols / 256);
        if (chunks > most) {
            // the floor of 256 columns binds: then at least fill whole rounds of the CUs
            chunks = most;
            const uint64_t cus = ctx->cu_count > 0 ? (uint64_t)ctx->cu_count : 256;
            if (nrt * chunks > cus) chunks = std::max<uint64_t>(1, (nrt * chunks / cus) * cus / nrt);
        }
        const uint64_t cc = ((maxcols + chunks - 1) / chunks + 7) & ~7ull;
        return std::min<uint64_t>(CC, std::max<uint64_t>(256, cc));
    };
    a.dbg = nullptr;
    a.row_pfx_stride = mg::compare_pfx_stride(rows->s);
    a.col_pfx_stride = mg::compare_pfx_stride(cols->s);
    // prefix shift of a class: its largest hash must stay below the three reserved prefixes
    auto class_shift = [&](const std::vector<uint32_t> &list, uint64_t *mx_out) -> int {
        uint64_t mx = 1;
        for (uint32_t i : list) mx = std::max(mx, rows->last[i]);
        const int bl = 64 - __builtin_clzll(mx);
        int shr = bl > 32 ? bl - 32 : 0;
        if ((mx >> shr) >= 0xFFFFFFFDull) shr++;      // 0xFFFFFFFD..F: saturated values, sentinel, padding
        *mx_out = mx;
        return shr;
    };
    if (windows_only) {
        // nothing may be launched unless every class can be windowed
        for (const auto &list : by_class) {
            if (list.empty()) continue;
            uint64_t mx;
            const int shr = class_shift(list, &mx);
            WindowPlan plan;
            rc = plan_windows(ctx, rows, cols, list, shr, mx >> shr, a.s, true, &plan);
            if (rc != MG_OK) return rc;
            if (!plan.rows) return kNoWindowPlan;
        }
    }
    for (const auto &list : by_class) {
        if (list.empty()) continue;
        uint64_t mx;
        const int shr = class_shift(list, &mx);
        rc = table_prefix(ctx, rows, shr, &a.row_pfx);
        if (rc == MG_OK) rc = table_prefix(ctx, cols, shr, &a.col_pfx);
        if (rc != MG_OK) return rc;
        a.pfx_shr = (uint32_t)shr;
        // ---- window plan of this class (large sketches) ----
        WindowPlan plan;
        const uint64_t xmax = mx >> shr;
        if (want_win) {
            rc = plan_windows(ctx, rows, cols, list, shr, xmax, a.s, windows_only, &plan);
            if (rc != MG_OK) return rc;
            if (windows_only && !plan.rows) return fail(ctx, MG_ERR_HIP, "compare: window plan changed between passes");
        }
        const mg_table::Windows *wr = plan.rows, *wc = plan.cols;
        const uint32_t delta = plan.delta, nwin = plan.nwin;
        // rows of a tile: positions [first, last) of the class's list -- the window plan's groups, or R at a time
        std::vector<std::pair<uint32_t, uint32_t>> plain_groups;
        if (!wr)
            for (size_t k = 0; k < list.size(); k += R_plain) plain_groups.emplace_back((uint32_t)k, (uint32_t)std::min(list.size(), k + R_plain));
        const std::vector<std::pair<uint32_t, uint32_t>> &groups = wr ? plan.groups : plain_groups;
        a.rows_per_tile = wr ? mg::compare_window_rows(a.s) : R_plain;
        const uint64_t CCc = chunk_for(groups.size());
        std::vector<mg::MergedTile> mtiles;
        // Longest tiles first: in a triangle a row group needs the columns below its last row, so within
        // a column chunk the tiles grow with the row index (from a handful of columns to the whole chunk).
        // Handing the workgroups out in that order left the largest tiles for the end -- at 20 000 sketches
        // a tail of one full tile, a fifth of the launch; later chunks hold ever fewer and shorter tiles,
        // so chunk-major order with the groups reversed is longest-first overall.
        for (uint64_t c0 = 0; c0 < maxcols; c0 += CCc) {
            for (auto git = groups.rbegin(); git != groups.rend(); ++git) {
                const auto &g = *git;
                const uint32_t last = list[g.second - 1];
                const uint64_t cend = triangle ? last : cols->n;         // columns needed: [0, cend)
                if (c0 >= cend) continue;
                mg::MergedTile tl;
                for (uint32_t r = 0; r < 32; r++) tl.rows[r] = g.first + r < g.second ? list[g.first + r] : 0xFFFFFFFFu;
                tl.col0 = (uint32_t)c0;
                tl.col1 = (uint32_t)std::min<uint64_t>(c0 + CCc, cend);
                mtiles.push_back(tl);
            }
        }
        if (mtiles.empty()) continue;
        void *d_mt = nullptr;
        int slot = 0;
        rc = stage_tiles(ctx, mtiles.data(), mtiles.size() * sizeof(mg::MergedTile), &d_mt, &slot);
        if (rc != MG_OK) return rc;
        unsigned long long *d_dbg = nullptr;
        const size_t dbg_sets = wr ? nwin : 1;                  // one {start, built, end} set per tile and launch
        if (ctx_opt(ctx, "MASHGPU_COMPARE_DBG")) {
            hipMalloc(&d_dbg, dbg_sets * mtiles.size() * 24);
            hipMemsetAsync(d_dbg, 0, dbg_sets * mtiles.size() * 24, ctx->stream);
        }
        a.dbg = d_dbg;
        a.mtiles = static_cast<const mg::MergedTile *>(d_mt);
        hipError_t e = hipSuccess;
        void *d_mask = nullptr;
        if (wr) {
            // live-column masks: one byte per wave and batch of 8 columns, kept between the launches
            a.win_kmax = (uint32_t)(((CCc + 7) / 8 + 15) / 16);
            const size_t mbytes = mtiles.size() * 16 * (size_t)a.win_kmax;
            if (ctx_malloc(ctx, &d_mask, mbytes) != hipSuccess) return fail(ctx, MG_ERR_NOMEM, "compare: allocation failed (window masks)");
            e = hipMemsetAsync(d_mask, 0, mbytes, ctx->stream);
            a.win_mask = static_cast<uint8_t *>(d_mask);
        }
        if (wr && e == hipSuccess) {
            a.row_win = wr->dev;
            a.col_win = wc->dev;
            a.nwin = nwin;
            for (uint32_t w = 0; w < nwin && e == hipSuccess; w++) {         // stream order: window w + 1 resumes window w
                a.win = w;
                a.win_lo = w * delta;
                a.win_hi = (uint32_t)std::min<uint64_t>((uint64_t)(w + 1) * delta, xmax + 1);
                a.dbg = d_dbg ? d_dbg + (size_t)w * mtiles.size() * 3 : nullptr;
                prof_begin(ctx, ctx->prof_compare);
                e = mg::launch_compare_merged(a, (uint32_t)mtiles.size(), ctx->stream);
                prof_end(ctx, ctx->prof_compare);
            }
            a.row_win = a.col_win = nullptr;
            a.nwin = a.win = 0;
            a.win_mask = nullptr;
        } else if (!wr) {
            prof_begin(ctx, ctx->prof_compare);
            e = mg::launch_compare_merged(a, (uint32_t)mtiles.size(), ctx->stream);
            prof_end(ctx, ctx->prof_compare);
        }
        // (no synchronisation: the tile list sits in its own slot of the ring, the masks go back to
        //  the block cache in stream order)
        hipError_t e2 = e == hipSuccess ? (tiles_release(ctx, slot) == MG_OK ? hipSuccess : hipErrorUnknown) : hipSuccess;
        ctx_free(ctx, d_mask);
        if (d_dbg) {
            hipStreamSynchronize(ctx->stream);
            std::vector<unsigned long long> h(dbg_sets * mtiles.size() * 3);
            hipMemcpy(h.data(), d_dbg, h.size() * 8, hipMemcpyDeviceToHost);
            for (size_t w = 0; w < dbg_sets; w++) {
                double bsum = 0, tsum = 0, tmax = 0;
                unsigned long long first = ~0ull, last = 0;
                for (size_t i = 0; i < mtiles.size(); i++) {
                    const unsigned long long *q = &h[(w * mtiles.size() + i) * 3];
                    bsum += (double)(q[1] - q[0]);
                    tsum += (double)(q[2] - q[0]);
                    tmax = std::max(tmax, (double)(q[2] - q[0]));
                    first = std::min(first, q[0]);
                    last = std::max(last, q[2]);
                }
                fprintf(stderr, "compare dbg: shift %d window %zu/%zu, %zu rows, %zu tiles, build %.0f clk avg, tile %.0f clk avg, "
                        "longest %.0f, launch %.0f\n", shr, w, dbg_sets, list.size(), mtiles.size(), bsum / mtiles.size(),
                        tsum / mtiles.size(), tmax, (double)(last - first));
            }
            hipFree(d_dbg);
        }
        if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("compare launch: ") + hipGetErrorString(e));
        if (e2 != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("compare kernel: ") + hipGetErrorString(e2));
    }
    return MG_OK;
}

// ---- inverted-index engine (compare_sparse.hip) ----------------------------------------------
//
// Index of a table for sketch size s: every (value, row) entry of the rows' first min(nhash, s)
// hashes sorted by value, rows ascending inside a value (stable sort over row-major image indices).
// Built once per table and sketch size, cached in the mg_table like the prefix images (dropped by
// mg_table_invalidate; its blocks then go back to the context's pool and the next table of the same shape
// takes them).  Retained: the sorted values (rect queries are located in them), the row per sorted
// position, the end of every group of equal values (kept at the group's first position), and two images in
// the table's own layout (row stride rs): the CODE image (2 x first sorted position of the entry's value --
// ordered and equal exactly as the values are) and the POSITION image (the entry's own sorted position: the
// rows below it that hold the same value are sorted_rows[code / 2 .. position)).  A table the engine cannot
// take (2^31 entries and more, a real hash equal to the padding value, no memory) is marked unusable and
// keeps the tile engine.
// Host work per build is O(n) loops and three synchronisations (row classes, copy suspects, build
// statistics); sorting of digests and of the visiting order happens on the device.
static int table_sparse_index(mg_ctx *ctx, const mg_table *t, uint32_t s, bool clustered, mg_table::Sparse **out)
{
    for (mg_table::Sparse *sp : t->sparse)
        if (sp->s == s && sp->clustered == clustered) { *out = sp; return MG_OK; }
    int rc = table_classes(ctx, t);                        // host copies of nhash and the rows' largest hashes
    if (rc != MG_OK) return rc;
    mg_table::Sparse *sp = new mg_table::Sparse;
    sp->s = s;
    sp->clustered = clustered;
    t->sparse.push_back(sp);
    *out = sp;
    auto unusable = [&](const char *why) { sp->usable = false; sp->why = why; return MG_OK; };
    const uint64_t n = t->n;
    if (n == 0) return unusable("empty table");
    if (n >= (1ull << 31)) return unusable("too many rows");
    sp->rs = mg::sparse_img_stride(s);
    if (n * sp->rs >= (1ull << 32)) return unusable("image index beyond 32 bits");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const auto t_begin = std::chrono::steady_clock::now();
    // (the "index" phase of the library's HIP-event records: everything this function queues -- clustering, digests, sort,
    //  images, dense groups -- incl. the waits between its steps)
    prof_begin(ctx, ctx->prof_index);
    struct ProfEnd { mg_ctx *c; ~ProfEnd() { prof_end(c, c->prof_index); } } prof_end_guard{ctx};
    // ---- identical rows (see compare_sparse.hip): digest every row, sort the digests on the device; rows whose
    // digest and length equal their predecessor's are suspects, verified value by value; copies then stay out
    // of the index
    std::vector<uint32_t> cnt_true(n);
    uint32_t max_cnt = 0;
    for (uint64_t i = 0; i < n; i++) {
        cnt_true[i] = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(t->nh[i], t->s), s);
        max_cnt = std::max(max_cnt, cnt_true[i]);
    }
    std::vector<uint32_t> rep;                              // empty: no copies
    DevBuf<uint32_t> d_cnt(ctx);
    if (d_cnt.alloc(n) != hipSuccess) { (void)hipGetLastError(); return unusable("no device memory for the index"); }
    HIP_TRY(ctx, hipMemcpyAsync(d_cnt, cnt_true.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    // ---- the clustered variant: rows that share one of their smallest hashes next to each other (labels on the device,
    // two small sorts), the table copied in that order; everything below then works on the copy as if it were the table
    const uint64_t *H = t->hashes;                          // what the index is built from
    std::vector<uint64_t> last_p;                           // the rows' largest hashes in index order (empty: t->last)
    std::vector<uint32_t> lab_sorted;                       // label of every index row (clustered variant)
    if (clustered && n >= 16) {
        DevBuf<unsigned long long> k_a(ctx), k_b(ctx);
        DevBuf<uint32_t> r_a(ctx), r_b(ctx), l_a(ctx), l_b(ctx), d_inv(ctx), d_lab(ctx);
        DevBuf<unsigned char> d_tmp(ctx);
        const size_t tb = mg::dense_cluster_temp_bytes((uint32_t)n);
        std::vector<uint32_t> inv(n);
        lab_sorted.resize(n);
        if (k_a.alloc(4 * n) == hipSuccess && k_b.alloc(4 * n) == hipSuccess && r_a.alloc(4 * n) == hipSuccess && r_b.alloc(4 * n) == hipSuccess &&
            l_a.alloc(n) == hipSuccess && l_b.alloc(n) == hipSuccess && d_inv.alloc(n) == hipSuccess && d_lab.alloc(n) == hipSuccess &&
            d_tmp.alloc(std::max<size_t>(tb, 16)) == hipSuccess) {
            HIP_TRY(ctx, mg::dense_cluster_rows(t->hashes, t->s, d_cnt, (uint32_t)n, d_tmp, tb, k_a, k_b, r_a, r_b, l_a, l_b, d_inv, d_lab, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(inv.data(), d_inv, n * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(lab_sorted.data(), d_lab, n * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            bool identity = true;
            for (uint64_t a = 0; a < n && identity; a++) identity = inv[a] == a;
            if (!identity) {
                void *pi = nullptr, *ph = nullptr;
                if (ctx_malloc(ctx, &pi, n * 4) != hipSuccess || ctx_malloc(ctx, &ph, std::max<uint64_t>(n * t->s, 1) * 8) != hipSuccess) {
                    (void)hipGetLastError();
                    ctx_free(ctx, pi);
                    lab_sorted.clear();                     // no memory for the copy: the table's own order
                } else {
                    sp->inv = static_cast<uint32_t *>(pi);
                    sp->phashes = static_cast<uint64_t *>(ph);
                    HIP_TRY(ctx, hipMemcpyAsync(sp->inv, d_inv, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
                    HIP_TRY(ctx, mg::launch_dense_gather_rows(t->hashes, t->s, sp->inv, (uint32_t)n, sp->phashes, ctx->stream));
                    H = sp->phashes;
                    std::vector<uint32_t> c2(n);
                    last_p.resize(n);
                    for (uint64_t a = 0; a < n; a++) { c2[a] = cnt_true[inv[a]]; last_p[a] = t->last[inv[a]]; }
                    cnt_true.swap(c2);
                    HIP_TRY(ctx, hipMemcpyAsync(d_cnt, cnt_true.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
                    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // (c2, the old counts, leaves scope; the buffers of this block go back to the pool)
                }
            }
        } else {
            (void)hipGetLastError();
            lab_sorted.clear();
        }
    }
    const std::vector<uint64_t> &lastv = last_p.empty() ? t->last : last_p;
    // which neighbouring rows are near-copies of each other (dense groups, compare_dense.hip): a sample test per row,
    // read back with the copy suspects below
    std::vector<uint8_t> link;
    bool want_dense = true;
    if (const char *e = ctx_opt(ctx, "MASHGPU_COMPARE_DENSE")) want_dense = atoi(e) != 0;
    DevBuf<uint8_t> d_link(ctx);
    if (want_dense && n >= 8 && s <= 16384 && !lab_sorted.empty()) {
        link.assign(n, 0);                                  // clustered variant: neighbours with the same label
        for (uint64_t a = 1; a < n; a++) link[a] = (lab_sorted[a] == lab_sorted[a - 1] && cnt_true[a] && cnt_true[a - 1]) ? 1 : 0;
    } else if (want_dense && n >= 8 && s <= 16384 && d_link.alloc(n) == hipSuccess) {      // (u16 counters of the extras, one bit a flag)
        link.resize(n);
        HIP_TRY(ctx, mg::launch_dense_neighbors(H, t->s, d_cnt, (uint32_t)n, d_link, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(link.data(), d_link, n, hipMemcpyDeviceToHost, ctx->stream));
        if (ctx_opt(ctx, "MASHGPU_SPARSE_NO_DEDUP")) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    } else {
        (void)hipGetLastError();
    }
    if (!ctx_opt(ctx, "MASHGPU_SPARSE_NO_DEDUP")) {
        DevBuf<unsigned long long> d_dig(ctx), d_dig_sorted(ctx);
        DevBuf<uint32_t> d_rows_sorted(ctx), d_flags(ctx), d_nflag(ctx);
        DevBuf<unsigned char> d_tmp(ctx);
        const size_t tb = mg::sparse_dup_temp_bytes((uint32_t)n);
        if (d_dig.alloc(n) != hipSuccess || d_dig_sorted.alloc(n) != hipSuccess || d_rows_sorted.alloc(n) != hipSuccess ||
            d_flags.alloc(n) != hipSuccess || d_nflag.alloc(1) != hipSuccess || d_tmp.alloc(std::max<size_t>(tb, 16)) != hipSuccess) {
            (void)hipGetLastError();
            return unusable("no device memory for the index");
        }
        uint32_t nflag = 0;
        HIP_TRY(ctx, mg::launch_sparse_row_digest(H, t->s, d_cnt, (uint32_t)n, d_dig, ctx->stream));
        HIP_TRY(ctx, mg::launch_sparse_dup_suspects(d_dig, d_cnt, (uint32_t)n, d_tmp, tb, d_dig_sorted, d_rows_sorted, d_flags, d_nflag, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&nflag, d_nflag, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (nflag) {
            std::vector<uint32_t> rows_sorted(n), flags(n);
            HIP_TRY(ctx, hipMemcpyAsync(rows_sorted.data(), d_rows_sorted, n * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(flags.data(), d_flags, n * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            std::vector<uint2> pairs;                      // {row, first row of its run of equal digests and lengths}
            pairs.reserve(nflag);
            for (uint64_t k = 1, g0 = 0; k < n; k++) {
                if (flags[k]) pairs.push_back(make_uint2(rows_sorted[k], rows_sorted[g0]));
                else g0 = k;
            }
            DevBuf<uint2> d_pairs(ctx);
            DevBuf<uint32_t> d_eq(ctx);
            if (d_pairs.alloc(pairs.size()) != hipSuccess || d_eq.alloc(pairs.size()) != hipSuccess) { (void)hipGetLastError(); return unusable("no device memory for the index"); }
            std::vector<uint32_t> eq(pairs.size());
            HIP_TRY(ctx, hipMemcpyAsync(d_pairs, pairs.data(), pairs.size() * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, mg::launch_sparse_row_equal(H, t->s, d_cnt, d_pairs, (uint32_t)pairs.size(), d_eq, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(eq.data(), d_eq, pairs.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            for (size_t k = 0; k < pairs.size(); k++)
                if (eq[k]) {
                    if (rep.empty()) { rep.resize(n); for (uint64_t i = 0; i < n; i++) rep[i] = (uint32_t)i; }
                    rep[pairs[k].x] = pairs[k].y;
                    sp->copies++;
                }
        }
    }
    std::vector<uint32_t> cls_of, cls_off, cls_rows, cls_first;
    if (sp->copies) {
        cls_of.assign(n, 0xFFFFFFFFu);
        std::vector<uint32_t> size(n, 0);
        for (uint64_t i = 0; i < n; i++) size[rep[i]]++;
        uint32_t ncls = 0, tot = 0;
        for (uint64_t i = 0; i < n; i++)
            if (size[i] >= 2) { cls_of[i] = ncls++; cls_off.push_back(tot); tot += size[i]; }
        cls_off.push_back(tot);
        cls_rows.resize(tot);
        std::vector<uint32_t> fillp(cls_off.begin(), cls_off.end() - 1);
        for (uint64_t i = 0; i < n; i++) {                 // ascending rows inside a class
            const uint32_t k = cls_of[rep[i]];
            if (k != 0xFFFFFFFFu) cls_rows[fillp[k]++] = (uint32_t)i;
        }
        cls_first.resize(tot);
        for (uint32_t k = 0; k < ncls; k++) {
            const uint64_t m = cls_off[k + 1] - cls_off[k];
            sp->cls_pairs += m * (m - 1) / 2;
            for (uint32_t u = cls_off[k]; u < cls_off[k + 1]; u++) cls_first[u] = cls_off[k];
        }
        sp->cls_members = tot;
        if (ncls == 1 && tot == n && cnt_true[0] > 0) sp->one_class = (uint32_t)cnt_true[0];
    }
    uint64_t E64 = 0, maxv = 0;
    sp->off_host.resize(n + 1);
    for (uint64_t i = 0; i < n; i++) {
        sp->off_host[i] = (uint32_t)E64;
        const uint64_t c = cnt_true[i];
        if (rep.empty() || rep[i] == i) E64 += c;           // copies stay out of the index
        if (E64 >= (1ull << 31)) return unusable("2^31 entries or more");
        if (c) {
            // a real hash equal to the padding value would sort among the padding: keep the tile engine
            if (lastv[i] == MG_HASH_PAD) return unusable("a hash equals the padding value");
            maxv = std::max(maxv, lastv[i]);
            if (c < s) sp->short_rows_host.push_back((uint32_t)i);
        } else {
            sp->short_rows_host.push_back((uint32_t)i);
            sp->has_empty = true;
        }
    }
    sp->off_host[n] = (uint32_t)E64;
    if (E64 == 0) return unusable("no hashes");
    const uint32_t E = (uint32_t)E64;
    sp->E = E;
    const uint32_t end_bit = (uint32_t)(64 - __builtin_clzll(maxv | 1ull));
    // transient buffers (back to the pool at the end of this function, in stream order)
    const uint32_t sort_begin_bit = mg::sparse_sort_begin_bit(E, end_bit, ctx_opt(ctx, "MASHGPU_SPARSE_SORT_BITS"), ctx_opt(ctx, "MASHGPU_SPARSE_SORT_ALL_BITS") != nullptr);
    const size_t temp_bytes = std::max(mg::sparse_sort_temp_bytes(E, end_bit, sort_begin_bit),
                                       std::max(mg::sparse_order_temp_bytes((uint32_t)n), mg::sparse_order_slice_temp_bytes((uint32_t)n)));
    DevBuf<unsigned char> temp(ctx);
    DevBuf<uint64_t> keys_a(ctx);
    DevBuf<uint32_t> idx_a(ctx), idx_sorted(ctx), gs_of(ctx);
    DevBuf<unsigned long long> key64_a(ctx), key64_b(ctx);
    struct Stat { unsigned long long shared; uint32_t max_group, groups, bad, tie_overflow; } h_stat = {0, 0, 0, 0, 0};
    DevBuf<Stat> d_stat(ctx);
    DevBuf<unsigned char> d_slots(ctx), d_ties(ctx);
    const bool want_order = !ctx_opt(ctx, "MASHGPU_SPARSE_NO_ORDER");
    bool ok = temp.alloc(std::max<size_t>(temp_bytes, 16)) == hipSuccess && keys_a.alloc(E) == hipSuccess && idx_a.alloc(E) == hipSuccess &&
              idx_sorted.alloc(E) == hipSuccess && gs_of.alloc(E) == hipSuccess && d_stat.alloc(1) == hipSuccess &&
              d_slots.alloc(mg::sparse_stat_scratch_bytes()) == hipSuccess && d_ties.alloc(mg::sparse_tie_scratch_bytes()) == hipSuccess &&
              (!want_order || (key64_a.alloc(n) == hipSuccess && key64_b.alloc(n) == hipSuccess));
    // retained buffers
    auto take = [&](auto **p, size_t count) {
        void *q = nullptr;
        if (ctx_malloc(ctx, &q, std::max<size_t>(count, 1) * sizeof(**p)) != hipSuccess) return false;
        *p = static_cast<std::remove_reference_t<decltype(*p)>>(q);
        return true;
    };
    ok = ok && take(&sp->off, n + 1) && take(&sp->keys_sorted, E) && take(&sp->gend, E) && take(&sp->sorted_rows, E) &&
         take(&sp->code_img, (size_t)n * sp->rs + 64) && take(&sp->pos_img, (size_t)n * sp->rs) && take(&sp->counters, 4) &&
         (!want_order || take(&sp->order, n));
    const size_t nshort = sp->short_rows_host.size();
    std::vector<uint32_t> short_cnt(nshort);
    for (size_t k = 0; k < nshort; k++) short_cnt[k] = cnt_true[sp->short_rows_host[k]];
    if (ok && nshort) ok = take(&sp->short_rows, nshort) && take(&sp->short_cnt, nshort);
    if (ok && sp->copies)
        ok = take(&sp->rep, n) && take(&sp->cls_of, n) && take(&sp->cls_off, cls_off.size()) && take(&sp->cls_rows, cls_rows.size()) &&
             take(&sp->cls_first, cls_first.size());
    hipError_t e = hipSuccess;
    if (ok && sp->copies) {
        e = hipMemcpyAsync(sp->rep, rep.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(sp->cls_of, cls_of.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(sp->cls_off, cls_off.data(), cls_off.size() * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(sp->cls_rows, cls_rows.data(), cls_rows.size() * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(sp->cls_first, cls_first.data(), cls_first.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    }
    if (ok && e == hipSuccess) {
        e = hipMemcpyAsync(sp->off, sp->off_host.data(), (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && nshort) e = hipMemcpyAsync(sp->short_rows, sp->short_rows_host.data(), nshort * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && nshort) e = hipMemcpyAsync(sp->short_cnt, short_cnt.data(), nshort * 4, hipMemcpyHostToDevice, ctx->stream);
        // (the sort looks at the values' leading bits only and repairs the few ties; a table that defeats that is sorted again, on every bit)
        for (uint32_t begin_bit = sort_begin_bit;; begin_bit = 0) {
            if (e == hipSuccess) e = hipMemsetAsync(d_stat, 0, sizeof(Stat), ctx->stream);
            if (e == hipSuccess)
                e = mg::sparse_build_index(H, t->s, sp->off, (uint32_t)n, E, sp->rs, end_bit, temp, temp_bytes, keys_a, idx_a,
                                           sp->keys_sorted, idx_sorted, /*head=*/idx_a, gs_of, sp->sorted_rows, sp->gend, sp->code_img, sp->pos_img,
                                           d_slots, begin_bit, d_ties, &d_stat.p->shared, &d_stat.p->max_group, &d_stat.p->groups, &d_stat.p->bad,
                                           &d_stat.p->tie_overflow, ctx->stream);
            // visiting order of the rows: by the run of their first shared value, larger rows first inside a run
            if (e == hipSuccess && want_order)
                e = mg::launch_sparse_row_order(sp->off, sp->code_img, sp->gend, sp->rep, (uint32_t)n, sp->rs, temp, temp_bytes, key64_a, key64_b,
                                                sp->order, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(&h_stat, d_stat, sizeof(Stat), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess || !h_stat.tie_overflow || begin_bit == 0) break;
        }
    }
    auto drop = [&]() {
        for (void **q : {(void **)&sp->off, (void **)&sp->keys_sorted, (void **)&sp->gend, (void **)&sp->sorted_rows, (void **)&sp->code_img,
                         (void **)&sp->pos_img, (void **)&sp->short_rows, (void **)&sp->short_cnt, (void **)&sp->counters, (void **)&sp->order,
                         (void **)&sp->rep, (void **)&sp->cls_of, (void **)&sp->cls_off, (void **)&sp->cls_rows, (void **)&sp->cls_first})
            if (*q) { ctx_free(ctx, *q); *q = nullptr; }
    };
    if (!ok) { (void)hipGetLastError(); drop(); return unusable("no device memory for the index"); }
    if (e != hipSuccess) { drop(); return fail(ctx, MG_ERR_HIP, std::string("compare (index build): ") + hipGetErrorString(e)); }
    if (h_stat.bad) { drop(); return unusable("sort order inside a value not by row"); }
    sp->G = h_stat.groups;
    sp->shared = h_stat.shared;
    sp->max_group = h_stat.max_group;
    sp->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    sp->usable = true;
    // ---- dense groups: runs of at least 8 consecutive rows linked to their predecessors.  Their universes come from the
    // index just built (gs_of is still alive), groups without one (or with one too large for a tile's LDS) are dropped,
    // the rest are encoded and the index's runs clipped for their rows.  Any failure here leaves the index as it is.
    if (!link.empty() && sp->copies == 0) {
        std::vector<mg::DenseGroup> cand_groups;
        for (uint64_t i = 1; i < n;) {
            if (!link[i]) { i++; continue; }
            uint64_t j = i;
            while (j < n && link[j]) j++;                    // rows [i - 1, j) form a chain
            if (j - (i - 1) >= 8) {
                mg::DenseGroup g{};
                g.g0 = (uint32_t)(i - 1); g.g1 = (uint32_t)j;
                cand_groups.push_back(g);
            }
            i = j + 1;
        }
        auto t_dense = std::chrono::steady_clock::now();
        while (!cand_groups.empty()) {                       // (a block to leave with `break`)
            const uint32_t ng = (uint32_t)cand_groups.size();
            std::vector<uint32_t> grp_of(n, 0xFFFFFFFFu);
            for (uint32_t g = 0; g < ng; g++)
                for (uint32_t r = cand_groups[g].g0; r < cand_groups[g].g1; r++) grp_of[r] = g;
            DevBuf<mg::DenseGroup> d_groups(ctx);
            DevBuf<uint32_t> d_grp_of(ctx), d_val(ctx), d_nlead(ctx), d_us(ctx), d_ue(ctx);
            DevBuf<unsigned long long> d_key(ctx), d_key2(ctx);
            DevBuf<unsigned char> d_tmp(ctx);
            if (d_groups.alloc(ng) != hipSuccess || d_grp_of.alloc(n) != hipSuccess || d_nlead.alloc(2) != hipSuccess || d_us.alloc(ng) != hipSuccess ||
                d_ue.alloc(ng) != hipSuccess) { (void)hipGetLastError(); break; }
            // leaders: one list entry each, appended to one of a thousand lists (room for a quarter of the entries in all, evenly;
            // a table with more, or with lists that fill unevenly, is searched a second time with the room the first pass asked for)
            const uint32_t L = mg::dense_sublists();
            DevBuf<unsigned long long> d_keyj(ctx);
            DevBuf<uint32_t> d_valj(ctx), d_cnt_sub(ctx), d_off_sub(ctx);
            uint32_t tot[2] = {0, 0}, cap_sub = std::max<uint32_t>(E / 4u / L + 64u, 256u);
            hipError_t e2 = hipMemcpyAsync(d_groups, cand_groups.data(), ng * sizeof(mg::DenseGroup), hipMemcpyHostToDevice, ctx->stream);
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(d_grp_of, grp_of.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
            if (e2 == hipSuccess && (d_cnt_sub.alloc(L) != hipSuccess || d_off_sub.alloc(L) != hipSuccess)) { (void)hipGetLastError(); e2 = hipErrorOutOfMemory; }
            for (int attempt = 0; e2 == hipSuccess; attempt++) {
                for (void **q : {(void **)&d_key.p, (void **)&d_val.p, (void **)&d_keyj.p, (void **)&d_valj.p})
                    if (*q) { ctx_free(ctx, *q); *q = nullptr; }
                const uint64_t room = (uint64_t)L * cap_sub;
                if (d_key.alloc(room) != hipSuccess || d_val.alloc(room) != hipSuccess || d_keyj.alloc(room) != hipSuccess || d_valj.alloc(room) != hipSuccess) {
                    (void)hipGetLastError();
                    e2 = hipErrorOutOfMemory;
                    break;
                }
                e2 = mg::dense_find_leaders(sp->sorted_rows, gs_of, sp->gend, d_grp_of, d_groups, E, d_key, d_val, cap_sub, d_keyj, d_valj, d_cnt_sub, d_off_sub,
                                            d_nlead, ctx->stream);
                if (e2 == hipSuccess) e2 = hipMemcpyAsync(tot, d_nlead, 8, hipMemcpyDeviceToHost, ctx->stream);
                if (e2 == hipSuccess) e2 = hipStreamSynchronize(ctx->stream);
                if (e2 != hipSuccess || tot[1] <= cap_sub || attempt >= 1) break;
                cap_sub = tot[1];
            }
            const uint32_t nlead = tot[0];
            if (e2 != hipSuccess || nlead == 0 || tot[1] > cap_sub) { (void)hipGetLastError(); break; }
            uint32_t gbits = 1;
            while ((1u << gbits) < ng) gbits++;
            std::vector<uint32_t> us(ng, 0), ue(ng, 0);
            void *ul = nullptr, *up = nullptr;
            const size_t tb = mg::dense_universe_temp_bytes(nlead);
            if (d_key2.alloc(nlead) != hipSuccess || d_tmp.alloc(std::max<size_t>(tb, 16)) != hipSuccess ||
                ctx_malloc(ctx, &ul, (size_t)nlead * 4) != hipSuccess || ctx_malloc(ctx, &up, (size_t)nlead * 4) != hipSuccess) {
                (void)hipGetLastError();
                ctx_free(ctx, ul);
                break;
            }
            sp->ulist = static_cast<uint32_t *>(ul);
            sp->upos = static_cast<uint32_t *>(up);
            e2 = hipMemsetAsync(d_us, 0, ng * 4, ctx->stream);
            if (e2 == hipSuccess) e2 = hipMemsetAsync(d_ue, 0, ng * 4, ctx->stream);
            if (e2 == hipSuccess)
                e2 = mg::dense_sort_universes(d_keyj, d_valj, nlead, d_tmp, tb, d_key2, sp->ulist, sp->upos, d_us, d_ue, gbits, ctx->stream);
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(us.data(), d_us, ng * 4, hipMemcpyDeviceToHost, ctx->stream);
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(ue.data(), d_ue, ng * 4, hipMemcpyDeviceToHost, ctx->stream);
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(ctx->stream);
            if (e2 != hipSuccess) { (void)hipGetLastError(); break; }
            // the groups that stay: a universe of at least 32 values (else the rows are not near-copies and the pairs are
            // cheap elsewhere) and at most as many words as a tile's rows fit the LDS with
            const uint32_t kMaxWords = mg::dense_max_words();
            std::fill(grp_of.begin(), grp_of.end(), 0xFFFFFFFFu);
            uint32_t xrows = 0, wmax = 0;
            uint64_t words = 0;
            for (uint32_t g = 0; g < ng; g++) {
                mg::DenseGroup G = cand_groups[g];
                G.u = ue[g] > us[g] ? ue[g] - us[g] : 0u;
                G.ustart = us[g];
                G.W = (G.u >> 6) + 1u;
                if (G.u < 32u || G.W > kMaxWords) continue;
                G.xrow0 = xrows;
                G.data_off = words;
                const uint64_t m = G.g1 - G.g0;
                xrows += (uint32_t)m;
                words += ((m + 127) / 128) * (128ull * G.W + 32ull * (G.W + 1u));
                wmax = std::max(wmax, G.W);
                for (uint32_t r = G.g0; r < G.g1; r++) grp_of[r] = (uint32_t)sp->dgroups_host.size();
                sp->dgroups_host.push_back(G);
            }
            if (sp->dgroups_host.empty()) break;
            sp->dn_wmax = wmax;
            sp->dn_xs = ((s + 7u) & ~7u) + 8u;
            void *p1 = nullptr, *p2 = nullptr, *p3 = nullptr, *p4 = nullptr, *p5 = nullptr;
            if (ctx_malloc(ctx, &p1, sp->dgroups_host.size() * sizeof(mg::DenseGroup)) != hipSuccess || ctx_malloc(ctx, &p2, n * 4) != hipSuccess ||
                ctx_malloc(ctx, &p3, words * 8) != hipSuccess || ctx_malloc(ctx, &p4, (size_t)xrows * sp->dn_xs * 2) != hipSuccess ||
                ctx_malloc(ctx, &p5, (size_t)xrows * wmax * 24) != hipSuccess) {
                (void)hipGetLastError();
                for (void *q : {p1, p2, p3, p4, p5}) ctx_free(ctx, q);
                sp->dgroups_host.clear();
                break;
            }
            sp->dgroups = static_cast<mg::DenseGroup *>(p1);
            sp->grp_of = static_cast<uint32_t *>(p2);
            sp->gdata = static_cast<unsigned long long *>(p3);
            sp->ext = static_cast<uint16_t *>(p4);
            sp->xm = static_cast<unsigned long long *>(p5);
            e2 = hipMemcpyAsync(sp->dgroups, sp->dgroups_host.data(), sp->dgroups_host.size() * sizeof(mg::DenseGroup), hipMemcpyHostToDevice, ctx->stream);
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(sp->grp_of, grp_of.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
            // masks, extras, and the index's runs clipped for the rows of the groups: discovery sees the partners outside only
            if (e2 == hipSuccess)
                e2 = mg::launch_dense_encode(sp->off, sp->code_img, sp->pos_img, sp->rs, sp->grp_of, sp->dgroups, sp->ulist, sp->upos, sp->gdata, sp->xm,
                                             sp->ext, sp->dn_xs, (uint32_t)n, wmax, ctx->stream);
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(ctx->stream);   // (grp_of, the host vector, is read by the copy above)
            if (e2 != hipSuccess) {
                // the runs may be half clipped: this index is not to be used
                drop();
                for (void **q : {(void **)&sp->dgroups, (void **)&sp->grp_of, (void **)&sp->gdata, (void **)&sp->ext, (void **)&sp->xm, (void **)&sp->ulist,
                                 (void **)&sp->upos})
                    if (*q) { ctx_free(ctx, *q); *q = nullptr; }
                sp->dgroups_host.clear();
                sp->usable = false;
                return fail(ctx, MG_ERR_HIP, std::string("compare (index build, dense groups): ") + hipGetErrorString(e2));
            }
            sp->dn_lists = ctx_opt(ctx, "MASHGPU_DENSE_LISTS") != nullptr;          // (test knob: every word resolved from the lists)
            break;
        }
        if (sp->dgroups_host.empty() && sp->ulist) { ctx_free(ctx, sp->ulist); ctx_free(ctx, sp->upos); sp->ulist = sp->upos = nullptr; }
        sp->build_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dense).count();
        if (ctx_opt(ctx, "MASHGPU_SPARSE_DBG")) {
            uint64_t rows_in = 0;
            for (auto &G : sp->dgroups_host) rows_in += G.g1 - G.g0;
            fprintf(stderr, "compare dense: %zu chains of related rows, %zu groups kept (%llu rows, widest universe %u words)\n", cand_groups.size(),
                    sp->dgroups_host.size(), (unsigned long long)rows_in, sp->dn_wmax);
        }
    }
    if (ctx_opt(ctx, "MASHGPU_SPARSE_DBG"))
        fprintf(stderr, "compare sparse: index of %llu rows (%llu copies of earlier rows), s %u: %u entries, %u distinct, shared %llu, largest run %u, %.2f ms\n",
                (unsigned long long)n, (unsigned long long)sp->copies, s, E, sp->G, (unsigned long long)sp->shared, sp->max_group, sp->build_ms);
    return MG_OK;
}

// Pairs of the job and the engine choice.  `force`: MASHGPU_COMPARE_KERNEL=sparse.  *handled = false:
// the caller goes on to the tile engine (table outside the index's reach, or the job is one the
// tile engine does faster: nearly every pair shares a few hashes -- a candidate costs a merge of
// ~2 s steps here, an unrelated pair there costs 1/30 of that).
// `job` != nullptr: only the candidates are wanted (thresholded calls: a pair that shares no hash has
// distance 1 and p-value 1, no filter lets it through) -- discover + merge run, the output is neither
// filled nor touched, and the job describes the candidate list {row, col} / {common, denom} left in
// the index's buffers.  Refused (handled = false) where pairs outside the list could survive.
// Order of a pass: DISCOVER first (it also counts: the candidates K and the shared hashes I of the job,
// which decide the engine the first time a job is seen -- there is no separate counting pass), then fill,
// merge, scatter.
struct SparseJob { mg::SparseArgs args; uint64_t cand = 0; mg_table::Sparse *ix = nullptr; };

static int run_compare_sparse(mg_ctx *ctx, const mg_table *rows, const mg_table *cols, uint64_t row_begin, uint64_t row_end,
                              bool triangle, uint32_t s, mg_counts *out_dev, bool force, bool *handled, SparseJob *job = nullptr)
{
    *handled = false;
    const uint64_t nrows = row_end - row_begin;
    const uint64_t pairs = triangle ? (row_end * (row_end - 1) / 2 - (row_begin ? row_begin * (row_begin - 1) / 2 : 0)) : nrows * cols->n;
    if (pairs == 0) return MG_OK;
    if (!force && pairs < 4000000ull) return MG_OK;        // small jobs: one tile launch beats an index
    if (const char *e = ctx_opt(ctx, "MASHGPU_COMPARE_SPARSE")) { if (atoi(e) == 0 && !force) return MG_OK; }
    if (nrows >= (1ull << 31) || cols->n >= (1ull << 31)) return MG_OK;
    if (!mg::sparse_discover_supported((uint32_t)(triangle ? row_end : cols->n))) return MG_OK;
    mg_table::Sparse *ix = nullptr;
    // the plain full triangle takes the CLUSTERED variant of the index (built on the table with related rows next to each
    // other: dense groups whatever the order of the collection); row ranges, rect and list jobs address table rows
    bool clustered = triangle && !job && row_begin == 0 && row_end == cols->n;
    if (const char *e = ctx_opt(ctx, "MASHGPU_COMPARE_CLUSTER")) clustered = clustered && atoi(e) != 0;
    int rc = table_sparse_index(ctx, cols, s, clustered, &ix);
    if (rc != MG_OK) return rc;
    if (!ix->usable && clustered) {                         // (whatever stopped it may not stop the plain variant)
        clustered = false;
        rc = table_sparse_index(ctx, cols, s, false, &ix);
        if (rc != MG_OK) return rc;
    }
    if (!ix->usable) return MG_OK;
    // (list mode: two copies of one sketch are a pair at distance 0 that is no candidate, two EMPTY sketches
    //  likewise -- such tables take the matrix path; so do tables with dense groups, whose inner pairs are in no list)
    if (job && (ix->copies || ix->has_empty || !ix->dgroups_host.empty())) return MG_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    // ---- row side ----
    mg::SparseArgs a;
    a.sorted_rows = ix->sorted_rows;
    a.col_img = ix->code_img;
    a.col_cnt_off = ix->off;
    a.rs_col = ix->rs;
    a.ncols = (uint32_t)cols->n;
    a.triangle = triangle ? 1u : 0u;
    a.s = s;
    a.out = reinterpret_cast<uint2 *>(out_dev);
    a.counters = ix->counters;
    a.rep = ix->rep;
    a.cls_of = ix->cls_of;
    a.cls_off = ix->cls_off;
    a.cls_rows = ix->cls_rows;
    a.gend = ix->gend;
    a.inv = ix->inv;
    a.res = nullptr;
    a.seg_base = nullptr;
    a.seg_cnt = nullptr;
    a.chunk_inc = nullptr;
    DevBuf<uint32_t> q_off(ctx), q_img(ctx), q_lo(ctx), q_hi(ctx), q_short(ctx), q_short_cnt(ctx);
    std::vector<uint32_t> qshort_h, qshort_cnt_h;
    const uint32_t *short_rows_dev = nullptr, *short_rcnt_dev = nullptr;
    uint32_t nshort_rows = 0;
    if (triangle) {
        a.lo_img = ix->code_img;
        a.hi_img = ix->pos_img;
        a.lo_shift = 1;
        a.off = ix->off;
        a.row_img = ix->code_img;
        a.rs_row = ix->rs;
        a.row_begin = (uint32_t)row_begin;
        a.row_end = (uint32_t)row_end;
        a.out_base = row_begin ? row_begin * (row_begin - 1) / 2 : 0;
        // short rows inside [row_begin, row_end): a slice of the table's ascending list
        const auto &sr = ix->short_rows_host;
        const size_t k0 = std::lower_bound(sr.begin(), sr.end(), (uint32_t)row_begin) - sr.begin();
        const size_t k1 = std::lower_bound(sr.begin(), sr.end(), (uint32_t)row_end) - sr.begin();
        nshort_rows = (uint32_t)(k1 - k0);
        short_rows_dev = ix->short_rows ? ix->short_rows + k0 : nullptr;
        short_rcnt_dev = ix->short_cnt ? ix->short_cnt + k0 : nullptr;
    } else {
        // queries [row_begin, row_end) of `rows`, located in the reference table's index
        rc = table_classes(ctx, rows);
        if (rc != MG_OK) return rc;
        std::vector<uint32_t> qoff(nrows + 1);
        uint64_t Eq = 0;
        for (uint64_t q = 0; q < nrows; q++) {
            qoff[q] = (uint32_t)Eq;
            const uint64_t c = std::min<uint64_t>(std::min<uint64_t>(rows->nh[row_begin + q], rows->s), s);
            Eq += c;
            if (Eq >= (1ull << 31)) return MG_OK;
            if (c < s) { qshort_h.push_back((uint32_t)q); qshort_cnt_h.push_back((uint32_t)c); }
            if (c == 0 && job) return MG_OK;
            if (c && rows->last[row_begin + q] == MG_HASH_PAD) return MG_OK;
        }
        qoff[nrows] = (uint32_t)Eq;
        const uint32_t rsq = ix->rs;
        if (nrows * rsq >= (1ull << 32)) return MG_OK;
        if (q_off.alloc(nrows + 1) != hipSuccess || q_img.alloc(nrows * rsq) != hipSuccess || q_lo.alloc(nrows * rsq) != hipSuccess ||
            q_hi.alloc(nrows * rsq) != hipSuccess || q_short.alloc(std::max<size_t>(qshort_h.size(), 1)) != hipSuccess ||
            q_short_cnt.alloc(std::max<size_t>(qshort_h.size(), 1)) != hipSuccess) {
            (void)hipGetLastError();
            return MG_OK;                                   // no memory for the query side: tile engine
        }
        HIP_TRY(ctx, hipMemcpyAsync(q_off, qoff.data(), (nrows + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        if (!qshort_h.empty()) {
            HIP_TRY(ctx, hipMemcpyAsync(q_short, qshort_h.data(), qshort_h.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(q_short_cnt, qshort_cnt_h.data(), qshort_h.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));    // the host vectors go out of scope below
        HIP_TRY(ctx, mg::launch_sparse_locate(rows->hashes, rows->s, q_off, (uint32_t)row_begin, (uint32_t)nrows, ix->keys_sorted, ix->gend,
                                              ix->E, rsq, q_lo, q_hi, q_img, ctx->stream));
        a.lo_img = q_lo;
        a.hi_img = q_hi;
        a.lo_shift = 0;
        a.off = q_off;
        a.row_img = q_img;
        a.rs_row = rsq;
        a.row_begin = 0;
        a.row_end = (uint32_t)nrows;
        a.out_base = 0;
        nshort_rows = (uint32_t)qshort_h.size();
        short_rows_dev = q_short;
        short_rcnt_dev = q_short_cnt;
    }

    // ---- the plan of a (rows, range) job: its slice of the visiting order; candidates, shared hashes and the engine
    // choice are learned from the first discover launch (rect: the query table may change between calls, so every
    // call is a first call)
    mg_table::Sparse::Plan *plan = nullptr;
    if (triangle)
        for (auto &pl : ix->plans)
            if (pl.rows == (const void *)rows && pl.rb == row_begin && pl.re == row_end && pl.triangle == triangle) plan = &pl;
    mg_table::Sparse::Plan fresh;
    const bool first = plan == nullptr;
    if (first) {
        fresh.rows = rows; fresh.rb = row_begin; fresh.re = row_end; fresh.triangle = triangle;
        fresh.cand = 0; fresh.shared = 0; fresh.use = true; fresh.order = nullptr;
        fresh.dtiles = nullptr; fresh.ndtiles = 0; fresh.dtile_rows = 32; fresh.dense_pairs = 0;
        if (triangle && !ix->dgroups_host.empty()) {
            // tiles of the dense groups' inner pairs: 32 or 8 rows (aligned to the group's first row) x a block of 128 columns
            uint64_t wave_rows = 0;                          // rows x column blocks x two waves: the work there is to hand out
            for (const mg::DenseGroup &G : ix->dgroups_host) {
                const uint64_t m = G.g1 - G.g0;
                wave_rows += m * ((m + 127) / 128);          // (about half of it below the diagonal)
            }
            const uint32_t R = mg::dense_rows_per_tile(wave_rows);
            fresh.dtile_rows = R;
            std::vector<mg::DenseTile> tiles;
            for (uint32_t g = 0; g < ix->dgroups_host.size(); g++) {
                const mg::DenseGroup &G = ix->dgroups_host[g];
                if (G.g1 <= row_begin || G.g0 >= row_end) continue;
                for (uint32_t row0 = G.g0; row0 < G.g1; row0 += R) {
                    const uint64_t a_lo = std::max<uint64_t>(std::max<uint64_t>(row0, row_begin), (uint64_t)G.g0 + 1), a_hi = std::min<uint64_t>(std::min<uint64_t>(row0 + R, G.g1), row_end);
                    if (a_lo >= a_hi) continue;
                    for (uint64_t a = a_lo; a < a_hi; a++) fresh.dense_pairs += a - G.g0;
                    const uint32_t cb_last = (uint32_t)((a_hi - 2 - G.g0) >> 7);         // the largest column is a_hi - 2
                    for (uint32_t cb = 0; cb <= cb_last; cb++) tiles.push_back({g, row0, cb});
                }
            }
            if (!tiles.empty()) {
                void *q = nullptr;
                if (ctx_malloc(ctx, &q, tiles.size() * sizeof(mg::DenseTile)) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, MG_ERR_NOMEM, "compare: no device memory for the dense tiles"); }
                hipError_t e = hipMemcpyAsync(q, tiles.data(), tiles.size() * sizeof(mg::DenseTile), hipMemcpyHostToDevice, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) { ctx_free(ctx, q); return fail(ctx, MG_ERR_HIP, std::string("compare (dense tiles): ") + hipGetErrorString(e)); }
                fresh.dtiles = static_cast<mg::DenseTile *>(q);
                fresh.ndtiles = (uint32_t)tiles.size();
            }
        }
        if (triangle && ix->order) {
            if (row_begin == 0 && row_end == cols->n) {
                fresh.order = nullptr;                      // the whole table: the index's own list
            } else {                                        // the rows of this job in visiting order
                void *q = nullptr, *tmp = nullptr, *cnt = nullptr;
                const size_t tb = mg::sparse_order_slice_temp_bytes((uint32_t)cols->n);
                if (ctx_malloc(ctx, &q, nrows * 4) == hipSuccess && ctx_malloc(ctx, &tmp, std::max<size_t>(tb, 16)) == hipSuccess &&
                    ctx_malloc(ctx, &cnt, 8) == hipSuccess &&
                    mg::launch_sparse_order_slice(ix->order, (uint32_t)cols->n, (uint32_t)row_begin, (uint32_t)row_end, tmp, tb,
                                                  static_cast<uint32_t *>(q), static_cast<uint32_t *>(cnt), ctx->stream) == hipSuccess) {
                    fresh.order = static_cast<uint32_t *>(q);
                    q = nullptr;
                } else {
                    (void)hipGetLastError();
                }
                ctx_free(ctx, q);
                ctx_free(ctx, tmp);
                ctx_free(ctx, cnt);
            }
        }
        plan = &fresh;
    }
    a.order = triangle && ix->order ? (plan->order ? plan->order : (row_begin == 0 && row_end == cols->n ? ix->order : nullptr)) : nullptr;

    // ---- lists of the job (grown on demand, kept with the index) ----
    auto ensure_lists = [&](uint64_t want_cand) -> int {
        if (want_cand > ix->cand_cap) {
            for (void **q : {(void **)&ix->cand, (void **)&ix->res})
                if (*q) { ctx_free(ctx, *q); *q = nullptr; }
            ix->cand_cap = 0;
            const uint64_t cap = want_cand + want_cand / 8 + 1024;
            void *c1 = nullptr, *c2 = nullptr;
            if (ctx_malloc(ctx, &c1, cap * sizeof(uint2)) != hipSuccess || ctx_malloc(ctx, &c2, cap * sizeof(uint2)) != hipSuccess) {
                (void)hipGetLastError();
                ctx_free(ctx, c1);
                return fail(ctx, MG_ERR_NOMEM, "compare: no device memory for the candidate list");
            }
            ix->cand = static_cast<uint2 *>(c1);
            ix->res = static_cast<uint2 *>(c2);
            ix->cand_cap = cap;
        }
        if (nrows > ix->seg_rows) {
            for (void **q : {(void **)&ix->seg_base, (void **)&ix->seg_cnt, (void **)&ix->chunks, (void **)&ix->chunk_inc, &ix->scan_temp})
                if (*q) { ctx_free(ctx, *q); *q = nullptr; }
            ix->seg_rows = 0;
            const uint64_t cap = nrows + nrows / 8 + 256;
            ix->scan_temp_bytes = mg::sparse_scan_temp_bytes((uint32_t)cap);
            void *p1 = nullptr, *p2 = nullptr, *p3 = nullptr, *p4 = nullptr, *p5 = nullptr;
            if (ctx_malloc(ctx, &p1, cap * 8) != hipSuccess || ctx_malloc(ctx, &p2, cap * 4) != hipSuccess ||
                ctx_malloc(ctx, &p3, cap * 4) != hipSuccess || ctx_malloc(ctx, &p4, cap * 4) != hipSuccess ||
                ctx_malloc(ctx, &p5, std::max<size_t>(ix->scan_temp_bytes, 16)) != hipSuccess) {
                (void)hipGetLastError();
                for (void *q : {p1, p2, p3, p4, p5}) ctx_free(ctx, q);
                return fail(ctx, MG_ERR_NOMEM, "compare: no device memory for the merge work list");
            }
            ix->seg_base = static_cast<unsigned long long *>(p1);
            ix->seg_cnt = static_cast<uint32_t *>(p2);
            ix->chunks = static_cast<uint32_t *>(p3);
            ix->chunk_inc = static_cast<uint32_t *>(p4);
            ix->scan_temp = p5;
            ix->seg_rows = cap;
        }
        return MG_OK;
    };
    if (!first && !force && !plan->use) return MG_OK;      // a job the tile engine was found to do faster
    // first sight of a job: room for one candidate per two index entries, at most 2^27 (2 GB of lists from the pool; C3
    // has one per twenty, the clade table one per two); a job that holds more is discovered twice, the second time
    // with the count the first one left
    uint64_t want = first ? std::max<uint64_t>(ix->cand_cap, std::min<uint64_t>(pairs, std::min<uint64_t>(std::max<uint64_t>((uint64_t)ix->E / 2, 1u << 16), 1ull << 27)))
                          : plan->cand;
    unsigned long long h[3] = {0, 0, 0};
    const bool nothing_to_find = triangle && ix->one_class != 0;      // nothing but copies of one sketch: every pair is inside the class
    for (int attempt = 0; !nothing_to_find; attempt++) {
        rc = ensure_lists(want);
        if (rc != MG_OK) { if (first && fresh.order) ctx_free(ctx, fresh.order); return rc; }
        a.cand = ix->cand;
        a.res = ix->res;
        a.cand_cap = ix->cand_cap;
        a.seg_base = ix->seg_base;
        a.seg_cnt = ix->seg_cnt;
        a.chunk_inc = ix->chunk_inc;
        HIP_TRY(ctx, hipMemsetAsync(ix->counters, 0, 4 * 8, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ix->seg_cnt, 0, nrows * 4, ctx->stream));
        prof_begin(ctx, ctx->prof_discover);
        hipError_t e = mg::launch_sparse_discover(a, false, ctx->stream);
        prof_end(ctx, ctx->prof_discover);
        if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("compare (discover): ") + hipGetErrorString(e));
        if (!first) break;                                  // a job seen before: its list has the size it needed then (checked at the end)
        HIP_TRY(ctx, hipMemcpyAsync(h, ix->counters, 24, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (h[2] == 0) break;
        if (attempt >= 1) { if (fresh.order) ctx_free(ctx, fresh.order); return fail(ctx, MG_ERR_INVALID, "compare: the table changed while it was compared"); }
        want = h[0];
    }
    if (first) {
        fresh.cand = h[0];
        fresh.shared = h[1];
        // seconds, one MI355X (measured: profiles/r03_sparse_phases.txt)
        const double np = (double)pairs;
        const double t_sparse = np * 8.0 / 4.5e12 + (double)fresh.shared * 2.0e-12 + (double)nrows * s * 4.0e-11 + (double)fresh.cand * 1.0e-9 + 2.0e-5 +
                                (triangle ? (double)ix->cls_pairs * 8.0 / 2.0e12 : 0.0) + (double)fresh.dense_pairs * 3.0e-11;
        const double dense_rate = np < 3.0e8 ? 8.0e9 : np < 2.0e9 ? 1.5e10 : 3.0e10;
        const double t_dense = np / dense_rate + (double)fresh.shared * 2.2e-12;
        fresh.use = t_sparse < t_dense;
        if (ctx_opt(ctx, "MASHGPU_SPARSE_DBG"))
            fprintf(stderr, "compare sparse: rows [%llu, %llu) %s: %llu pairs, %llu candidates, %llu shared hashes; model sparse %.3f ms, tiles %.3f ms\n",
                    (unsigned long long)row_begin, (unsigned long long)row_end, triangle ? "triangle" : "rect", (unsigned long long)pairs,
                    (unsigned long long)fresh.cand, (unsigned long long)fresh.shared, t_sparse * 1e3, t_dense * 1e3);
        if (triangle) {
            if (ix->plans.size() >= 64) {
                if (ix->plans.front().order) ctx_free(ctx, ix->plans.front().order);
                if (ix->plans.front().dtiles) ctx_free(ctx, ix->plans.front().dtiles);
                ix->plans.erase(ix->plans.begin());
            }
            ix->plans.push_back(fresh);
            plan = &ix->plans.back();
        }
    }
    if (!force && !plan->use) return MG_OK;

    // ---- fill.  The candidates' results are kept in list order and scattered into the output after it.
    // (Side by side with discover + merge on a second stream the fill was MEASURED to gain nothing -- discover's
    // loads queue behind 40 GB of writes, and a kernel that merely ends under the fill waits milliseconds for the
    // L2's write-back, profiles/r03_sparse_phases.json, r03_overlap_trace.txt -- so the phases run one after the other.)
    if (!job) {
        prof_begin(ctx, ctx->prof_fill);
        // a table of n copies of one sketch: the fill IS the answer, {c, c} in every slot, written once
        const bool all_copies = triangle && ix->one_class != 0;
        hipError_t e = all_copies ? mg::launch_sparse_fill_value(a.out, pairs, ix->one_class, ix->one_class, 16u, (uint32_t)ctx->cu_count, ctx->stream)
                                  : mg::launch_sparse_fill_value(a.out, pairs, 0u, s, 16u, (uint32_t)ctx->cu_count, ctx->stream);
        if (e == hipSuccess && nshort_rows && !ix->short_rows_host.empty() && !all_copies)      // (copies of a SHORT sketch are {c, c} too, not {0, 2c})
            e = mg::launch_sparse_fill_short(a.out, short_rows_dev, short_rcnt_dev, nshort_rows, ix->short_rows, ix->short_cnt,
                                             (uint32_t)ix->short_rows_host.size(), a.row_begin, a.ncols, a.triangle, a.out_base, s, a.inv, ctx->stream);
        // pairs of two copies of one sketch: {n, n} (after the fill)
        if (e == hipSuccess && triangle && ix->cls_members && !all_copies)
            e = mg::launch_sparse_class_pairs(a.out, ix->cls_rows, ix->cls_first, ix->off, ix->rep, ix->cls_members, a.row_begin, a.row_end,
                                              a.out_base, a.inv, ctx->stream);
        prof_end(ctx, ctx->prof_fill);
        if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("compare (fill): ") + hipGetErrorString(e));
        // the pairs inside the dense groups (over the fill; candidates never lie inside a group)
        if (plan->ndtiles) {
            prof_begin(ctx, ctx->prof_dense);
            e = mg::launch_dense_pairs(plan->dtiles, plan->ndtiles, plan->dtile_rows, ix->dgroups, ix->gdata, ix->xm, ix->dn_lists, ix->ext, ix->dn_xs, s, ix->dn_wmax,
                                       a.row_begin, a.row_end, a.out_base, a.inv, a.out, ctx->stream);
            prof_end(ctx, ctx->prof_dense);
            if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("compare (dense groups): ") + hipGetErrorString(e));
        }
    }
    *handled = true;
    if (job) { job->ix = ix; job->cand = plan->cand; job->args = a; }
    if (plan->cand == 0) return MG_OK;
    // ---- merge ----
    bool by_rows = mg::sparse_merge_rows_supported(a.rs_row);
    if (const char *ev = ctx_opt(ctx, "MASHGPU_SPARSE_MERGE")) by_rows = by_rows && strcmp(ev, "lanes") != 0;
    prof_begin(ctx, ctx->prof_merge);
    hipError_t e = hipSuccess;
    bool packed = false;
    // rows with few candidates each (a collection: C3 has 50 per row) share a work item; rows with hundreds (clades)
    // fill their own items and gain nothing from staging their neighbours (measured: 27.8 -> 33.4 ms on the clade table)
    bool pack = by_rows && plan->cand < 64ull * nrows;
    if (const char *ev = ctx_opt(ctx, "MASHGPU_SPARSE_MERGE_PACK")) pack = by_rows && atoi(ev) != 0;
    if (pack) e = mg::launch_sparse_merge_pack(a, plan->cand, ix->chunks, ix->scan_temp, ix->scan_temp_bytes, &packed, ctx->stream);
    if (!packed && e == hipSuccess)
        e = by_rows ? mg::launch_sparse_merge_rows(a, plan->cand, ix->chunks, ix->scan_temp, ix->scan_temp_bytes, ctx->stream)
                    : mg::launch_sparse_merge(a, plan->cand, (uint32_t)ctx->cu_count, ctx->stream);
    prof_end(ctx, ctx->prof_merge);
    if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("compare (merge): ") + hipGetErrorString(e));
    if (job) job->args = a;
    if (!job) {
        e = mg::launch_sparse_scatter(a, plan->cand, (uint32_t)ctx->cu_count, ctx->stream);
        if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("compare (scatter): ") + hipGetErrorString(e));
    }
    if (!first && !nothing_to_find && (!ctx->async || !triangle || job)) {
        // the candidate list was sized by the first pass over the same rows: an overflow or another count means the
        // tables changed under the cache (mg_table_wrap_dev's contract forbids it; mg_table_invalidate is the remedy)
        HIP_TRY(ctx, hipMemcpyAsync(h, ix->counters, 24, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (h[2] != 0 || h[0] != plan->cand) return fail(ctx, MG_ERR_INVALID, "compare: the table changed since its index was built (mg_table_invalidate)");
    } else if (!ctx->async || job) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MG_OK;
}

// Engine choice (MASHGPU_COMPARE_KERNEL forces one: sparse | merged | generic):
//   1. the inverted-index engine (compare_sparse.hip) when its counting pass says the job is sparse
//      enough -- nearly always for a collection;
//   2. the tile engine (compare_merged.hip): jobs below 4e6 pairs, jobs where nearly every pair shares a
//      few hashes, tables the index cannot take;
//   3. the generic kernel: sketch sizes the tile engine cannot window, and as the independent cross-check.
static int run_compare(mg_ctx *ctx, const mg_table *rows, const mg_table *cols, uint64_t row_begin,
                       uint64_t row_end, bool triangle, mg_counts *out_dev)
{
    if (row_end > rows->n) row_end = rows->n;
    if (row_begin >= row_end) return MG_OK;
    if (rows->n > 0xFFFFFFFFull || cols->n > 0xFFFFFFFFull) return fail(ctx, MG_ERR_INVALID, "compare: table too large");
    const uint64_t s64 = std::min(rows->s, cols->s);       // CommandDistance.cpp:313-315
    if (s64 > 0xFFFFFFFFull) return fail(ctx, MG_ERR_INVALID, "compare: sketch size too large");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mg::CompareArgs a;
    a.row_hashes = rows->hashes; a.row_nhash = rows->nhash; a.row_stride = rows->s;
    a.col_hashes = cols->hashes; a.col_nhash = cols->nhash; a.col_stride = cols->s;
    a.mtiles = nullptr;
    a.out = reinterpret_cast<uint2 *>(out_dev);
    a.row_begin = row_begin; a.row_end = row_end;
    a.ncols = cols->n;
    a.out_base = triangle ? row_begin * (row_begin - (row_begin ? 1 : 0)) / 2 : 0;
    a.s = (uint32_t)s64;
    a.triangle = triangle ? 1 : 0;
    a.rows_per_tile = 0;
    a.unroll = 0;
    a.row_win = a.col_win = nullptr;
    a.win = a.nwin = a.win_lo = a.win_hi = a.win_ecap = 0;
    a.win_mask = nullptr;
    a.win_kmax = 0;
    a.xcd_remap = 0;
    a.stage_pack = 0;
    a.dbg = nullptr;
    if (const char *e = ctx_opt(ctx, "MASHGPU_COMPARE_XCD")) a.xcd_remap = atoi(e) != 0;
    if (const char *e = ctx_opt(ctx, "MASHGPU_COMPARE_VARIANT")) a.unroll = (uint32_t)atoi(e);
    const char *force = ctx_opt(ctx, "MASHGPU_COMPARE_KERNEL");
    // Inverted-index engine first: it takes the job when the counting pass says so (or when forced)
    if (!force || strcmp(force, "sparse") == 0) {
        bool handled = false;
        const int rcs = run_compare_sparse(ctx, rows, cols, row_begin, row_end, triangle, a.s, out_dev, force != nullptr, &handled);
        if (rcs != MG_OK || handled) return rcs;
        if (force) return fail(ctx, MG_ERR_UNSUPPORTED, "compare: the sparse engine cannot take this table");
    }
    const bool want_generic = force && strcmp(force, "generic") == 0;
    const bool use_merged = !want_generic && mg::compare_merged_supported(a.s);
    a.row_pfx = a.col_pfx = nullptr;
    a.row_pfx_stride = a.col_pfx_stride = 0;
    a.pfx_shr = 0;
    if (!use_merged && !want_generic && a.s > 16384 && a.s <= (1u << 22) &&
        !(ctx_opt(ctx, "MASHGPU_COMPARE_WINDOWS") && atoi(ctx_opt(ctx, "MASHGPU_COMPARE_WINDOWS")) == 0)) {
        // Beyond the plain tile kernel's reach (s > 16 384) the value-window mode still applies: its
        // tiles hold one window's hashes whatever s is.  If some class cannot be windowed, nothing
        // has been launched and the generic kernel below takes the call.
        a.rows_per_tile = 1;
        const uint64_t maxc = triangle ? (row_end - 1) : cols->n;
        const int rcw = run_compare_merged(ctx, rows, cols, row_begin, row_end, triangle, a, 1, cols->n >= 40000 ? 16384 : 8192, maxc, true);
        if (rcw != kNoWindowPlan) return rcw;
    }
    if (!use_merged) {
        prof_begin(ctx, ctx->prof_compare);
        HIP_TRY(ctx, mg::launch_compare_generic(a, ctx->stream));
        prof_end(ctx, ctx->prof_compare);
        return MG_OK;
    }
    uint32_t R = mg::compare_merged_rows(a.s);
    if (const char *e = ctx_opt(ctx, "MASHGPU_COMPARE_ROWS")) { uint32_t v = (uint32_t)atoi(e); if (v >= 1 && v < R) R = v; }
    // columns per tile: long tiles amortise the table build and the ragged end of a tile
    // (profiles/r01_compare_sweep2.txt); smaller problems keep more tiles for balance
    uint64_t CC = cols->n >= 40000 ? 16384 : 8192;
    if (const char *e = ctx_opt(ctx, "MASHGPU_COMPARE_COLS")) { uint64_t v = strtoull(e, nullptr, 10); if (v >= 16) CC = v; }
    a.rows_per_tile = R;
    const uint64_t maxcols = triangle ? (row_end - 1) : cols->n;       // columns [0, maxcols)
    return run_compare_merged(ctx, rows, cols, row_begin, row_end, triangle, a, R, CC, maxcols, false);
}

static uint64_t tri_pairs(uint64_t row_begin, uint64_t row_end)
{
    // sum_{i=row_begin}^{row_end-1} i
    auto tri = [](uint64_t x) { return x ? x * (x - 1) / 2 : 0; };
    return tri(row_end) - tri(row_begin);
}

int mg_compare_tri_dev(mg_ctx *ctx, const mg_table *t, uint64_t row_begin, uint64_t row_end, mg_counts *out_dev)
{
    if (!ctx) return MG_ERR_INVALID;
    if (!t || !out_dev) return fail(ctx, MG_ERR_INVALID, "mg_compare_tri_dev: NULL argument");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    const int rc = run_compare(ctx, t, t, row_begin, row_end, true, out_dev);
    if (rc != MG_OK || ctx->async) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MG_OK;
}

int mg_compare_rect_dev(mg_ctx *ctx, const mg_table *ref, const mg_table *qry, uint64_t q_begin,
                        uint64_t q_end, mg_counts *out_dev)
{
    if (!ctx) return MG_ERR_INVALID;
    if (!ref || !qry || !out_dev) return fail(ctx, MG_ERR_INVALID, "mg_compare_rect_dev: NULL argument");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    const int rc = run_compare(ctx, qry, ref, q_begin, q_end, false, out_dev);
    if (rc != MG_OK || ctx->async) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MG_OK;
}

// host-output variants: bounded device staging, processed in row blocks
static int compare_host(mg_ctx *ctx, const mg_table *rows, const mg_table *cols, uint64_t rb, uint64_t re,
                        bool triangle, mg_counts *out_host)
{
    if (re > rows->n) re = rows->n;
    if (rb >= re) return MG_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t max_pairs = 1ull << 27;                 // 1 GiB of {numer,denom}
    mg_counts *d_out = nullptr;
    uint64_t done = 0, r = rb;
    uint64_t cap_pairs = 0;
    int rc = MG_OK;
    while (r < re && rc == MG_OK) {
        uint64_t r2 = r, pairs = 0;
        while (r2 < re) {
            const uint64_t add = triangle ? r2 : cols->n;
            if (pairs && pairs + add > max_pairs) break;
            pairs += add;
            r2++;
        }
        if (pairs > cap_pairs) {
            if (d_out) ctx_free(ctx, d_out);
            d_out = nullptr;
            if (ctx_malloc(ctx, (void **)&d_out, std::max<uint64_t>(pairs, 1) * sizeof(mg_counts)) != hipSuccess) {
                rc = fail(ctx, MG_ERR_NOMEM, "compare: device allocation failed");
                break;
            }
            cap_pairs = pairs;
        }
        rc = run_compare(ctx, rows, cols, r, r2, triangle, d_out);
        if (rc == MG_OK && pairs) {
            if (hipMemcpyAsync(out_host + done, d_out, pairs * sizeof(mg_counts), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                rc = fail(ctx, MG_ERR_HIP, "compare: D2H copy failed");
        }
        done += pairs;
        r = r2;
    }
    if (d_out) ctx_free(ctx, d_out);
    return rc;
}

int mg_compare_tri_host(mg_ctx *ctx, const mg_table *t, uint64_t row_begin, uint64_t row_end, mg_counts *out_host)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!t || !out_host) return fail(ctx, MG_ERR_INVALID, "mg_compare_tri_host: NULL argument");
    return compare_host(ctx, t, t, row_begin, row_end, true, out_host);
}

int mg_compare_rect_host(mg_ctx *ctx, const mg_table *ref, const mg_table *qry, uint64_t q_begin,
                         uint64_t q_end, mg_counts *out_host)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!ref || !qry || !out_host) return fail(ctx, MG_ERR_INVALID, "mg_compare_rect_host: NULL argument");
    return compare_host(ctx, qry, ref, q_begin, q_end, false, out_host);
}

/* ------------------------------------------------------------------ finishing */

double mg_distance(uint32_t numer, uint32_t denom, int kmer_size) { return mg::mash_distance(numer, denom, kmer_size); }

double mg_p_value(uint64_t x, uint64_t len_ref, uint64_t len_qry, double kmer_space, uint64_t sketch_size)
{
    return mg::p_value(x, len_ref, len_qry, kmer_space, sketch_size);
}

static inline void finish_one(const mg_counts &c, uint64_t len_ref, uint64_t len_qry, int k, double kmer_space,
                              double max_d, double max_p, mg_pair *o)
{
    memset(o, 0, sizeof *o);
    o->numer = c.numer;
    o->denom = c.denom;
    o->distance = mg::mash_distance(c.numer, c.denom, k);
    if (max_d >= 0 && o->distance > max_d) return;                      // CommandDistance.cpp:409-412
    o->p_value = mg::p_value(c.numer, len_ref, len_qry, kmer_space, c.denom);
    if (max_p >= 0 && o->p_value > max_p) return;                       // :419-422
    o->pass = 1;
}

// Distance and p-value are ~165 ns of scalar libm work per pair: at 10^7 pairs and more that, not
// the kernels, is what a caller waits for, so large batches are split over host threads
// (every pair is independent; the output is identical).
extern "C++" {
template <class F>
static void finish_parallel(uint64_t items, F fn)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (items < (1ull << 20) || nt < 2) { fn(0, items); return; }
    std::vector<std::thread> th;
    const uint64_t per = (items + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        const uint64_t b = t * per, e = std::min(items, b + per);
        if (b < e) th.emplace_back([=]() { fn(b, e); });
    }
    for (auto &x : th) x.join();
}
} // extern "C++"

int mg_finish_tri_host(const mg_counts *counts, const uint64_t *lengths, uint64_t row_begin, uint64_t row_end,
                       int kmer_size, double kmer_space, double max_distance, double max_p_value, mg_pair *out)
{
    if (!counts || !lengths || !out) return MG_ERR_INVALID;
    if (row_end <= row_begin) return MG_OK;
    const uint64_t base = tri_pairs(0, row_begin), total = tri_pairs(row_begin, row_end);
    // split by pairs, then round each cut up to a row boundary, so threads get equal work
    auto row_of = [=](uint64_t pair) {
        uint64_t lo = row_begin, hi = row_end;           // first row whose start is >= pair
        while (lo < hi) {
            const uint64_t mid = lo + (hi - lo) / 2;
            if (tri_pairs(0, mid) - base >= pair) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    finish_parallel(total, [=](uint64_t b, uint64_t e) {
        const uint64_t r0 = row_of(b), r1 = e >= total ? row_end : row_of(e);
        for (uint64_t i = r0; i < r1; i++) {
            uint64_t idx = tri_pairs(0, i) - base;
            for (uint64_t j = 0; j < i; j++, idx++)
                finish_one(counts[idx], lengths[i], lengths[j], kmer_size, kmer_space, max_distance, max_p_value, out + idx);
        }
    });
    return MG_OK;
}

int mg_finish_rect_host(const mg_counts *counts, const uint64_t *len_ref, uint64_t nref, const uint64_t *len_qry,
                        uint64_t nqry, int kmer_size, double kmer_space, double max_distance, double max_p_value,
                        mg_pair *out)
{
    if (!counts || !len_ref || !len_qry || !out) return MG_ERR_INVALID;
    finish_parallel(nqry * nref, [=](uint64_t b, uint64_t e) {
        for (uint64_t idx = b; idx < e; idx++) {
            const uint64_t q = idx / nref, r = idx - q * nref;
            finish_one(counts[idx], len_ref[r], len_qry[q], kmer_size, kmer_space, max_distance, max_p_value, out + idx);
        }
    });
    return MG_OK;
}

/* ------------------------------------------------- thresholded all-pairs (edge list) */

// smallest numer whose distance passes `max_d`, for every denom in [0, s]: the
// same host arithmetic finish_one uses, so the device's integer test selects
// exactly the pairs the reference's `distance > maxDistance` test keeps.
static void build_min_numer(uint32_t s, int k, double max_d, std::vector<uint32_t> &out)
{
    out.assign((size_t)s + 1, 0);
    for (uint32_t d = 0; d <= s; d++) {
        if (mg::mash_distance(0, d, k) <= max_d) { out[d] = 0; continue; }
        if (!(mg::mash_distance(d, d, k) <= max_d)) { out[d] = d + 1; continue; }
        uint32_t lo = 0, hi = d;                           // lo fails, hi passes
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (mg::mash_distance(mid, d, k) <= max_d) hi = mid; else lo = mid;
        }
        out[d] = hi;
    }
}

static int compare_filter(mg_ctx *ctx, const mg_table *rows, const mg_table *cols, uint64_t rb, uint64_t re,
                          bool triangle, int kmer_size, double max_distance, mg_edge *out_host, uint64_t capacity,
                          uint64_t *count_out)
{
    *count_out = 0;
    if (re > rows->n) re = rows->n;
    if (rb >= re) return MG_OK;
    if (kmer_size < 1) return fail(ctx, MG_ERR_INVALID, "compare filter: bad k-mer size");
    const uint64_t s64 = std::min(rows->s, cols->s);
    if (s64 > 0xFFFFFFFEull) return fail(ctx, MG_ERR_INVALID, "compare: sketch size too large");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> min_numer;
    build_min_numer((uint32_t)s64, kmer_size, max_distance, min_numer);

    // row blocks of up to 2^30 pairs (8 GiB of counts): large launches keep the
    // tail of the compare kernel short; survivors leave in windows of 2^26 edges
    const uint64_t max_pairs = 1ull << 30, window = 1ull << 26;
    const uint64_t all_pairs = triangle ? tri_pairs(rb, re) : (re - rb) * cols->n;
    const uint64_t blk_pairs = std::min(all_pairs, max_pairs + (triangle ? re : cols->n));
    mg_counts *d_counts = nullptr;
    uint4 *d_edges = nullptr;
    uint32_t *d_min = nullptr, *d_segc = nullptr;
    unsigned long long *d_sego = nullptr, *d_n = nullptr;
    uint64_t total = 0, r = rb;
    int rc = MG_OK;
    auto cleanup = [&]() {
        hipStreamSynchronize(ctx->stream);
        for (void *p : {(void *)d_counts, (void *)d_edges, (void *)d_min, (void *)d_segc, (void *)d_sego, (void *)d_n})
            if (p) hipFree(p);
    };
    if (all_pairs == 0) return MG_OK;
    const uint64_t nseg_max = mg::filter_segments(blk_pairs);
    if (hipMalloc(&d_min, min_numer.size() * 4) != hipSuccess || hipMalloc(&d_n, 8) != hipSuccess ||
        hipMalloc(&d_counts, blk_pairs * sizeof(mg_counts)) != hipSuccess ||
        hipMalloc(&d_edges, std::min(blk_pairs, window) * sizeof(uint4)) != hipSuccess ||
        hipMalloc(&d_segc, nseg_max * 4) != hipSuccess || hipMalloc(&d_sego, nseg_max * 8) != hipSuccess ||
        hipMemcpyAsync(d_min, min_numer.data(), min_numer.size() * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        cleanup();
        return fail(ctx, MG_ERR_NOMEM, "compare filter: device allocation failed");
    }
    while (r < re && rc == MG_OK) {
        uint64_t r2 = r, pairs = 0;
        while (r2 < re) {
            const uint64_t add = triangle ? r2 : cols->n;
            if (pairs && pairs + add > max_pairs) break;
            pairs += add;
            r2++;
        }
        if (pairs) {
            rc = run_compare(ctx, rows, cols, r, r2, triangle, d_counts);
            if (rc != MG_OK) break;
            mg::FilterArgs f;
            f.counts = reinterpret_cast<const uint2 *>(d_counts);
            f.min_numer = d_min;
            f.seg_count = d_segc;
            f.seg_off = d_sego;
            f.edges = d_edges;
            f.pairs = pairs;
            f.first_row = r;
            f.ncols = cols->n;
            f.win_lo = 0; f.win_n = 0;
            f.s = (uint32_t)s64;
            f.triangle = triangle ? 1 : 0;
            unsigned long long n_blk = 0;
            if (mg::launch_filter_count(f, d_n, ctx->stream) != hipSuccess ||
                hipMemcpyAsync(&n_blk, d_n, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) {
                rc = fail(ctx, MG_ERR_HIP, "compare filter: kernel failed");
                break;
            }
            // survivors already rank in reference order; skip the copy once `capacity` is exceeded
            for (uint64_t lo = 0; lo < n_blk && total + n_blk <= capacity; lo += window) {
                f.win_lo = lo;
                f.win_n = std::min<uint64_t>(window, n_blk - lo);
                if (mg::launch_filter_write(f, ctx->stream) != hipSuccess ||
                    hipMemcpyAsync(out_host + total + lo, d_edges, f.win_n * sizeof(mg_edge), hipMemcpyDeviceToHost,
                                   ctx->stream) != hipSuccess ||
                    hipStreamSynchronize(ctx->stream) != hipSuccess) {
                    rc = fail(ctx, MG_ERR_HIP, "compare filter: compaction failed");
                    break;
                }
            }
            total += n_blk;
        }
        r = r2;
    }
    cleanup();
    if (rc != MG_OK) return rc;
    *count_out = total;
    if (total > capacity) return fail(ctx, MG_ERR_NOMEM, "compare filter: more passing pairs than `capacity` (see *count_out)");
    return MG_OK;
}

int mg_compare_tri_filter_host(mg_ctx *ctx, const mg_table *t, uint64_t row_begin, uint64_t row_end, int kmer_size,
                               double max_distance, mg_edge *out_host, uint64_t capacity, uint64_t *count_out)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!t || !count_out || (!out_host && capacity)) return fail(ctx, MG_ERR_INVALID, "mg_compare_tri_filter_host: NULL argument");
    return compare_filter(ctx, t, t, row_begin, row_end, true, kmer_size, max_distance, out_host, capacity, count_out);
}

int mg_compare_rect_filter_host(mg_ctx *ctx, const mg_table *ref, const mg_table *qry, uint64_t q_begin, uint64_t q_end,
                                int kmer_size, double max_distance, mg_edge *out_host, uint64_t capacity,
                                uint64_t *count_out)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!ref || !qry || !count_out || (!out_host && capacity))
        return fail(ctx, MG_ERR_INVALID, "mg_compare_rect_filter_host: NULL argument");
    return compare_filter(ctx, qry, ref, q_begin, q_end, false, kmer_size, max_distance, out_host, capacity, count_out);
}

/* ------------------------------------------------- device tail of compareSketches */

static_assert(sizeof(mg_pair) == sizeof(mg::FinishPair) && sizeof(mg_result) == sizeof(mg::FinishEdge), "ABI structs");

// What the device finish needs besides the counts: the distance table (host libm, one row per
// denominator flagged in `seen`, row s always) and the integer form of the distance filter.
struct FinishTables {
    DevBuf<uint32_t> d_start, d_min;
    DevBuf<double> d_lut;
    bool complete = true;                   // false: some flagged denominator did not fit the budget (device yields NaN)
    explicit FinishTables(mg_ctx *c) : d_start(c), d_min(c), d_lut(c) {}
};

static int build_finish_tables(mg_ctx *ctx, uint32_t s, int k, double max_d, const std::vector<uint32_t> &seen, FinishTables &ft)
{
    const uint64_t budget = 1ull << 26;                       // doubles (512 MiB): every denominator up to s = 11 583
    std::vector<uint32_t> start((size_t)s + 1, 0xFFFFFFFFu);
    std::vector<double> lut;
    auto add_row = [&](uint32_t d) {
        if (start[d] != 0xFFFFFFFFu) return;
        if (lut.size() + (uint64_t)d + 1 > budget || lut.size() + (uint64_t)d + 1 > 0xFFFFFFF0ull) { ft.complete = false; return; }
        start[d] = (uint32_t)lut.size();
        for (uint32_t x = 0; x <= d; x++) lut.push_back(mg::mash_distance(x, d, k));
    };
    add_row(s);
    for (uint32_t d = 0; d <= s && d < seen.size(); d++)
        if (seen[d]) add_row(d);
    HIP_TRY(ctx, ft.d_start.alloc(start.size()));
    HIP_TRY(ctx, ft.d_lut.alloc(lut.size()));
    HIP_TRY(ctx, hipMemcpyAsync(ft.d_start, start.data(), start.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ft.d_lut, lut.data(), lut.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    if (max_d >= 0 && max_d < 1.0) {
        std::vector<uint32_t> mn;
        build_min_numer(s, k, max_d, mn);
        HIP_TRY(ctx, ft.d_min.alloc(mn.size()));
        HIP_TRY(ctx, hipMemcpyAsync(ft.d_min, mn.data(), mn.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          // the host vectors go out of scope
    return MG_OK;
}

// counts (device) of `pairs` pairs starting at row `first_row` -> mg_pair (device)
static int finish_pairs_dev(mg_ctx *ctx, const mg_table *rows, const mg_table *cols, const mg_counts *counts_dev, uint64_t pairs,
                            uint64_t first_row, bool triangle, int kmer_size, double kmer_space, double max_d, double max_p,
                            mg_pair *out_dev, bool *complete_out)
{
    if (pairs == 0) return MG_OK;
    if (!rows->lengths || !cols->lengths) return fail(ctx, MG_ERR_INVALID, "finish: the tables carry no lengths");
    if (kmer_size < 1) return fail(ctx, MG_ERR_INVALID, "finish: bad k-mer size");
    const uint64_t s64 = std::min(rows->s, cols->s);
    if (s64 > 0xFFFFFFFEull) return fail(ctx, MG_ERR_INVALID, "finish: sketch size too large");
    const uint32_t s = (uint32_t)s64;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf<uint32_t> d_seen(ctx);
    HIP_TRY(ctx, d_seen.alloc((uint64_t)s + 1));
    HIP_TRY(ctx, hipMemsetAsync(d_seen, 0, ((uint64_t)s + 1) * 4, ctx->stream));
    HIP_TRY(ctx, mg::launch_denom_flags(reinterpret_cast<const uint2 *>(counts_dev), pairs, s, d_seen, ctx->stream));
    std::vector<uint32_t> seen((size_t)s + 1);
    HIP_TRY(ctx, hipMemcpyAsync(seen.data(), d_seen, seen.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    FinishTables ft(ctx);
    int rc = build_finish_tables(ctx, s, kmer_size, max_d, seen, ft);
    if (rc != MG_OK) return rc;
    mg::FinishArgs a{};
    a.counts = reinterpret_cast<const uint2 *>(counts_dev);
    a.pairs = pairs;
    a.first_row = first_row;
    a.ncols = cols->n;
    a.len_row = rows->lengths;
    a.len_col = cols->lengths;
    a.min_numer = ft.d_min;
    a.lut_start = ft.d_start;
    a.lut = ft.d_lut;
    a.kmer_space = kmer_space;
    a.max_p = max_p;
    a.s = s;
    a.triangle = triangle ? 1 : 0;
    a.pairs_out = reinterpret_cast<mg::FinishPair *>(out_dev);
    HIP_TRY(ctx, mg::launch_finish_pairs(a, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));          // the tables are released on return
    if (complete_out) *complete_out = ft.complete;
    else if (!ft.complete)
        return fail(ctx, MG_ERR_UNSUPPORTED, "finish: too many distinct denominators for the device distance table (use mg_finish_*_host)");
    return MG_OK;
}

int mg_finish_tri_dev(mg_ctx *ctx, const mg_table *t, const mg_counts *counts_dev, uint64_t row_begin, uint64_t row_end,
                      int kmer_size, double kmer_space, double max_distance, double max_p_value, mg_pair *out_dev)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!t || !counts_dev || !out_dev) return fail(ctx, MG_ERR_INVALID, "mg_finish_tri_dev: NULL argument");
    if (row_end > t->n) row_end = t->n;
    if (row_begin >= row_end) return MG_OK;
    return finish_pairs_dev(ctx, t, t, counts_dev, tri_pairs(row_begin, row_end), row_begin, true, kmer_size, kmer_space,
                            max_distance, max_p_value, out_dev, nullptr);
}

int mg_finish_rect_dev(mg_ctx *ctx, const mg_table *ref, const mg_table *qry, const mg_counts *counts_dev, uint64_t q_begin,
                       uint64_t q_end, int kmer_size, double kmer_space, double max_distance, double max_p_value,
                       mg_pair *out_dev)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!ref || !qry || !counts_dev || !out_dev) return fail(ctx, MG_ERR_INVALID, "mg_finish_rect_dev: NULL argument");
    if (q_end > qry->n) q_end = qry->n;
    if (q_begin >= q_end) return MG_OK;
    return finish_pairs_dev(ctx, qry, ref, counts_dev, (q_end - q_begin) * ref->n, q_begin, false, kmer_size, kmer_space,
                            max_distance, max_p_value, out_dev, nullptr);
}

// compare + finish on the device, full PairOutput records to the host (32 B per pair), in row blocks
static int compare_pairs_host(mg_ctx *ctx, const mg_table *rows, const mg_table *cols, uint64_t rb, uint64_t re, bool triangle,
                              int kmer_size, double kmer_space, double max_d, double max_p, mg_pair *out_host)
{
    if (re > rows->n) re = rows->n;
    if (rb >= re) return MG_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t max_pairs = 1ull << 26;                   // 2 GiB of records, 512 MiB of counts
    DevBuf<mg_counts> d_counts(ctx);
    DevBuf<mg_pair> d_pairs(ctx);
    uint64_t cap = 0, done = 0, r = rb;
    std::vector<uint64_t> len_rows, len_cols;                // host copies, only if a block must be patched
    while (r < re) {
        uint64_t r2 = r, pairs = 0;
        while (r2 < re) {
            const uint64_t add = triangle ? r2 : cols->n;
            if (pairs && pairs + add > max_pairs) break;
            pairs += add;
            r2++;
        }
        if (pairs > cap) {
            if (d_counts.p) { ctx_free(ctx, d_counts.release()); }
            if (d_pairs.p) { ctx_free(ctx, d_pairs.release()); }
            if (d_counts.alloc(pairs) != hipSuccess || d_pairs.alloc(pairs) != hipSuccess)
                return fail(ctx, MG_ERR_NOMEM, "compare: device allocation failed");
            cap = pairs;
        }
        int rc = pairs ? run_compare(ctx, rows, cols, r, r2, triangle, d_counts) : MG_OK;
        if (rc != MG_OK) return rc;
        bool complete = true;
        if (pairs) {
            rc = finish_pairs_dev(ctx, rows, cols, d_counts, pairs, r, triangle, kmer_size, kmer_space, max_d, max_p, d_pairs, &complete);
            if (rc != MG_OK) return rc;
            if (hipMemcpyAsync(out_host + done, d_pairs, pairs * sizeof(mg_pair), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)
                return fail(ctx, MG_ERR_HIP, "compare: D2H copy failed");
            if (!complete) {
                // a denominator beyond the device table's budget left NaN distances: those pairs are finished here
                if (len_rows.empty()) {
                    len_rows.resize(rows->n);
                    len_cols.resize(cols->n);
                    if (hipMemcpy(len_rows.data(), rows->lengths, rows->n * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                        hipMemcpy(len_cols.data(), cols->lengths, cols->n * 8, hipMemcpyDeviceToHost) != hipSuccess)
                        return fail(ctx, MG_ERR_HIP, "compare: D2H copy failed");
                }
                uint64_t idx = 0;
                for (uint64_t i = r; i < r2; i++) {
                    const uint64_t ncol = triangle ? i : cols->n;
                    for (uint64_t j = 0; j < ncol; j++, idx++) {
                        mg_pair &pr = out_host[done + idx];
                        if (pr.distance == pr.distance) continue;
                        const mg_counts c{pr.numer, pr.denom};
                        finish_one(c, len_rows[i], len_cols[j], kmer_size, kmer_space, max_d, max_p, &pr);
                    }
                }
            }
        }
        done += pairs;
        r = r2;
    }
    return MG_OK;
}

int mg_compare_tri_pairs_host(mg_ctx *ctx, const mg_table *t, uint64_t row_begin, uint64_t row_end, int kmer_size,
                              double kmer_space, double max_distance, double max_p_value, mg_pair *out_host)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!t || !out_host) return fail(ctx, MG_ERR_INVALID, "mg_compare_tri_pairs_host: NULL argument");
    if (!t->lengths) return fail(ctx, MG_ERR_INVALID, "mg_compare_tri_pairs_host: the table carries no lengths");
    return compare_pairs_host(ctx, t, t, row_begin, row_end, true, kmer_size, kmer_space, max_distance, max_p_value, out_host);
}

int mg_compare_rect_pairs_host(mg_ctx *ctx, const mg_table *ref, const mg_table *qry, uint64_t q_begin, uint64_t q_end,
                               int kmer_size, double kmer_space, double max_distance, double max_p_value, mg_pair *out_host)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!ref || !qry || !out_host) return fail(ctx, MG_ERR_INVALID, "mg_compare_rect_pairs_host: NULL argument");
    if (!ref->lengths || !qry->lengths) return fail(ctx, MG_ERR_INVALID, "mg_compare_rect_pairs_host: the tables carry no lengths");
    return compare_pairs_host(ctx, qry, ref, q_begin, q_end, false, kmer_size, kmer_space, max_distance, max_p_value, out_host);
}

// compare + both filters + compaction on the device: survivors only, as full records, in reference order
static int compare_results(mg_ctx *ctx, const mg_table *rows, const mg_table *cols, uint64_t rb, uint64_t re, bool triangle,
                           int kmer_size, double kmer_space, double max_d, double max_p, mg_result *out_host, uint64_t capacity,
                           uint64_t *count_out)
{
    *count_out = 0;
    if (re > rows->n) re = rows->n;
    if (rb >= re) return MG_OK;
    if (kmer_size < 1) return fail(ctx, MG_ERR_INVALID, "compare: bad k-mer size");
    const uint64_t s64 = std::min(rows->s, cols->s);
    if (s64 > 0xFFFFFFFEull) return fail(ctx, MG_ERR_INVALID, "compare: sketch size too large");
    const uint32_t s = (uint32_t)s64;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t max_pairs = 1ull << 30, window = 1ull << 25;
    const uint64_t all_pairs = triangle ? tri_pairs(rb, re) : (re - rb) * cols->n;
    if (all_pairs == 0) return MG_OK;
    // ---- a filter is on: only pairs that share a hash can pass (numer = 0 means distance 1 and p-value 1), and
    // those are the inverted-index engine's candidates -- no matrix is filled, no 8 B per pair read back by
    // the filter pass: discover + merge, the candidates put into reference order, the same two finish passes
    // over that list.
    const char *force_kernel = ctx_opt(ctx, "MASHGPU_COMPARE_KERNEL");
    if (((max_d >= 0.0 && max_d < 1.0) || (max_p >= 0.0 && max_p < 1.0)) && (!force_kernel || strcmp(force_kernel, "sparse") == 0) &&
        !ctx_opt(ctx, "MASHGPU_RESULTS_MATRIX")) {
        SparseJob job;
        bool handled = false;
        int rc = run_compare_sparse(ctx, rows, cols, rb, re, triangle, s, nullptr, force_kernel != nullptr, &handled, &job);
        if (rc != MG_OK) return rc;
        // (a list of 2^32 candidates and more, or one whose scratch does not fit, is no error: the blocked matrix path
        //  below does the job in 2^30-pair blocks -- ADVICE r3)
        const uint64_t K = handled ? job.cand : 0;
        if (handled && K == 0) return MG_OK;
        const uint32_t nrows = (uint32_t)(re - rb);
        DevBuf<uint2> d_rc(ctx), d_cnt(ctx);
        DevBuf<uint32_t> d_byrow(ctx), d_base(ctx), d_segc2(ctx), d_seen2(ctx);
        DevBuf<unsigned long long> d_masks2(ctx), d_sego2(ctx), d_n2(ctx);
        DevBuf<mg::FinishEdge> d_edges2(ctx);
        DevBuf<unsigned char> d_temp(ctx);
        const size_t tb = mg::sparse_gather_temp_bytes(nrows);
        bool list_ok = handled && K < (1ull << 32);
        if (list_ok && (d_rc.alloc(K) != hipSuccess || d_cnt.alloc(K) != hipSuccess || d_byrow.alloc(nrows) != hipSuccess || d_base.alloc(nrows) != hipSuccess ||
                        d_temp.alloc(std::max<size_t>(tb, 16)) != hipSuccess || d_masks2.alloc(mg::finish_mask_words(K)) != hipSuccess ||
                        d_segc2.alloc(mg::finish_segments(K)) != hipSuccess || d_sego2.alloc(mg::finish_segments(K)) != hipSuccess || d_n2.alloc(1) != hipSuccess ||
                        d_seen2.alloc((uint64_t)s + 1) != hipSuccess || d_edges2.alloc(std::min(K, window)) != hipSuccess)) {
            (void)hipGetLastError();
            list_ok = false;
        }
        if (list_ok) {
            HIP_TRY(ctx, hipMemsetAsync(d_byrow, 0, (size_t)nrows * 4, ctx->stream));
            HIP_TRY(ctx, mg::launch_sparse_gather_rows(job.args, d_byrow, d_base, d_temp, tb, triangle ? 0u : (uint32_t)rb, d_rc, d_cnt, ctx->stream));
            std::vector<uint32_t> seen2((size_t)s + 1, 0);
            FinishTables fa(ctx);
            rc = build_finish_tables(ctx, s, kmer_size, max_d, seen2, fa);
            if (rc != MG_OK) return rc;
            mg::FinishArgs f{};
            f.counts = d_cnt;
            f.list_rc = d_rc;
            f.pairs = K;
            f.first_row = rb;
            f.ncols = cols->n;
            f.len_row = rows->lengths;
            f.len_col = cols->lengths;
            f.min_numer = fa.d_min;
            f.lut_start = fa.d_start;
            f.lut = fa.d_lut;
            f.kmer_space = kmer_space;
            f.max_p = max_p;
            f.s = s;
            f.triangle = triangle ? 1 : 0;
            f.masks = d_masks2;
            f.seg_count = d_segc2;
            f.seg_off = d_sego2;
            f.denom_seen = d_seen2;
            f.edges = d_edges2;
            unsigned long long n_all = 0;
            HIP_TRY(ctx, hipMemsetAsync(d_seen2, 0, ((uint64_t)s + 1) * 4, ctx->stream));
            HIP_TRY(ctx, mg::launch_finish_mark(f, d_n2, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(&n_all, d_n2, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(seen2.data(), d_seen2, seen2.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            *count_out = n_all;
            if (n_all > capacity) return fail(ctx, MG_ERR_NOMEM, "compare: more passing pairs than `capacity` (see *count_out)");
            if (n_all) {
                FinishTables fb(ctx);
                rc = build_finish_tables(ctx, s, kmer_size, max_d, seen2, fb);
                if (rc != MG_OK) return rc;
                f.lut_start = fb.d_start;
                f.lut = fb.d_lut;
                f.min_numer = fb.d_min;
                for (uint64_t lo = 0; lo < n_all; lo += window) {
                    f.win_lo = lo;
                    f.win_n = std::min<uint64_t>(window, n_all - lo);
                    HIP_TRY(ctx, mg::launch_finish_write(f, ctx->stream));
                    HIP_TRY(ctx, hipMemcpyAsync(out_host + lo, d_edges2, f.win_n * sizeof(mg_result), hipMemcpyDeviceToHost, ctx->stream));
                    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                }
                if (!fb.complete)
                    for (uint64_t i = 0; i < n_all; i++) {
                        mg_result &e = out_host[i];
                        if (e.distance != e.distance) e.distance = mg::mash_distance(e.numer, e.denom, kmer_size);
                    }
            }
            return MG_OK;
        }
    }
    const uint64_t blk_pairs = std::min(all_pairs, max_pairs + (triangle ? re : cols->n));
    DevBuf<mg_counts> d_counts;
    DevBuf<mg::FinishEdge> d_edges;
    DevBuf<unsigned long long> d_masks, d_sego, d_n;
    DevBuf<uint32_t> d_segc, d_seen;
    if (d_counts.alloc(blk_pairs) != hipSuccess || d_edges.alloc(std::min(blk_pairs, window)) != hipSuccess ||
        d_masks.alloc(mg::finish_mask_words(blk_pairs)) != hipSuccess || d_segc.alloc(mg::finish_segments(blk_pairs)) != hipSuccess ||
        d_sego.alloc(mg::finish_segments(blk_pairs)) != hipSuccess || d_n.alloc(1) != hipSuccess || d_seen.alloc((uint64_t)s + 1) != hipSuccess)
        return fail(ctx, MG_ERR_NOMEM, "compare: device allocation failed");
    // pass A needs the distance filter but no distances: tables without any extra denominator row
    std::vector<uint32_t> seen((size_t)s + 1, 0);
    uint64_t total = 0, r = rb;
    std::vector<uint64_t> len_rows, len_cols;
    while (r < re) {
        uint64_t r2 = r, pairs = 0;
        while (r2 < re) {
            const uint64_t add = triangle ? r2 : cols->n;
            if (pairs && pairs + add > max_pairs) break;
            pairs += add;
            r2++;
        }
        if (pairs) {
            int rc = run_compare(ctx, rows, cols, r, r2, triangle, d_counts);
            if (rc != MG_OK) return rc;
            FinishTables fa(ctx);
            std::fill(seen.begin(), seen.end(), 0u);
            rc = build_finish_tables(ctx, s, kmer_size, max_d, seen, fa);
            if (rc != MG_OK) return rc;
            mg::FinishArgs a{};
            a.counts = reinterpret_cast<const uint2 *>(d_counts.p);
            a.pairs = pairs;
            a.first_row = r;
            a.ncols = cols->n;
            a.len_row = rows->lengths;
            a.len_col = cols->lengths;
            a.min_numer = fa.d_min;
            a.lut_start = fa.d_start;
            a.lut = fa.d_lut;
            a.kmer_space = kmer_space;
            a.max_p = max_p;
            a.s = s;
            a.triangle = triangle ? 1 : 0;
            a.masks = d_masks;
            a.seg_count = d_segc;
            a.seg_off = d_sego;
            a.denom_seen = d_seen;
            a.edges = d_edges;
            unsigned long long n_blk = 0;
            HIP_TRY(ctx, hipMemsetAsync(d_seen, 0, ((uint64_t)s + 1) * 4, ctx->stream));
            HIP_TRY(ctx, mg::launch_finish_mark(a, d_n, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(&n_blk, d_n, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(seen.data(), d_seen, seen.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (n_blk && total + n_blk <= capacity) {
                FinishTables fb(ctx);                        // now with the rows of the survivors' denominators
                rc = build_finish_tables(ctx, s, kmer_size, max_d, seen, fb);
                if (rc != MG_OK) return rc;
                a.lut_start = fb.d_start;
                a.lut = fb.d_lut;
                a.min_numer = fb.d_min;
                for (uint64_t lo = 0; lo < n_blk; lo += window) {
                    a.win_lo = lo;
                    a.win_n = std::min<uint64_t>(window, n_blk - lo);
                    HIP_TRY(ctx, mg::launch_finish_write(a, ctx->stream));
                    HIP_TRY(ctx, hipMemcpyAsync(out_host + total + lo, d_edges, a.win_n * sizeof(mg_result), hipMemcpyDeviceToHost, ctx->stream));
                    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                }
                if (!fb.complete) {
                    for (uint64_t i = 0; i < n_blk; i++) {
                        mg_result &e = out_host[total + i];
                        if (e.distance != e.distance) e.distance = mg::mash_distance(e.numer, e.denom, kmer_size);
                    }
                }
            }
            total += n_blk;
        }
        r = r2;
    }
    *count_out = total;
    if (total > capacity) return fail(ctx, MG_ERR_NOMEM, "compare: more passing pairs than `capacity` (see *count_out)");
    return MG_OK;
}

int mg_compare_tri_results_host(mg_ctx *ctx, const mg_table *t, uint64_t row_begin, uint64_t row_end, int kmer_size,
                                double kmer_space, double max_distance, double max_p_value, mg_result *out_host,
                                uint64_t capacity, uint64_t *count_out)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!t || !count_out || (!out_host && capacity)) return fail(ctx, MG_ERR_INVALID, "mg_compare_tri_results_host: NULL argument");
    if (!t->lengths) return fail(ctx, MG_ERR_INVALID, "mg_compare_tri_results_host: the table carries no lengths");
    return compare_results(ctx, t, t, row_begin, row_end, true, kmer_size, kmer_space, max_distance, max_p_value, out_host, capacity, count_out);
}

int mg_compare_rect_results_host(mg_ctx *ctx, const mg_table *ref, const mg_table *qry, uint64_t q_begin, uint64_t q_end,
                                 int kmer_size, double kmer_space, double max_distance, double max_p_value, mg_result *out_host,
                                 uint64_t capacity, uint64_t *count_out)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!ref || !qry || !count_out || (!out_host && capacity))
        return fail(ctx, MG_ERR_INVALID, "mg_compare_rect_results_host: NULL argument");
    if (!ref->lengths || !qry->lengths) return fail(ctx, MG_ERR_INVALID, "mg_compare_rect_results_host: the tables carry no lengths");
    return compare_results(ctx, qry, ref, q_begin, q_end, false, kmer_size, kmer_space, max_distance, max_p_value, out_host, capacity, count_out);
}

/* ------------------------------------------------- several GPUs: communicator, replicated tables, row-block sharding */

// every pair is independent, so the all-pairs matrix is cut into row blocks,
// one per GPU, against a sketch table that is resident on every GPU.  The one exchange is the
// BROADCAST of that table from GPU 0 (RCCL over xGMI); the compare data path has no collective.
// Two shapes of the same thing:
//   local : one process drives every GPU (the `mash` CLI): a context per device, ncclCommInitAll;
//   rank  : one process per GPU (bench.py under torchrun): ncclCommInitRank on an id the caller
//           hands round (128 bytes, any transport).
struct mg_comm {
    bool local = false;
    int nranks = 1, rank = 0;                 // rank mode: this process; local mode: rank is unused
    std::vector<mg_ctx *> ctxs;               // local: one per device, owned; rank: the caller's context
    std::vector<ncclComm_t> comms;            // local: one per device; rank: one; empty = no RCCL (see below)
    std::string err;
};

struct mg_dtable {
    mg_comm *comm = nullptr;
    std::vector<mg_table *> t;                // one replica per context of the communicator -- or, row-sharded
                                              // (mg_dtable_upload_rows), context g's rows [row0[g], row0[g + 1])
    bool by_rows = false;
    std::vector<uint64_t> row0;               // row-sharded: G + 1 boundaries
    uint64_t n = 0, s = 0;
    // replicated tables compared by reference rows: views of a replica's row slice, kept for their caches
    struct View { size_t g; uint64_t lo, hi; mg_table *t; };
    mutable std::vector<View> views;
    mutable std::mutex views_mu;
};

static int comm_fail(mg_comm *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_error = msg;
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
const char *mg_comm_last_error(mg_comm *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

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

static int comm_sync_all(mg_comm *c)
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

extern "C++" {
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
}  // extern "C++"

static int dtable_check(mg_comm *c, const mg_dtable *t, const char *who, bool rows_ok = false)
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

extern "C++" {
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
}  // extern "C++"

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
extern "C++" {
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
}  // extern "C++"

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

/* ------------------------------------------------------------------ screening */

struct mg_screen {
    mg_ctx *ctx = nullptr;
    mg_params p;
    const mg_table *db = nullptr;
    unsigned long long *keys = nullptr;
    uint32_t *obs = nullptr;
    uint64_t slots = 0;
    uint64_t key_max = 0;
    bool translate = false;             // mixture is nucleotide, queries are amino-acid sketches
    std::vector<uint64_t> mix;          // running bottom-s of the mixture (host, ascending, distinct)
    uint64_t distinct = 0;              // distinct hashes of the database (counted while the table is built)
    // what a job touched: slots whose counter left 0 (device list), so that results and reset are O(touched)
    uint32_t *touched = nullptr;
    unsigned long long *ntouched = nullptr;      // device; [1] = cursor of the hit list
    uint64_t touched_cap = 0;
    // rows by slot (built at the first sparse finish): slot_end[slot] = end of its run in ent
    uint32_t *slot_end = nullptr, *ent = nullptr;
    // second tier of the key bound (SketchArgs::probe_tier): keys above `tier` are announced by a bitmap
    uint64_t tier = 0, bits_scale = 0;
    uint32_t *bits = nullptr;
    std::string tier_note;
};

// Two-tier key bound.  A k-mer hash above the table's largest key cannot be a key; that bound is only as
// good as the database's SMALLEST genome (bottom-s hashes of a 30 kbp virus reach 1/30 of the hash range, those
// of a 5 Mbp bacterium 1/5000).  So the range is cut at `tier`: below it a hash goes to the table as before,
// above it only if its bit in a bitmap over (tier, key_max] is set.  tier = the candidate (key_max / 2^j)
// with the least expected cost per k-mer, a table probe counting 1 and a bitmap read 0.15.
static int screen_plan_tiers(mg_ctx *ctx, mg_screen *sc)
{
    sc->tier = sc->key_max;
    if (sc->distinct == 0 || sc->key_max < (1ull << 40)) return MG_OK;
    // (a bound that already spares all but a few k-mers in a thousand needs no second tier: C4's database of
    //  like-sized genomes sends 0.1 % of the mixture's k-mers to the table)
    if ((double)sc->key_max / 18446744073709551616.0 < 0.004 && !ctx_opt(ctx, "MASHGPU_SCREEN_TIERS")) {
        sc->tier_note = "one tier (the largest key already spares all but a few k-mers in a thousand)";
        return MG_OK;
    }
    if (const char *e = ctx_opt(ctx, "MASHGPU_SCREEN_TIERS")) { if (atoi(e) == 0) { sc->tier_note = "off (MASHGPU_SCREEN_TIERS=0)"; return MG_OK; } }
    uint32_t log_bits = 27;
    if (const char *e = ctx_opt(ctx, "MASHGPU_SCREEN_BITS")) log_bits = (uint32_t)std::min(34, std::max(10, atoi(e)));
    const uint64_t B = 1ull << log_bits;
    const uint32_t NB = 24;
    std::vector<uint64_t> bounds(NB);
    for (uint32_t j = 0; j < NB; j++) bounds[j] = sc->key_max >> (j + 1);
    DevBuf<uint64_t> d_bounds(ctx);
    DevBuf<unsigned long long> d_below(ctx);
    std::vector<unsigned long long> below(NB, 0);
    if (d_bounds.alloc(NB) != hipSuccess || d_below.alloc(NB) != hipSuccess) return fail(ctx, MG_ERR_NOMEM, "mg_screen_create: device allocation failed");
    hipError_t e = hipMemcpyAsync(d_bounds, bounds.data(), NB * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_below, 0, NB * 8, ctx->stream);
    if (e == hipSuccess) e = mg::launch_screen_count_below(sc->keys, sc->slots, d_bounds, NB, d_below, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(below.data(), d_below, NB * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("mg_screen_create (tiers): ") + hipGetErrorString(e));
    const double R = 18446744073709551616.0, one = (double)sc->key_max / R;
    double best = one;
    int best_j = -1;
    for (uint32_t j = 0; j < NB; j++) {
        const double t = (double)bounds[j] / R, above = (double)(sc->distinct - below[j]);
        const double cost = t + (one - t) * (0.15 + std::min(1.0, above / (double)B));
        if (cost < best) { best = cost; best_j = (int)j; }
    }
    char note[200];
    if (best_j < 0 || best > 0.8 * one) {
        snprintf(note, sizeof note, "one tier (key bound %.3g of the hash range, best two-tier cost %.3g)", one, best);
        sc->tier_note = note;
        return MG_OK;
    }
    const uint64_t tier = bounds[best_j], range = sc->key_max - tier;
    if (range < 2 * B) return MG_OK;
    sc->bits_scale = (uint64_t)(((unsigned __int128)B << 64) / range);
    if (hipMalloc(&sc->bits, B / 8) != hipSuccess) return fail(ctx, MG_ERR_NOMEM, "mg_screen_create: device allocation failed");
    e = hipMemsetAsync(sc->bits, 0, B / 8, ctx->stream);
    if (e == hipSuccess) e = mg::launch_screen_bits(sc->keys, sc->slots, tier, sc->bits_scale, sc->bits, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("mg_screen_create (tiers): ") + hipGetErrorString(e));
    sc->tier = tier;
    snprintf(note, sizeof note, "two tiers: table below %.3g of the hash range, %llu keys behind a %u-bit bitmap up to %.3g (cost %.3g -> %.3g)",
             (double)tier / R, (unsigned long long)(sc->distinct - below[best_j]), log_bits, one, one, best);
    sc->tier_note = note;
    return MG_OK;
}

int mg_screen_create(mg_ctx *ctx, const mg_params *p, const mg_table *db, mg_screen **out)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!p || !db || !out) return fail(ctx, MG_ERR_INVALID, "mg_screen_create: NULL argument");
    const bool dna = alphabet_is_dna(p);
    if (!p->noncanonical && !dna) return fail(ctx, MG_ERR_UNSUPPORTED, "mg_screen: canonical k-mers need the ACGT alphabet");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mg_screen *sc = new mg_screen;
    sc->ctx = ctx;
    sc->p = *p;
    sc->db = db;
    {
        const int rc = table_max(ctx, db, &sc->key_max);
        if (rc != MG_OK) { delete sc; return rc; }
    }
    uint64_t slots = 1024;
    while (slots < 2 * db->n * db->s) slots <<= 1;
    sc->slots = slots;
    // (slots are addressed with 32 bits in the touched list and the rows-by-slot index: at most 2^32 slots = 2^31 hashes.
    //  A larger database keeps the dense results -- mg_screen_counts_dev / mg_screen_finish_host, which need neither --
    //  and has no touched list: the sparse results and the O(touched) reset are refused for it)
    const bool listed = db->n * db->s <= (1ull << 31);
    sc->touched_cap = listed ? std::max<uint64_t>(db->n * db->s, 1) : 0;
    hipError_t e = hipMalloc(&sc->keys, slots * 8);
    if (e == hipSuccess) e = hipMalloc(&sc->obs, slots * 4);
    if (e == hipSuccess && listed) e = hipMalloc(&sc->touched, sc->touched_cap * 4);
    if (e == hipSuccess) e = hipMalloc(&sc->ntouched, 16);
    if (e == hipSuccess) e = hipMemsetAsync(sc->keys, 0xFF, slots * 8, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(sc->obs, 0, slots * 4, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(sc->ntouched, 0, 16, ctx->stream);
    if (e == hipSuccess) e = mg::launch_screen_build(db->hashes, db->nhash, db->n, db->s, sc->keys, slots - 1, sc->ntouched + 1, ctx->stream);
    unsigned long long distinct = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&distinct, sc->ntouched + 1, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        mg_screen_free(sc);
        return fail(ctx, MG_ERR_HIP, std::string("mg_screen_create: ") + hipGetErrorString(e));
    }
    sc->distinct = distinct;
    {
        const int rc = screen_plan_tiers(ctx, sc);
        if (rc != MG_OK) { mg_screen_free(sc); return rc; }
    }
    *out = sc;
    return MG_OK;
}

int mg_screen_create_translated(mg_ctx *ctx, const mg_params *p, const mg_table *db, mg_screen **out)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!p || !out) return fail(ctx, MG_ERR_INVALID, "mg_screen_create_translated: NULL argument");
    if (!p->noncanonical || alphabet_is_dna(p))
        return fail(ctx, MG_ERR_INVALID, "mg_screen_create_translated: needs an amino-acid (noncanonical) alphabet");
    const int rc = mg_screen_create(ctx, p, db, out);
    if (rc == MG_OK) (*out)->translate = true;
    return rc;
}

static int screen_add_translated(mg_screen *sc, const uint8_t *bases_dev, uint64_t nbases);

int mg_screen_add_dev(mg_screen *sc, const uint8_t *bases_dev, uint64_t nbases)
{
    if (!sc) return MG_ERR_INVALID;
    mg_ctx *ctx = sc->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (sc->translate) return screen_add_translated(sc, bases_dev, nbases);
    const uint64_t k = (uint64_t)sc->p.kmer_size;
    if (nbases < k) return MG_OK;
    if (((uintptr_t)bases_dev & 15) != 0) return fail(ctx, MG_ERR_INVALID, "mg_screen_add: bases must be 16-byte aligned");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // one pass over the batch: its bottom-s (folded into the mixture's running sketch) and,
    // fused into the same kernel, the table probe of every k-mer (hashCounts[key]++)
    const uint64_t s = sc->p.sketch_size;
    uint64_t *d_h = nullptr;
    uint32_t *d_n = nullptr;
    HIP_TRY(ctx, hipMalloc(&d_h, s * 8));
    if (hipMalloc(&d_n, 4) != hipSuccess) { hipFree(d_h); return fail(ctx, MG_ERR_NOMEM, "mg_screen_add: allocation failed"); }
    const uint64_t off[2] = {0, nbases};
    const ProbeHook hook{sc->keys, sc->obs, sc->slots - 1, sc->key_max, sc->touched, sc->ntouched, sc->touched_cap, sc->tier, sc->bits, sc->bits_scale};
    int rc = sketch_dev_impl(ctx, &sc->p, bases_dev, nbases, off, 1, d_h, d_n, nullptr, &hook);
    std::vector<uint64_t> bh(s);
    uint32_t bn = 0;
    if (rc == MG_OK) {
        if (hipMemcpyAsync(bh.data(), d_h, s * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(&bn, d_n, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = fail(ctx, MG_ERR_HIP, "mg_screen_add: D2H copy failed");
    }
    hipFree(d_h);
    hipFree(d_n);
    if (rc != MG_OK) return rc;
    std::vector<uint64_t> merged;
    merged.reserve(sc->mix.size() + bn);
    std::merge(sc->mix.begin(), sc->mix.end(), bh.begin(), bh.begin() + bn, std::back_inserter(merged));
    merged.erase(std::unique(merged.begin(), merged.end()), merged.end());
    if (merged.size() > s) merged.resize(s);
    sc->mix.swap(merged);
    return MG_OK;
}

// amino-acid queries: translate the nucleotide batch in six frames on the device, then run the
// ordinary (table-alphabet, forward-only) pass over the translated bytes
static int screen_add_translated(mg_screen *sc, const uint8_t *bases_dev, uint64_t nbases)
{
    mg_ctx *ctx = sc->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (nbases < 3) return MG_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t seg = (nbases / 3 + 1 + 15) & ~15ull;        // >= one separator byte after every frame
    uint8_t *d_aa = nullptr;
    HIP_TRY(ctx, hipMalloc(&d_aa, 6 * seg + 64));
    hipError_t e = mg::launch_translate6(bases_dev, nbases, d_aa, seg, !sc->p.preserve_case, ctx->stream);
    int rc = MG_OK;
    if (e != hipSuccess) rc = fail(ctx, MG_ERR_HIP, std::string("mg_screen_add (translate): ") + hipGetErrorString(e));
    if (rc == MG_OK) {
        sc->translate = false;                                   // the translated bytes take the plain path
        rc = mg_screen_add_dev(sc, d_aa, 6 * seg);
        sc->translate = true;
    }
    hipStreamSynchronize(ctx->stream);
    hipFree(d_aa);
    return rc;
}

int mg_screen_add_host(mg_screen *sc, const uint8_t *bases, uint64_t nbases)
{
    if (!sc) return MG_ERR_INVALID;
    mg_ctx *ctx = sc->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!bases && nbases) return fail(ctx, MG_ERR_INVALID, "mg_screen_add_host: NULL bases");
    if (nbases == 0) return MG_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint8_t *d = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, nbases + 64));
    int rc = MG_OK;
    if (hipMemcpyAsync(d, bases, nbases, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = fail(ctx, MG_ERR_HIP, "mg_screen_add_host: H2D copy failed");
    else
        rc = mg_screen_add_dev(sc, d, nbases);
    hipStreamSynchronize(ctx->stream);
    hipFree(d);
    return rc;
}

int mg_screen_counts_dev(mg_screen *sc, uint32_t *counts_out_dev)
{
    if (!sc) return MG_ERR_INVALID;
    mg_ctx *ctx = sc->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!counts_out_dev) return fail(ctx, MG_ERR_INVALID, "mg_screen_counts_dev: NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (sc->db->n * sc->db->s == 0) return MG_OK;
    hipError_t e = mg::launch_screen_gather(sc->db->hashes, sc->db->nhash, sc->db->n, sc->db->s, sc->keys, sc->obs,
                                            sc->slots - 1, counts_out_dev, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("mg_screen_counts_dev: ") + hipGetErrorString(e));
    return MG_OK;
}

int mg_screen_finish_host(mg_screen *sc, uint32_t *counts_out, uint64_t *mix_hashes_out, uint32_t *mix_nhash_out,
                          uint64_t *distinct_out)
{
    if (!sc) return MG_ERR_INVALID;
    mg_ctx *ctx = sc->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t total = sc->db->n * sc->db->s;
    if (counts_out && total) {
        uint32_t *d = nullptr;
        HIP_TRY(ctx, hipMalloc(&d, total * 4));
        hipError_t e = mg::launch_screen_gather(sc->db->hashes, sc->db->nhash, sc->db->n, sc->db->s, sc->keys, sc->obs,
                                                sc->slots - 1, d, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(counts_out, d, total * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        hipFree(d);
        if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("mg_screen_finish: ") + hipGetErrorString(e));
    }
    const uint64_t s = sc->p.sketch_size;
    if (mix_hashes_out) {
        for (uint64_t i = 0; i < s; i++) mix_hashes_out[i] = i < sc->mix.size() ? sc->mix[i] : MG_HASH_PAD;
    }
    if (mix_nhash_out) *mix_nhash_out = (uint32_t)sc->mix.size();
    if (distinct_out) *distinct_out = sc->distinct;         // counted while the table was built
    return MG_OK;
}

static int screen_touched(mg_screen *sc, uint64_t *nt)
{
    unsigned long long v = 0;
    if (hipMemcpyAsync(&v, sc->ntouched, 8, hipMemcpyDeviceToHost, sc->ctx->stream) != hipSuccess || hipStreamSynchronize(sc->ctx->stream) != hipSuccess)
        return fail(sc->ctx, MG_ERR_HIP, "mg_screen: D2H copy failed");
    *nt = std::min<uint64_t>(v, sc->touched_cap);
    return MG_OK;
}

int mg_screen_reset(mg_screen *sc)
{
    if (!sc) return MG_ERR_INVALID;
    mg_ctx *ctx = sc->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!sc->touched) return fail(ctx, MG_ERR_UNSUPPORTED, "mg_screen_reset: databases of more than 2^31 hashes keep no touched list (create a new screen)");
    uint64_t nt = 0;
    const int rc = screen_touched(sc, &nt);
    if (rc != MG_OK) return rc;
    hipError_t e = mg::launch_screen_reset(sc->touched, nt, sc->obs, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(sc->ntouched, 0, 8, ctx->stream);
    if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("mg_screen_reset: ") + hipGetErrorString(e));
    sc->mix.clear();
    return MG_OK;
}

const char *mg_screen_tier_note(const mg_screen *sc) { return sc ? sc->tier_note.c_str() : ""; }

// rows by slot, once per database
static int screen_ensure_index(mg_screen *sc)
{
    if (sc->slot_end) return MG_OK;
    mg_ctx *ctx = sc->ctx;
    const mg_table *db = sc->db;
    const size_t tb = mg::screen_index_temp_bytes(sc->slots);
    DevBuf<uint8_t> temp(ctx);
    hipError_t e = hipMalloc(&sc->slot_end, sc->slots * 4);
    if (e == hipSuccess) e = hipMalloc(&sc->ent, std::max<uint64_t>(db->n * db->s, 1) * 4);
    if (e == hipSuccess) e = temp.alloc(std::max<size_t>(tb, 1));
    if (e == hipSuccess) e = hipMemsetAsync(sc->slot_end, 0, sc->slots * 4, ctx->stream);
    if (e == hipSuccess) e = mg::launch_screen_index(db->hashes, db->nhash, db->n, db->s, sc->keys, sc->slots - 1, sc->slot_end, sc->ent, temp, tb, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        if (sc->slot_end) hipFree(sc->slot_end);
        if (sc->ent) hipFree(sc->ent);
        sc->slot_end = sc->ent = nullptr;
        return fail(ctx, MG_ERR_HIP, std::string("mg_screen (index): ") + hipGetErrorString(e));
    }
    return MG_OK;
}

int mg_screen_finish_sparse_host(mg_screen *sc, mg_screen_hit *hits_out, uint64_t capacity, uint64_t *nhits_out,
                                 uint64_t *mix_hashes_out, uint32_t *mix_nhash_out, uint64_t *distinct_out)
{
    if (!sc) return MG_ERR_INVALID;
    mg_ctx *ctx = sc->ctx;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!nhits_out || (!hits_out && capacity)) return fail(ctx, MG_ERR_INVALID, "mg_screen_finish_sparse_host: NULL argument");
    if (!sc->touched) return fail(ctx, MG_ERR_UNSUPPORTED, "mg_screen_finish_sparse_host: databases of more than 2^31 hashes have dense results only (mg_screen_finish_host)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = screen_ensure_index(sc);
    uint64_t nt = 0;
    if (rc == MG_OK) rc = screen_touched(sc, &nt);
    if (rc != MG_OK) return rc;
    unsigned long long total = 0;
    if (nt) {
        hipError_t e = hipMemsetAsync(sc->ntouched + 1, 0, 8, ctx->stream);
        if (e == hipSuccess) e = mg::launch_screen_hits(sc->touched, nt, sc->keys, sc->obs, sc->slot_end, sc->ent, nullptr, sc->ntouched + 1, 0, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&total, sc->ntouched + 1, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("mg_screen_finish_sparse: ") + hipGetErrorString(e));
    }
    *nhits_out = total;
    const uint64_t take = std::min<uint64_t>(total, capacity);
    if (take) {
        DevBuf<mg::ScreenHit> d_hits(ctx);
        if (d_hits.alloc(total) != hipSuccess) return fail(ctx, MG_ERR_NOMEM, "mg_screen_finish_sparse: device allocation failed");
        hipError_t e = hipMemsetAsync(sc->ntouched + 1, 0, 8, ctx->stream);
        if (e == hipSuccess) e = mg::launch_screen_hits(sc->touched, nt, sc->keys, sc->obs, sc->slot_end, sc->ent, d_hits, sc->ntouched + 1, total, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("mg_screen_finish_sparse: ") + hipGetErrorString(e));
        // in a defined order: by row, then hash (the kernel emits them in the order the slots were touched)
        DevBuf<mg::ScreenHit> d_sorted(ctx);
        DevBuf<unsigned long long> k64a(ctx), k64b(ctx);
        DevBuf<uint32_t> u32a(ctx), u32b(ctx), u32c(ctx), u32d(ctx);
        DevBuf<uint8_t> temp(ctx);
        const size_t tb = mg::screen_sort_temp_bytes(total);
        if (d_sorted.alloc(total) != hipSuccess || k64a.alloc(total) != hipSuccess || k64b.alloc(total) != hipSuccess || u32a.alloc(total) != hipSuccess ||
            u32b.alloc(total) != hipSuccess || u32c.alloc(total) != hipSuccess || u32d.alloc(total) != hipSuccess || temp.alloc(std::max<size_t>(tb, 1)) != hipSuccess)
            return fail(ctx, MG_ERR_NOMEM, "mg_screen_finish_sparse: device allocation failed");
        e = mg::launch_screen_sort_hits(d_hits, total, d_sorted, k64a, k64b, u32a, u32b, u32c, u32d, temp, tb, ctx->stream);
        static_assert(sizeof(mg_screen_hit) == sizeof(mg::ScreenHit), "mg_screen_hit layout");
        if (e == hipSuccess) e = hipMemcpyAsync(hits_out, d_sorted, take * sizeof(mg_screen_hit), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("mg_screen_finish_sparse: ") + hipGetErrorString(e));
    }
    const uint64_t s = sc->p.sketch_size;
    if (mix_hashes_out) for (uint64_t i = 0; i < s; i++) mix_hashes_out[i] = i < sc->mix.size() ? sc->mix[i] : MG_HASH_PAD;
    if (mix_nhash_out) *mix_nhash_out = (uint32_t)sc->mix.size();
    if (distinct_out) *distinct_out = sc->distinct;
    return MG_OK;
}

double mg_identity(uint64_t common, uint64_t denom, int kmer_size)
{
    if (common == denom) return 1.;                       // avoid -0
    if (common == 0) return 0.;                           // avoid inf
    return pow((double)common / (double)denom, 1. / kmer_size);
}

double mg_p_value_within(uint64_t x, uint64_t set_size, double kmer_space, uint64_t sketch_size)
{
    if (x == 0) return 1.;
    const double r = (double)set_size / kmer_space;
    return mg::binomial_q(x - 1, r, sketch_size);
}

void mg_screen_free(mg_screen *sc)
{
    if (!sc) return;
    hipSetDevice(sc->ctx->device);
    for (void *q : {(void *)sc->keys, (void *)sc->obs, (void *)sc->touched, (void *)sc->ntouched, (void *)sc->slot_end, (void *)sc->ent, (void *)sc->bits})
        if (q) hipFree(q);
    delete sc;
}

/* ------------------------------------------------- screening on several GPUs (local communicator) */

// The mixture is sharded by BATCH: batch b goes to device b mod G, which screens it against its own
// replica of the query table on its own host thread while the caller parses the next batch; the
// one exchange is the sum of the per-hash observation counters at the end (ncclReduce to GPU 0
// over xGMI, or host adds when the communicator has no RCCL) plus the merge of the G mixture
// sketches (bottom-s of their union).  Same results as one mg_screen fed every batch.
struct mg_dscreen {
    mg_comm *comm = nullptr;
    const mg_dtable *db = nullptr;
    std::vector<mg_screen *> sc;
    struct Worker {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        std::vector<uint8_t> buf;
        bool busy = false, stop = false;
        int rc = MG_OK;
    };
    std::vector<std::unique_ptr<Worker>> w;
    unsigned next = 0;
};

int mg_dscreen_create(mg_comm *c, const mg_params *p, const mg_dtable *db, int translated, mg_dscreen **out)
{
    int rc = dtable_check(c, db, "mg_dscreen_create");
    if (rc != MG_OK) return rc;
    if (!p || !out) return comm_fail(c, MG_ERR_INVALID, "mg_dscreen_create: NULL argument");
    mg_dscreen *d = new mg_dscreen;
    d->comm = c;
    d->db = db;
    const size_t G = c->ctxs.size();
    for (size_t g = 0; g < G; g++) {
        mg_screen *s1 = nullptr;
        rc = translated ? mg_screen_create_translated(c->ctxs[g], p, db->t[g], &s1) : mg_screen_create(c->ctxs[g], p, db->t[g], &s1);
        if (rc != MG_OK) { c->err = c->ctxs[g]->err; mg_dscreen_free(d); return rc; }
        d->sc.push_back(s1);
    }
    for (size_t g = 0; g < G; g++) {
        d->w.emplace_back(new mg_dscreen::Worker);
        mg_dscreen::Worker *wk = d->w.back().get();
        mg_screen *s1 = d->sc[g];
        wk->th = std::thread([wk, s1]() {
            std::unique_lock<std::mutex> lk(wk->m);
            for (;;) {
                wk->cv.wait(lk, [&] { return wk->busy || wk->stop; });
                if (wk->stop && !wk->busy) return;
                lk.unlock();
                const int r = mg_screen_add_host(s1, wk->buf.data(), wk->buf.size());
                lk.lock();
                if (r != MG_OK && wk->rc == MG_OK) wk->rc = r;
                wk->busy = false;
                wk->cv.notify_all();
            }
        });
    }
    *out = d;
    return MG_OK;
}

// hands one batch (records separated by MG_RECORD_SEP) to the next device; returns once the bytes are
// copied (the caller's buffer is free again), not when the batch is screened
int mg_dscreen_add_host(mg_dscreen *d, const uint8_t *bases, uint64_t nbases)
{
    if (!d) return MG_ERR_INVALID;
    if (!bases && nbases) return comm_fail(d->comm, MG_ERR_INVALID, "mg_dscreen_add_host: NULL bases");
    if (nbases == 0) return MG_OK;
    mg_dscreen::Worker *wk = d->w[d->next++ % d->w.size()].get();
    std::unique_lock<std::mutex> lk(wk->m);
    wk->cv.wait(lk, [&] { return !wk->busy; });
    if (wk->rc != MG_OK) return wk->rc;
    wk->buf.assign(bases, bases + nbases);
    wk->busy = true;
    wk->cv.notify_all();
    return MG_OK;
}

int mg_dscreen_finish_host(mg_dscreen *d, uint32_t *counts_out, uint64_t *mix_hashes_out, uint32_t *mix_nhash_out,
                           uint64_t *distinct_out)
{
    if (!d) return MG_ERR_INVALID;
    mg_comm *c = d->comm;
    const size_t G = d->sc.size();
    for (size_t g = 0; g < G; g++) {                                      // drain the workers
        mg_dscreen::Worker *wk = d->w[g].get();
        std::unique_lock<std::mutex> lk(wk->m);
        wk->cv.wait(lk, [&] { return !wk->busy; });
        if (wk->rc != MG_OK) { c->err = c->ctxs[g]->err; return wk->rc; }
    }
    const uint64_t total = d->db->t[0]->n * d->db->t[0]->s;
    const uint64_t s = d->sc[0]->p.sketch_size;
    // device 0 delivers its own counts, the distinct-hash number and (below) receives the others' counts
    std::vector<uint64_t> mix0(s);
    uint32_t mn0 = 0;
    if (G == 1) return mg_screen_finish_host(d->sc[0], counts_out, mix_hashes_out, mix_nhash_out, distinct_out);
    int rc = mg_screen_finish_host(d->sc[0], nullptr, mix0.data(), &mn0, distinct_out);
    if (rc != MG_OK) { c->err = c->ctxs[0]->err; return rc; }
    std::vector<uint64_t> merged(mix0.begin(), mix0.begin() + mn0);
    for (size_t g = 1; g < G; g++) {                                      // mixture sketch: bottom-s of the union
        const std::vector<uint64_t> &m = d->sc[g]->mix;
        std::vector<uint64_t> u;
        u.reserve(merged.size() + m.size());
        std::merge(merged.begin(), merged.end(), m.begin(), m.end(), std::back_inserter(u));
        u.erase(std::unique(u.begin(), u.end()), u.end());
        if (u.size() > s) u.resize(s);
        merged.swap(u);
    }
    if (mix_hashes_out) for (uint64_t i = 0; i < s; i++) mix_hashes_out[i] = i < merged.size() ? merged[i] : MG_HASH_PAD;
    if (mix_nhash_out) *mix_nhash_out = (uint32_t)merged.size();
    if (!counts_out || total == 0) return MG_OK;
    // observation counters: sum over the devices
    std::vector<uint32_t *> bufs(G, nullptr);
    auto release = [&]() { for (size_t g = 0; g < G; g++) if (bufs[g]) { hipSetDevice(c->ctxs[g]->device); hipFree(bufs[g]); } };
    for (size_t g = 0; g < G; g++) {
        if (hipSetDevice(c->ctxs[g]->device) != hipSuccess || hipMalloc(&bufs[g], total * 4) != hipSuccess) {
            release();
            return comm_fail(c, MG_ERR_NOMEM, "mg_dscreen_finish_host: device allocation failed");
        }
        rc = mg_screen_counts_dev(d->sc[g], bufs[g]);
        if (rc != MG_OK) { c->err = c->ctxs[g]->err; release(); return rc; }
    }
    if (!c->comms.empty()) {
        ncclResult_t r = ncclGroupStart();
        for (size_t g = 0; g < G && r == ncclSuccess; g++)
            r = ncclReduce(bufs[g], bufs[g], total, ncclUint32, ncclSum, 0, c->comms[g], c->ctxs[g]->stream);
        const ncclResult_t r2 = ncclGroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) { release(); return comm_fail(c, MG_ERR_HIP, std::string("ncclReduce: ") + ncclGetErrorString(r)); }
        rc = comm_sync_all(c);
        if (rc == MG_OK && (hipSetDevice(c->ctxs[0]->device) != hipSuccess ||
                            hipMemcpy(counts_out, bufs[0], total * 4, hipMemcpyDeviceToHost) != hipSuccess))
            rc = comm_fail(c, MG_ERR_HIP, "mg_dscreen_finish_host: D2H copy failed");
    } else {
        std::vector<uint32_t> part(total);
        memset(counts_out, 0, total * 4);
        for (size_t g = 0; g < G && rc == MG_OK; g++) {
            if (hipSetDevice(c->ctxs[g]->device) != hipSuccess || hipMemcpy(part.data(), bufs[g], total * 4, hipMemcpyDeviceToHost) != hipSuccess)
                rc = comm_fail(c, MG_ERR_HIP, "mg_dscreen_finish_host: D2H copy failed");
            else
                for (uint64_t i = 0; i < total; i++) counts_out[i] += part[i];
        }
    }
    release();
    return rc;
}

static int dscreen_drain(mg_dscreen *d)
{
    for (size_t g = 0; g < d->sc.size(); g++) {
        mg_dscreen::Worker *wk = d->w[g].get();
        std::unique_lock<std::mutex> lk(wk->m);
        wk->cv.wait(lk, [&] { return !wk->busy; });
        if (wk->rc != MG_OK) { d->comm->err = d->comm->ctxs[g]->err; return wk->rc; }
    }
    return MG_OK;
}

int mg_dscreen_finish_sparse_host(mg_dscreen *d, mg_screen_hit *hits_out, uint64_t capacity, uint64_t *nhits_out,
                                  uint64_t *mix_hashes_out, uint32_t *mix_nhash_out, uint64_t *distinct_out)
{
    if (!d) return MG_ERR_INVALID;
    mg_comm *c = d->comm;
    if (!nhits_out || (!hits_out && capacity)) return comm_fail(c, MG_ERR_INVALID, "mg_dscreen_finish_sparse_host: NULL argument");
    int rc = dscreen_drain(d);
    if (rc != MG_OK) return rc;
    const size_t G = d->sc.size();
    if (G == 1) {
        rc = mg_screen_finish_sparse_host(d->sc[0], hits_out, capacity, nhits_out, mix_hashes_out, mix_nhash_out, distinct_out);
        if (rc != MG_OK) c->err = c->ctxs[0]->err;
        return rc;
    }
    const uint64_t s = d->sc[0]->p.sketch_size;
    std::vector<mg_screen_hit> all, part, next;
    std::vector<uint64_t> merged;
    auto before = [](const mg_screen_hit &a, const mg_screen_hit &b) { return a.row != b.row ? a.row < b.row : a.hash < b.hash; };
    for (size_t g = 0; g < G; g++) {
        uint64_t n = 0;
        rc = mg_screen_finish_sparse_host(d->sc[g], nullptr, 0, &n, nullptr, nullptr, g == 0 ? distinct_out : nullptr);
        part.resize(n);
        if (rc == MG_OK && n) rc = mg_screen_finish_sparse_host(d->sc[g], part.data(), n, &n, nullptr, nullptr, nullptr);
        if (rc != MG_OK) { c->err = c->ctxs[g]->err; return rc; }
        next.resize(all.size() + part.size());                             // every device's list is ordered: a linear merge
        std::merge(all.begin(), all.end(), part.begin(), part.end(), next.begin(), before);
        all.swap(next);
        const std::vector<uint64_t> &m = d->sc[g]->mix;                 // mixture sketch: bottom-s of the union
        std::vector<uint64_t> u;
        u.reserve(merged.size() + m.size());
        std::merge(merged.begin(), merged.end(), m.begin(), m.end(), std::back_inserter(u));
        u.erase(std::unique(u.begin(), u.end()), u.end());
        if (u.size() > s) u.resize(s);
        merged.swap(u);
    }
    size_t w = 0;
    for (size_t i = 0; i < all.size(); i++) {
        if (w && all[w - 1].row == all[i].row && all[w - 1].hash == all[i].hash) all[w - 1].count += all[i].count;
        else all[w++] = all[i];
    }
    *nhits_out = w;
    if (capacity) memcpy(hits_out, all.data(), std::min<uint64_t>(w, capacity) * sizeof(mg_screen_hit));
    if (mix_hashes_out) for (uint64_t i = 0; i < s; i++) mix_hashes_out[i] = i < merged.size() ? merged[i] : MG_HASH_PAD;
    if (mix_nhash_out) *mix_nhash_out = (uint32_t)merged.size();
    return MG_OK;
}

int mg_dscreen_reset(mg_dscreen *d)
{
    if (!d) return MG_ERR_INVALID;
    int rc = dscreen_drain(d);
    for (size_t g = 0; g < d->sc.size() && rc == MG_OK; g++) {
        rc = mg_screen_reset(d->sc[g]);
        if (rc != MG_OK) d->comm->err = d->comm->ctxs[g]->err;
    }
    return rc;
}

void mg_dscreen_free(mg_dscreen *d)
{
    if (!d) return;
    for (auto &wk : d->w) {
        { std::lock_guard<std::mutex> lk(wk->m); wk->stop = true; }
        wk->cv.notify_all();
        if (wk->th.joinable()) wk->th.join();
    }
    for (mg_screen *s1 : d->sc) mg_screen_free(s1);
    delete d;
}

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
    for (auto *v : {&ctx->prof_compare, &ctx->prof_sketch, &ctx->prof_fill, &ctx->prof_discover, &ctx->prof_merge, &ctx->prof_index, &ctx->prof_dense}) {
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
    else if (strcmp(name, "compare_discover") == 0) v = &ctx->prof_discover;
    else if (strcmp(name, "compare_merge") == 0) v = &ctx->prof_merge;
    else if (strcmp(name, "compare_index") == 0) v = &ctx->prof_index;
    else if (strcmp(name, "compare_dense") == 0) v = &ctx->prof_dense;
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

}  /* extern "C" */
