// host_internal.h -- what the host-side translation units of libmashgpu.so share: the context (its stream, block pool,
// knobs, profiling records), the table with its cached derived structures, and the helpers every entry point uses.
// mashgpu.cpp      context, pool, parameters, tables, profiling
// host_sketch.cpp  sketching: batch, packed input, streamed sessions, reads mode
// host_compare.cpp comparing: tile engine, inverted-index engine, finishing, thresholded and list outputs
// host_comm.cpp    several GPUs: communicator, replicated / row-sharded tables, sharded calls
// host_screen.cpp  screening, on one GPU and on several
// The entry points of include/mashgpu.h get their C linkage from that header; everything declared here is C++.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <memory>
#include <iterator>
#include <functional>
#include <map>
#include <mutex>
#include <queue>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <system_error>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/mashgpu.h"
#include "compare_internal.h"
#include "finish_internal.h"
#include "pvalue.h"
#include "screen_internal.h"
#include "sketch_internal.h"

struct ProfRec { hipEvent_t a, b; };

// What the compare dispatch believes things cost (host_compare.cpp: SparseJobRun::join / choose_engine): seconds on one MI355X,
// each a measured number (profiles/r03_sparse_phases.txt; the dense pairs and the tile engine's price of a shared hash from
// rounds 4 and 5; the join engine's from round 6), written down in ONE place.  These are the DEFAULTS: every context keeps a
// copy and corrects it from the phases of its own jobs (SparseJobRun::learn).
struct SparseCosts {
    double fill_bytes_s = 4.5e12;            // the fill writes 8 B per pair
    double discover_per_shared = 2.0e-12;    // a run entry read by discovery
    double discover_per_entry = 4.0e-11;     // a row's entry looked at (n x s of them)
    double merge_per_candidate = 1.0e-9;     // a merge of ~2 s steps
    double launches = 2.0e-5;
    double class_bytes_s = 2.0e12;           // pairs inside classes of identical rows
    double dense_per_pair = 3.0e-11;         // a pair inside a dense group
    // the tile engine: pairs per second by job size, and what a shared hash costs it -- 2.2e-12 s between copies of one
    // sketch, 5.5e-12 inside clades, 1.2e-11 in a collection of one species, where every pair shares a few hundred values and
    // no two rows the same ones (round 5's one_species bracket: 1.58 s for 5.4e8 pairs where the model said 0.33); priced at
    // the upper middle, the copies and clades having engines of their own by now
    double tiles_rate_small = 8.0e9, tiles_rate_mid = 1.5e10, tiles_rate_large = 3.0e10;
    double tiles_per_shared = 8.0e-12;
    // the join engine (compare_join.hip): a counter update per (pair, shared value); an intersection step of 64 x 64 group ids
    // per tile (the lists' groups are bounded by their entries before the lists exist); the lists' sort per slot
    // (round 6, one species of 32 768 rows: 1.25e11 shared hashes in 37.5 ms all in, lists in family order)
    double join_per_shared = 2.2e-13;
    double join_per_step = 4.0e-11;
    double join_per_slot = 8.0e-11;
    double join_min_shared_per_pair = 4.0;   // below this many shared hashes per pair of the TABLE the engine is not even priced
};


struct mg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int cu_count = 0;
    std::string err;
    bool prof = false;
    std::vector<ProfRec> prof_compare, prof_sketch;
    // phases of the inverted-index compare engine (compare_sparse.hip), each its own kernel
    std::vector<ProfRec> prof_fill, prof_discover, prof_merge, prof_index, prof_dense, prof_join;
    std::vector<ProfRec> prof_fill_aside;                  // the fill beside the index build, on `aux` (SparseJobRun::prefill)
    void *pin = nullptr;                                    // ctx_pinned
    size_t pin_cap = 0;
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
    // the compare dispatch's prices, corrected from this context's own launches; the event pairs that time a phase
    SparseCosts costs;
    struct CostClock { hipEvent_t a = nullptr, b = nullptr; };
    CostClock cost_clk[5];
    uint64_t cost_updates = 0;
    // the constant fill of a matrix job beside the index build (host_compare.cpp: SparseJobRun::prefill)
    hipStream_t aux = nullptr;
    hipEvent_t aux_go = nullptr, aux_done = nullptr;
    uint32_t *aux_ctr = nullptr;                            // the next chunk of the output nobody has taken yet (two counters, 64 B apart)
    // set while a job's fill runs beside its index build: the build calls it when the table turns out to be nothing but
    // copies of one sketch of c hashes -- the fill's constant is {c, c} then, not {0, s}
    std::function<void(uint32_t)> aside_all_copies;
    // ... and this one when the bucket sorts (K4) are about to be queued: where the build is long, the fill starts there
    std::function<void()> aside_at_sort;
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
        std::vector<uint32_t> grp_of_host; // (the source of grp_of's upload: alive as long as the copy may be pending)
        uint32_t *grp_of = nullptr;        // [n] group of a row, 0xFFFFFFFF: none
        uint32_t *ulist = nullptr, *upos = nullptr;        // the groups' universes: values and the positions of their leaders
        unsigned long long *gdata = nullptr;                // the groups' blocks: masks, counts, the extras' bit planes (compare_internal.h)
        uint16_t *ext = nullptr;
        uint32_t dn_wmax = 0, dn_xs = 0;
        bool dn_lists = false;             // (test knob) every word resolved from the extras' lists instead of their masks
        // The index may be built on the table in ANOTHER ROW ORDER (rows that belong together next to each other, so that
        // they form dense groups whatever the order of the collection; compare_dense.hip: dense_cluster_rows): `clustered`
        // says this variant was asked for, inv != nullptr that the order differs -- index row a is table row inv[a], the
        // index reads the reordered copy `phashes`, and every kernel that writes results maps rows back.  Only the plain
        // full-triangle job uses it (row ranges, rect and list jobs address table rows and take the other variant).
        bool clustered = false;
        uint32_t split = 0;                // clustered: the rows from `split` on form a segment of their own (a triangle job over [split, n))
        bool by_tiles = false;             // built by index_build.hip (tiles + bucket sorts), not by the general sort
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
        // join engine (compare_join.hip): the block lists of the table's rows, built the first time a job takes that engine
        // (only_shared: values held by one row left out -- the triangle's variant; a rect job needs every entry)
        struct Join {
            bool built = false, only_shared = false;
            bool ordered = false;          // the lists stand on the rows in an order of their own (src / map): a triangle job over the
            uint32_t split = 0;            // table's last rows [split, n), those rows in a segment of their own behind the others
            mg::JoinSide side;
            const uint32_t *src = nullptr, *map = nullptr;
            void *bufs[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            double build_ms = 0;
        } jn;
        // what a (rows, range) job costs, learned by a counting pass the first time it is seen
        struct Plan { const void *rows; uint64_t rb, re; bool triangle; uint64_t cand, shared; bool use; bool use_list; bool join; uint32_t *order;
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
    // views of the table's first rows (table_prefix_view): a triangle job over rows [rb, re) only ever looks at rows below re,
    // so everything derived for it -- the inverted index above all -- is derived from the first re rows; owned by this table,
    // dropped with its derived data
    mutable std::vector<mg_table *> prefix_views;
};

#define HIP_TRY(ctx, call)                                                           \
    do {                                                                             \
        hipError_t e__ = (call);                                                     \
        if (e__ != hipSuccess) {                                                     \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);        \
            return MG_ERR_HIP;                                                       \
        }                                                                            \
    } while (0)


// ---- mashgpu.cpp
bool ctx_is_live(const void *c);
// Scratch of the entry points comes from a per-context cache of device blocks (see mashgpu.cpp)
void ctx_trim(mg_ctx *ctx);
hipError_t ctx_malloc(mg_ctx *ctx, void **out, size_t bytes);
// the first n rows of t as a table of their own (n < t->n; cached with t: at most four views, the oldest goes first)
const mg_table *table_prefix_view(const mg_table *t, uint64_t n);
void ctx_free(mg_ctx *ctx, void *p);

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
        if (p) { if (owner) ctx_free(owner, p); else hipFree(p); p = nullptr; }     // (a second alloc: the first block goes back)
        const size_t bytes = std::max<uint64_t>(count, 1) * sizeof(T);
        return owner ? ctx_malloc(owner, reinterpret_cast<void **>(&p), bytes) : hipMalloc(&p, bytes);
    }
    T *release() { T *q = p; p = nullptr; return q; }
    operator T *() const { return p; }
};

int fail(mg_ctx *ctx, int code, const std::string &msg);
// a knob: the context's own setting, else the environment's (nullptr: not set)
const char *ctx_opt(const mg_ctx *ctx, const char *name);
// host_index.cpp: the inverted index of table t for sketch size s (cached in the table; (*out)->usable says whether the engine
// can take it), in the table's own order or -- clustered -- on a copy with related rows next to each other
int table_sparse_index(mg_ctx *ctx, const mg_table *t, uint32_t s, bool clustered, mg_table::Sparse **out, uint32_t split = 0);
// The context's block of pinned host memory (grown on demand, at least `bytes`; nullptr: none to be had -- copy into pageable
// memory instead).  For read-backs that the host wants QUEUED, not waited for one by one: a copy into pageable memory
// returns when it is done, 40 us each behind an idle stream.  One user at a time (a call holds the context's lock).
void *ctx_pinned(mg_ctx *ctx, size_t bytes);
void prof_begin(mg_ctx *ctx, std::vector<ProfRec> &v, hipStream_t stream = nullptr);
void prof_end(mg_ctx *ctx, std::vector<ProfRec> &v, hipStream_t stream = nullptr);


// ---- host_sketch.cpp
bool alphabet_is_dna(const mg_params *p);
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

// packed bases the sketch kernel reads itself (SketchArgs::packed): base skip + o / mask bit mskip + o = byte o of the batch
struct PackedSrc { const uint32_t *packed, *mask; uint32_t skip, mskip; };
// the worker behind mg_sketch_dev / mg_sketch_host / the packed and screening paths (probe, packed: nullable; with `packed`
// bases_dev is unused -- plain sketches only: no multiplicities, no min_copies, sketch sizes of the LDS selector)
int sketch_dev_impl(mg_ctx *ctx, const mg_params *p, const uint8_t *bases_dev, uint64_t nbases, const uint64_t *sketch_off,
                    uint64_t nsketch, uint64_t *hashes_out_dev, uint32_t *nhash_out_dev, uint32_t *counts_out_dev,
                    const ProbeHook *probe, const PackedSrc *packed = nullptr);

// ---- host_compare.cpp
int table_max(mg_ctx *ctx, const mg_table *t, uint64_t *out);          // largest hash of a table (device reduction, cached)
uint64_t tri_pairs(uint64_t row_begin, uint64_t row_end);              // pairs of the triangle's rows [row_begin, row_end)

// ---- host_comm.cpp
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

int comm_fail(mg_comm *c, int code, const std::string &msg);
int dtable_check(mg_comm *c, const mg_dtable *t, const char *who, bool rows_ok = false);
int comm_sync_all(mg_comm *c);          // every context of a local communicator: wait for its stream
