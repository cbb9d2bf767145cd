// host_index.cpp -- the inverted index of a table for the compare engine (compare_sparse.hip, index_build.hip,
// compare_dense.hip): what host_compare.cpp's run_compare_sparse asks for once per table, sketch size and variant.
//
// Index of a table for sketch size s: every (value, row) entry of the rows' first min(nhash, s)
// hashes sorted by value, rows ascending inside a value.  Built once per table and sketch size, cached in the
// mg_table like the prefix images (dropped by mg_table_invalidate; its blocks then go back to the context's pool and
// the next table of the same shape takes them).  Retained: the sorted values (rect queries are located in them), the
// row per sorted position, the end of every group of equal values (kept at the group's first position), and two images
// in the table's own layout (row stride rs): the CODE image (2 x first sorted position of the entry's value -- ordered
// and equal exactly as the values are) and the POSITION image (the entry's own sorted position: the rows below it that
// hold the same value are sorted_rows[code / 2 .. position)).  A table the engine cannot take (2^31 entries and more, a
// real hash equal to the padding value, no memory) is marked unusable and keeps the tile engine.
//
// The build is one object (SparseIndexBuild) with one function per phase, run in this order:
//   scan_rows       everything the host must know about the rows is LAUNCHED (classes, copy suspects, clustering or
//                   neighbour links) and read back behind through one pinned block: the first of three waits;
//   order_rows      the first host pass over the rows (counts, densities), the table copied in clustered order (with
//                   the build's window offsets if the plan is known already), which neighbours form chains;
//   find_copies     only if there are suspects: rows verified value by value, classes of identical rows;
//   lay_out         offsets, short rows, the sort's bits, the tiles' plan, every buffer, the uploads;
//   build_by_tiles  index_build.hip in three stages with the candidate groups laid out in between: the second wait
//                   (an event behind the bucket sorts);  build_by_sort: rounds 3-4's way, for what the tiles refuse;
//   dense_groups    universes, layout and encoding of the groups of near-identical rows: the third wait.
#include "host_internal.h"
#include "index_build.h"
#ifdef IX_PHASE_CLOCKS
namespace mg { void index_dump_clocks(); }
#endif
#include "sort_bits.h"

namespace {

// One clade of many thousands of rows: every value of its pool has more holders than a bucket's LDS sort takes, so the whole
// index would go through the two-level sort of the big buckets -- correct, but measured 0.5 ms behind the general sort on the
// one-clade bracket (32 768 rows: 15.2 against 14.7 ms per table; clades of a thousand rows are the tiles': 16.9 against
// 19.1 on the 100 x 1 000 bracket).  The clustered order tells: rows of one label stand next to each other.
const uint64_t kTilesLongestRun = 6144;
static const uint64_t kSplitMinRun = 128;       // rows of the longest family in a segment from which a range job's index is reordered

uint64_t longest_label_run(const uint32_t *lab_sorted, uint64_t n)
{
    if (!lab_sorted || n == 0) return 0;
    uint64_t run = 1, longest = 1;
    for (uint64_t a = 1; a < n; a++) {
        run = lab_sorted[a] == lab_sorted[a - 1] ? run + 1 : 1;
        longest = std::max(longest, run);
    }
    return longest;
}

struct SparseIndexBuild {
    // ---- what is built, from what
    mg_ctx *ctx;
    const mg_table *t;
    const uint32_t s;
    const bool clustered;
    const uint32_t split;                                   // clustered: rows [split, n) in a segment of their own behind the others
    mg_table::Sparse *sp;
    const uint64_t n;
    const std::chrono::steady_clock::time_point t_begin = std::chrono::steady_clock::now();
    bool stop = false;                                      // a phase has settled the matter (the table keeps the tile engine)

    // ---- scan_rows: options, device scratch, what comes back
    bool need_classes = false, dedup = true, want_dense = true, try_cluster = false, try_links = false;
    DevBuf<uint8_t> dc_cls, d_link;
    DevBuf<unsigned long long> dc_last, d_dig, d_dig_sorted, k_a, k_b;
    DevBuf<uint32_t> d_cnt, d_rows_sorted, d_flags, d_nflag, r_a, r_b, l_a, l_b, d_inv, d_lab;
    DevBuf<unsigned char> d_tmp_dup, d_tmp_cl;
    std::vector<uint8_t> hc_cls, link;
    std::vector<uint64_t> hc_last;
    std::vector<uint32_t> hc_nh, inv_v, lab_v;
    const uint32_t *inv = nullptr, *lab_sorted = nullptr;   // the clustering's read-backs (pinned memory or the vectors); lab_sorted == nullptr: no labels
    uint32_t nflag = 0;
    size_t tb_dup = 0, tb_cl = 0;
    bool cluster_q = false, links_q = false;

    // ---- order_rows
    const char *ix_mode = nullptr;
    bool ix_verify = false, ix_tiles = true;
    std::vector<uint32_t> cnt_true, cnt_perm;
    uint32_t max_cnt = 0;
    uint64_t maxv = 0, E_all = 0;
    double dens[65] = {0};                                  // by bit length of a row's largest hash: values per unit of the hash range
    double dens0 = 0;                                       // ... and all of them: the density where the table is densest (below every row's largest hash)
    std::vector<uint32_t> rep;                              // empty: no copies
    const uint64_t *H = nullptr;                            // what the index is built from
    mg::IxPlan early_plan;                                  // the tiles' plan, if it could be made before the clustered copy ...
    DevBuf<unsigned char> d_lb;                             // ... whose kernel then left the rows' window offsets here
    bool lb_made = false;
    bool cnt_stale = false;                                 // d_cnt still holds the counts in the table's order

    // ---- find_copies
    std::vector<uint32_t> cls_of, cls_off, cls_rows, cls_first;

    // ---- lay_out
    uint32_t E = 0, end_bit = 0, sort_begin_bit = 0;
    std::vector<mg::DenseGroup> cand_groups;
    std::vector<uint32_t> grp_of_h;
    DevBuf<mg::DenseGroup> d_groups;
    DevBuf<uint32_t> d_grp_of, d_lead_rows, d_val, d_valj, d_cnt_sub, d_off_sub, d_nlead;
    DevBuf<unsigned long long> d_key, d_keyj;
    uint32_t lead_lists = 0, lead_tot[2] = {0, 0}, lead_cap = 0;
    bool lead_ready = false, lead_done = false, cand_prepared = false;
    mg::IxPlan plan;
    size_t temp_bytes = 0;
    DevBuf<unsigned char> temp;
    DevBuf<uint32_t> gs_of;
    DevBuf<unsigned long long> key64_a, key64_b;
    struct Stat { unsigned long long shared; uint32_t max_group, groups, bad, tie_overflow; uint32_t ixf[4]; } h_stat = {0, 0, 0, 0, 0, {0, 0, 0, 0}};
    DevBuf<Stat> d_stat;
    DevBuf<unsigned char> d_slots;
    bool want_order = true, ok = false;
    size_t nshort = 0;
    std::vector<uint32_t> short_cnt;
    hipError_t e = hipSuccess;

    // ---- build
    bool built = false;

    SparseIndexBuild(mg_ctx *c, const mg_table *tab, uint32_t sketch_size, bool clustered_variant, mg_table::Sparse *index, uint32_t split_row)
        : ctx(c), t(tab), s(sketch_size), clustered(clustered_variant), split(split_row), sp(index), n(tab->n), dc_cls(c), d_link(c), dc_last(c), d_dig(c),
          d_dig_sorted(c), k_a(c), k_b(c), d_cnt(c), d_rows_sorted(c), d_flags(c), d_nflag(c), r_a(c), r_b(c), l_a(c), l_b(c), d_inv(c),
          d_lab(c), d_tmp_dup(c), d_tmp_cl(c), d_lb(c), d_groups(c), d_grp_of(c), d_lead_rows(c), d_val(c), d_valj(c), d_cnt_sub(c),
          d_off_sub(c), d_nlead(c), d_key(c), d_keyj(c), temp(c), gs_of(c), key64_a(c), key64_b(c), d_stat(c), d_slots(c)
    {
    }

    int unusable(const char *why)
    {
        sp->usable = false;
        sp->why = why;
        stop = true;
        return MG_OK;
    }

    // the device's counts in the index's order: only if somebody asks (copy suspects, no clustering to be had)
    hipError_t counts_to_device()
    {
            if (!cnt_stale) return hipSuccess;
            cnt_stale = false;
            return hipMemcpyAsync(d_cnt, cnt_true.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
    }

    template <class T> bool take(T **p, size_t count)       // a retained buffer of the index, from the context's pool
    {
        void *q = nullptr;
        if (ctx_malloc(ctx, &q, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return false;
        *p = static_cast<T *>(q);
        return true;
    }

    void drop()                                             // the retained buffers back to the pool: this index is not to be used
    {
            for (void **q : {(void **)&sp->off, (void **)&sp->keys_sorted, (void **)&sp->gend, (void **)&sp->sorted_rows, (void **)&sp->code_img,
                             (void **)&sp->pos_img, (void **)&sp->short_rows, (void **)&sp->short_cnt, (void **)&sp->counters, (void **)&sp->order,
                             (void **)&sp->rep, (void **)&sp->cls_of, (void **)&sp->cls_off, (void **)&sp->cls_rows, (void **)&sp->cls_first})
                if (*q) { ctx_free(ctx, *q); *q = nullptr; }
    }

    int scan_rows();
    int order_rows();
    int find_copies();
    int lay_out();
    void prepare_candidates();
    void finish_build();
    int build_by_tiles();
    int build_by_sort();
    int check_build();
    int dense_groups();
    int run();
};

int SparseIndexBuild::scan_rows()
{
    // ---- ONE wait for everything the host must know before it can lay the index out (round 4 waited four times here):
    //  * the rows' hash counts and largest hashes, if the table is new to the library (table_classes' kernel);
    //  * whether any two rows may be copies of each other: every row's digest, sorted on the device; rows whose digest
    //    and length equal their predecessor's are suspects.  Taken in the TABLE's order -- whether copies exist is a
    //    property of the set of rows; only a table that has suspects digests its rows again in the index's order, has
    //    them verified value by value and keeps the copies out of the index (see compare_sparse.hip);
    //  * the clustered variant: rows that share one of their smallest hashes next to each other (labels on the device,
    //    two small sorts); the plain variant: which neighbouring rows are near-copies of each other (dense groups).
    need_classes = t->cls.size() != n;
    dedup = !ctx_opt(ctx, "MASHGPU_SPARSE_NO_DEDUP");
    want_dense = true;
    if (const char *v = ctx_opt(ctx, "MASHGPU_COMPARE_DENSE")) want_dense = atoi(v) != 0;
    try_cluster = clustered && n >= 16;
    try_links = !try_cluster && want_dense && n >= 8 && s <= 16384;      // (u16 counters of the extras, one bit a flag)
    tb_dup = mg::sparse_dup_temp_bytes((uint32_t)n);
    tb_cl = mg::dense_cluster_temp_bytes((uint32_t)n);
    if (d_cnt.alloc(n) != hipSuccess || (need_classes && (dc_cls.alloc(n) != hipSuccess || dc_last.alloc(n) != hipSuccess)) ||
        (dedup && (d_dig.alloc(n) != hipSuccess || d_dig_sorted.alloc(n) != hipSuccess || d_rows_sorted.alloc(n) != hipSuccess ||
                   d_flags.alloc(n) != hipSuccess || d_nflag.alloc(1) != hipSuccess || d_tmp_dup.alloc(std::max<size_t>(tb_dup, 16)) != hipSuccess))) {
        (void)hipGetLastError();
        return unusable("no device memory for the index");
    }
    {
        // Everything is LAUNCHED first and read back behind: a copy into pageable memory holds the host until the stream has
        // reached it.  (Measured and dropped: the copy suspects -- one large kernel, a small sort -- on a second stream beside
        // the clustering's two dozen small kernels end 0.18 ms earlier, and the six read-backs at 40 us each give it back.)
        hipError_t e = hipSuccess;
        if (need_classes) {
            hc_cls.resize(n);
            hc_last.resize(n);
            hc_nh.resize(n);
            e = mg::launch_row_classes(t->hashes, t->nhash, n, t->s, dc_cls, dc_last, ctx->stream);
        }
        // the rows' entry counts min(nhash, s) straight from the table's counts
        if (e == hipSuccess) e = mg::launch_sparse_row_counts(t->nhash, (uint32_t)n, (uint32_t)std::min<uint64_t>(t->s, s), d_cnt, ctx->stream);
        if (e == hipSuccess && dedup) {
            e = mg::launch_sparse_row_digest(t->hashes, t->s, d_cnt, (uint32_t)n, d_dig, ctx->stream);
            if (e == hipSuccess)
                e = mg::launch_sparse_dup_suspects(d_dig, d_cnt, (uint32_t)n, d_tmp_dup, tb_dup, d_dig_sorted, d_rows_sorted, d_flags, d_nflag, ctx->stream);
        }
        if (e == hipSuccess && try_cluster) {
            if (k_a.alloc(4 * n) == hipSuccess && k_b.alloc(4 * n) == hipSuccess && r_a.alloc(4 * n) == hipSuccess && r_b.alloc(4 * n) == hipSuccess &&
                l_a.alloc(n) == hipSuccess && l_b.alloc(n) == hipSuccess && d_inv.alloc(n) == hipSuccess && d_lab.alloc(n) == hipSuccess &&
                d_tmp_cl.alloc(std::max<size_t>(tb_cl, 16)) == hipSuccess) {
                e = mg::dense_cluster_rows(t->hashes, t->s, d_cnt, (uint32_t)n, d_tmp_cl, tb_cl, k_a, k_b, r_a, r_b, l_a, l_b, d_inv, d_lab, ctx->stream, split);
                cluster_q = true;
            } else {
                (void)hipGetLastError();                    // no memory for the clustering: the table's own order
            }
        }
        if (e == hipSuccess && try_links) {
            if (d_link.alloc(n) == hipSuccess) {
                link.resize(n);
                e = mg::launch_dense_neighbors(t->hashes, t->s, d_cnt, (uint32_t)n, d_link, ctx->stream);
                links_q = true;
            } else {
                (void)hipGetLastError();
            }
        }
        // the read-backs: into one pinned block if there is one -- queued all at once, one wait, then handed to the vectors
        // that outlive this call -- else straight into the vectors, one blocking copy after the other
        const size_t pin_bytes = 8 * n + 4 * n + 4 * n + 4 * n + 8 + n + n + 64;
        unsigned char *pin = static_cast<unsigned char *>(ctx_pinned(ctx, pin_bytes));
        unsigned long long *p_last = pin ? reinterpret_cast<unsigned long long *>(pin) : nullptr;
        uint32_t *p_nh = pin ? reinterpret_cast<uint32_t *>(pin + 8 * n) : nullptr, *p_inv = pin ? p_nh + n : nullptr, *p_lab = pin ? p_inv + n : nullptr;
        uint32_t *p_nflag = pin ? p_lab + n : &nflag;
        uint8_t *p_cls = pin ? pin + 20 * n + 8 : nullptr, *p_link = pin ? p_cls + n : nullptr;
        if (!pin) {
            if (cluster_q) { inv_v.resize(n); lab_v.resize(n); p_inv = inv_v.data(); p_lab = lab_v.data(); }
            p_nh = hc_nh.data(); p_cls = hc_cls.data(); p_last = reinterpret_cast<unsigned long long *>(hc_last.data()); p_link = link.data();
        }
        if (e == hipSuccess && need_classes) {
            e = hipMemcpyAsync(p_nh, t->nhash, n * 4, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(p_cls, dc_cls, n, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(p_last, dc_last, n * 8, hipMemcpyDeviceToHost, ctx->stream);
        }
        if (e == hipSuccess && dedup) e = hipMemcpyAsync(p_nflag, d_nflag, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && cluster_q) {
            e = hipMemcpyAsync(p_inv, d_inv, n * 4, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(p_lab, d_lab, n * 4, hipMemcpyDeviceToHost, ctx->stream);
        }
        if (e == hipSuccess && links_q) e = hipMemcpyAsync(p_link, d_link, n, hipMemcpyDeviceToHost, ctx->stream);
        const hipError_t es = hipStreamSynchronize(ctx->stream);      // (also when something failed: host memory is the target of copies)
        if (e == hipSuccess) e = es;
        if (e == hipSuccess && pin) {
            if (need_classes) {
                memcpy(hc_nh.data(), p_nh, n * 4);
                memcpy(hc_cls.data(), p_cls, n);
                memcpy(hc_last.data(), p_last, n * 8);
            }
            if (dedup) nflag = *p_nflag;
            if (links_q) memcpy(link.data(), p_link, n);
        }
        if (cluster_q) { inv = p_inv; lab_sorted = p_lab; }
        if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("compare (index: rows, copies, order): ") + hipGetErrorString(e));
    }
    if (need_classes) {
        t->cls.swap(hc_cls);
        t->last.swap(hc_last);
        t->nh.swap(hc_nh);
    }
    if (!links_q) link.clear();
    return MG_OK;
}

int SparseIndexBuild::order_rows()
{
    // How the index is built (MASHGPU_SPARSE_INDEX): "tiles" (default) -- index_build.hip; "sort" -- rounds 3-4, also what a
    // table takes that the tiles refuse; "verify" -- both, compared word by word on the device (tests).
    ix_mode = ctx_opt(ctx, "MASHGPU_SPARSE_INDEX");
    ix_verify = ix_mode && strcmp(ix_mode, "verify") == 0;
    ix_tiles = !ix_mode || strcmp(ix_mode, "sort") != 0;
    cnt_true.assign(n, 0);
    // (what does not depend on the rows' order is taken here, in the table's: the pass in the index's order below has the
    //  offsets left.  Copies of earlier rows, if there are any, count towards the densities: a bucket width is all they decide.)
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t c = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(t->nh[i], t->s), s);
        cnt_true[i] = c;
        E_all += c;
        max_cnt = std::max(max_cnt, c);
        if (c) {
            const uint64_t last = t->last[i];
            // a real hash equal to the padding value would sort among the padding: keep the tile engine
            if (last == MG_HASH_PAD) return unusable("a hash equals the padding value");
            maxv = std::max(maxv, last);
            const double d = (double)c / ((double)last + 1.0);
            dens[64 - __builtin_clzll(last | 1ull)] += d;
            dens0 += d;
        }
    }
    // ---- the clustered variant: the table copied in the clustered order; everything below then works on the copy as if
    // it were the table
    H = t->hashes;
    // A job over a RANGE of rows (split != 0) pays for the permuted copy and the clustering's order only where the collection
    // has large families: the split order keeps the pairs INSIDE each segment of a family dense, the pairs across the cut are
    // merged one by one either way.  With families of a hundred rows (C3) that is 1.9 against 2.6 ms of merges per pass for
    // 1.4 ms more per table (tools/range_check.py); with a clade of thousands it is the difference between a dense kernel and
    // a merge for every pair.  Below kSplitMinRun rows (at s = 1000) in the longest family of a segment the table keeps its own order.
    bool keep_table_order = false;
    // (a merge costs what the sketches are long: the rule is on rows x sketch size, 128 rows at s = 1000 -- C5's families of
    //  50 + 50 rows at s = 10 000 take the split order: 7 x against 10 x their share of a whole pass)
    if (cluster_q && split != 0 && lab_sorted && !ctx_opt(ctx, "MASHGPU_SPLIT_ALWAYS") && longest_label_run(lab_sorted, n) * (uint64_t)s < kSplitMinRun * 1000ull) {
        keep_table_order = true;
        lab_sorted = nullptr;
    }
    if (cluster_q) {
        bool identity = true;
        for (uint64_t a = 0; a < n && identity; a++) identity = inv[a] == a;
        if (!identity && !keep_table_order) {
            // (Measured and dropped: this copy queued BEFORE the read-backs above.  They are copies into pageable memory and
            //  wait for whatever the stream holds in front of them -- the copy kernel included; the build started 0.4 ms later.)
            void *pi = nullptr, *ph = nullptr;
            if (ctx_malloc(ctx, &pi, n * 4) != hipSuccess || ctx_malloc(ctx, &ph, std::max<uint64_t>(n * t->s, 1) * 8) != hipSuccess) {
                (void)hipGetLastError();
                ctx_free(ctx, pi);
                lab_sorted = nullptr;                       // no memory for the copy: the table's own order
            } else {
                sp->inv = static_cast<uint32_t *>(pi);
                sp->phashes = static_cast<uint64_t *>(ph);
                HIP_TRY(ctx, hipMemcpyAsync(sp->inv, d_inv, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
                // With no copy suspects every row enters the index, so its plan is known NOW (it depends on sums over the rows,
                // not on their order): the copy kernel then leaves the rows' window offsets (K0 of index_build.hip) on its way
                // -- one pass over the table less.
                if (ix_tiles && !(dedup && nflag) && E_all > 0 && E_all < (1ull << 31) && longest_label_run(lab_sorted, n) <= kTilesLongestRun) {
                    early_plan = mg::index_plan((uint32_t)n, (uint32_t)E_all, s, sp->rs, t->s, maxv, dens0, ix_verify);
                    if (early_plan.ok && d_lb.alloc(early_plan.lb_bytes) == hipSuccess) {
                        HIP_TRY(ctx, mg::index_gather_rows(early_plan, t->hashes, sp->inv, d_cnt, sp->phashes, d_lb, ctx->stream));
                        lb_made = true;
                    } else {
                        (void)hipGetLastError();
                    }
                }
                if (!lb_made) HIP_TRY(ctx, mg::launch_dense_gather_rows(t->hashes, t->s, sp->inv, (uint32_t)n, sp->phashes, ctx->stream));
                H = sp->phashes;
                cnt_perm.resize(n);
                for (uint64_t a = 0; a < n; a++) cnt_perm[a] = cnt_true[inv[a]];
                cnt_true.swap(cnt_perm);
                // (the device's counts follow only if somebody asks -- copy suspects, no clustering to be had: a copy from
                //  pageable memory waits for the gather in front of it, and the host has the index to plan meanwhile)
                cnt_stale = true;
            }
        }
    } else {
        lab_sorted = nullptr;
    }
    // which neighbouring rows are near-copies of each other (dense groups, compare_dense.hip)
    if (want_dense && n >= 8 && s <= 16384 && lab_sorted) {
        link.assign(n, 0);                                  // clustered variant: neighbours with the same label
        // (never across the two segments of a split order: index row `split` is the first row of the job's own segment)
        for (uint64_t a = 1; a < n; a++) link[a] = (a != split && lab_sorted[a] == lab_sorted[a - 1] && cnt_true[a] && cnt_true[a - 1]) ? 1 : 0;
    } else if (cluster_q && want_dense && n >= 8 && s <= 16384 && d_link.alloc(n) == hipSuccess) {
        // (the clustering was asked for and came to nothing: the neighbours of the table's own order after all)
        link.resize(n);
        HIP_TRY(ctx, counts_to_device());
        HIP_TRY(ctx, mg::launch_dense_neighbors(H, t->s, d_cnt, (uint32_t)n, d_link, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(link.data(), d_link, n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    } else if (!links_q) {
        (void)hipGetLastError();
        link.clear();
    }
    return MG_OK;
}

int SparseIndexBuild::find_copies()
{
    if (dedup && nflag) {
        // suspects: the digests again, of the rows the index will be built from and in its order
        HIP_TRY(ctx, counts_to_device());
        HIP_TRY(ctx, mg::launch_sparse_row_digest(H, t->s, d_cnt, (uint32_t)n, d_dig, ctx->stream));
        HIP_TRY(ctx, mg::launch_sparse_dup_suspects(d_dig, d_cnt, (uint32_t)n, d_tmp_dup, tb_dup, d_dig_sorted, d_rows_sorted, d_flags, d_nflag, ctx->stream));
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
        if (sp->one_class && ctx->aside_all_copies) ctx->aside_all_copies(sp->one_class);     // (a fill beside this build has the wrong constant)
    }
    return MG_OK;
}

int SparseIndexBuild::lay_out()
{
    uint64_t E64 = 0;
    sp->off_host.resize(n + 1);
    for (uint64_t i = 0; i < n; i++) {
        sp->off_host[i] = (uint32_t)E64;
        const uint64_t c = cnt_true[i];
        if (rep.empty() || rep[i] == i) E64 += c;           // copies stay out of the index
        if (E64 >= (1ull << 31)) return unusable("2^31 entries or more");
        if (c < s) sp->short_rows_host.push_back((uint32_t)i);
        if (c == 0) sp->has_empty = true;
    }
    sp->off_host[n] = (uint32_t)E64;
    if (E64 == 0) return unusable("no hashes");
    E = (uint32_t)E64;
    sp->E = E;
    end_bit = (uint32_t)(64 - __builtin_clzll(maxv | 1ull));
    // The fallback sort looks at the values' leading bits only (compare_sparse.hip: sparse_sort_begin_bit sizes that for values spread
    // evenly below the largest one).  A collection of genomes of many sizes is not spread evenly: a row's s smallest hashes
    // fill [0, its largest hash], so the low end of the range holds the values of every row and the high end those of the
    // small genomes only.  Where the values are dense they need more bits to be told apart: the expected number of pairs
    // of different values in one bucket of 2^b is 2^b / 2 x the integral of the squared density, taken from the rows'
    // largest hashes (by bit length); b is chosen for 2^13 of them at most, and never above the even-spread rule (sort_bits.h).
    sort_begin_bit = mg::sparse_sort_begin_bit(E, end_bit, ctx_opt(ctx, "MASHGPU_SPARSE_SORT_BITS"), ctx_opt(ctx, "MASHGPU_SPARSE_SORT_ALL_BITS") != nullptr);
    if (sort_begin_bit > 0 && !ctx_opt(ctx, "MASHGPU_SPARSE_SORT_BITS")) sort_begin_bit = mg::sort_begin_bit_from_density(dens, end_bit, sort_begin_bit);
    // room for the dense groups' leaders (prepare_candidates): one list entry each, a quarter of the entries in all
    lead_lists = mg::dense_sublists();
    lead_cap = std::max<uint32_t>(E / 4u / lead_lists + 64u, 256u);
    if (const char *ev = ctx_opt(ctx, "MASHGPU_DENSE_LEAD_CAP")) lead_cap = std::max<uint32_t>((uint32_t)atoi(ev), 1u);      // (test knob: lists that overflow)
    const bool tiles_hopeless = ix_tiles && !ix_verify && lab_sorted && !ctx_opt(ctx, "MASHGPU_SPARSE_INDEX") &&
                                longest_label_run(lab_sorted, n) > kTilesLongestRun;
    if (ix_tiles && !tiles_hopeless) plan = mg::index_plan((uint32_t)n, E, s, sp->rs, t->s, maxv, dens0, ix_verify);
    else if (tiles_hopeless) plan.why = "a clade of more rows than a bucket's sort takes";
    // the tiles' plan (index_build.hip; MASHGPU_SPARSE_INDEX=sort: none).  The window offsets the clustered copy left are
    // those of this plan, or K0 makes them again.
    if (lb_made && !(plan.ok && memcmp(&plan.g, &early_plan.g, sizeof plan.g) == 0)) lb_made = false;
    // transient buffers (members: back to the pool when the build object goes, in stream order) and the retained ones
    temp_bytes = std::max(std::max(mg::sparse_order_temp_bytes((uint32_t)n), mg::sparse_order_slice_temp_bytes((uint32_t)n)),
                          mg::sparse_offsets_temp_bytes((uint32_t)n));
    want_order = !ctx_opt(ctx, "MASHGPU_SPARSE_NO_ORDER");
    ok = temp.alloc(std::max<size_t>(temp_bytes, 16)) == hipSuccess && gs_of.alloc(E) == hipSuccess && d_stat.alloc(1) == hipSuccess &&
              d_slots.alloc(std::max(mg::sparse_stat_scratch_bytes(), mg::index_stat_scratch_bytes())) == hipSuccess &&
              (!want_order || (key64_a.alloc(n) == hipSuccess && key64_b.alloc(n) == hipSuccess));
    // retained buffers
    ok = ok && take(&sp->off, n + 1) && take(&sp->keys_sorted, E) && take(&sp->gend, E) && take(&sp->sorted_rows, E) &&
         take(&sp->code_img, (size_t)n * sp->rs + 64) && take(&sp->pos_img, (size_t)n * sp->rs) && take(&sp->counters, 4) &&
         (!want_order || take(&sp->order, n));
    nshort = sp->short_rows_host.size();
    short_cnt.assign(nshort, 0);
    for (size_t k = 0; k < nshort; k++) short_cnt[k] = cnt_true[sp->short_rows_host[k]];
    if (ok && nshort) ok = take(&sp->short_rows, nshort) && take(&sp->short_cnt, nshort);
    if (ok && sp->copies)
        ok = take(&sp->rep, n) && take(&sp->cls_of, n) && take(&sp->cls_off, cls_off.size()) && take(&sp->cls_rows, cls_rows.size()) &&
             take(&sp->cls_first, cls_first.size());
    e = hipSuccess;
    if (ok && sp->copies) {
        e = hipMemcpyAsync(sp->rep, rep.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(sp->cls_of, cls_of.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(sp->cls_off, cls_off.data(), cls_off.size() * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(sp->cls_rows, cls_rows.data(), cls_rows.size() * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(sp->cls_first, cls_first.data(), cls_first.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    }
    if (ok && e == hipSuccess) {
        // the rows' offsets: made on the device from its counts (sp_offsets_kernel) unless copies stay out of the index
        // (its temporary: the ordering's, not in use before the index stands)
        if (rep.empty()) e = mg::launch_sparse_offsets(d_cnt, cnt_stale ? sp->inv : nullptr, (uint32_t)n, temp, sp->off, ctx->stream);
        else e = hipMemcpyAsync(sp->off, sp->off_host.data(), (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && nshort) e = hipMemcpyAsync(sp->short_rows, sp->short_rows_host.data(), nshort * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && nshort) e = hipMemcpyAsync(sp->short_cnt, short_cnt.data(), nshort * 4, hipMemcpyHostToDevice, ctx->stream);
    }
    return MG_OK;
}

// ---- candidates for dense groups (compare_dense.hip): runs of at least 8 consecutive rows linked to their predecessors.
// Known before the index exists, so the build by tiles looks for their leaders while it has every group of equal values in
// LDS (index_build.h, IxLeaders); the build by the sort searches the finished index for them (dense_find_leaders).
void SparseIndexBuild::prepare_candidates()
{
    if (cand_prepared) return;
    cand_prepared = true;
    if (link.empty() || sp->copies != 0) return;
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
    if (cand_groups.empty()) return;
    const uint32_t ng = (uint32_t)cand_groups.size();
    grp_of_h.assign(n, 0xFFFFFFFFu);
    for (uint32_t g = 0; g < ng; g++)
        for (uint32_t r = cand_groups[g].g0; r < cand_groups[g].g1; r++) grp_of_h[r] = g;
    // the device's copies -- every row's group, and per row {group, its first row, one past its last, 0}: what K4
    // gathers -- are made there from the groups (dn_group_rows_kernel)
    const uint64_t room = (uint64_t)lead_lists * lead_cap;
    if (d_groups.alloc(ng) == hipSuccess && d_grp_of.alloc(n) == hipSuccess && d_lead_rows.alloc(4 * n) == hipSuccess && d_nlead.alloc(2) == hipSuccess && d_cnt_sub.alloc(lead_lists) == hipSuccess &&
        d_off_sub.alloc(lead_lists) == hipSuccess && d_key.alloc(room) == hipSuccess && d_val.alloc(room) == hipSuccess && d_keyj.alloc(room) == hipSuccess &&
        d_valj.alloc(room) == hipSuccess &&
        hipMemcpyAsync(d_groups, cand_groups.data(), ng * sizeof(mg::DenseGroup), hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
        mg::launch_dense_group_rows(d_groups, ng, (uint32_t)n, d_grp_of, d_lead_rows, ctx->stream) == hipSuccess &&
        hipMemsetAsync(d_cnt_sub, 0, lead_lists * 4, ctx->stream) == hipSuccess && hipMemsetAsync(d_nlead, 0, 8, ctx->stream) == hipSuccess) {
        lead_ready = true;
    } else {
        (void)hipGetLastError();
        cand_groups.clear();                            // no room: no dense groups
    }
}

// visiting order of the rows (by the run of their first shared value, larger rows first inside a run), the statistics, one wait
void SparseIndexBuild::finish_build()
{
    if (e == hipSuccess && want_order)
        e = mg::launch_sparse_row_order(sp->off, sp->code_img, sp->gend, sp->rep, (uint32_t)n, sp->rs, temp, temp_bytes, key64_a, key64_b,
                                        sp->order, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_stat, d_stat, sizeof(Stat), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
}

int SparseIndexBuild::build_by_tiles()
{
    if (ok && e == hipSuccess && plan.ok) {
        DevBuf<unsigned char> d_tcnt(ctx), d_start(ctx), d_big(ctx), d_pk(ctx), d_tc(ctx);
        if ((lb_made || d_lb.alloc(plan.lb_bytes) == hipSuccess) && d_tcnt.alloc(plan.cnt_bytes) == hipSuccess && d_start.alloc(plan.start_bytes) == hipSuccess &&
            d_big.alloc(plan.big_bytes) == hipSuccess && d_pk.alloc(plan.pk_bytes) == hipSuccess && d_tc.alloc(plan.tc_bytes) == hipSuccess) {
            e = hipMemsetAsync(d_stat, 0, sizeof(Stat), ctx->stream);
            // the partition (K0 - K3) goes first: the host lays out the candidates for dense groups while it runs
            if (e == hipSuccess)
                e = mg::index_build(plan, H, sp->off, d_lb, d_tcnt, d_start, d_big, d_pk, d_tc, sp->keys_sorted, sp->sorted_rows, sp->gend, gs_of, sp->code_img,
                                    sp->pos_img, d_slots, &d_stat.p->shared, &d_stat.p->max_group, &d_stat.p->groups, d_stat.p->ixf, nullptr, ctx->stream,
                                    lb_made ? 1 | 8 : 1);
            // (test knob: what a ticket served out of lane order would leave behind -- the sorts' order check has to refuse the table)
            DevBuf<uint32_t> d_swapped(ctx);
            if (e == hipSuccess && ctx_opt(ctx, "MASHGPU_IX_DEBUG_SWAP") && d_swapped.alloc(1) == hipSuccess) {
                e = hipMemsetAsync(d_swapped, 0, 4, ctx->stream);
                if (e == hipSuccess) e = mg::index_debug_swap(plan, d_pk, d_start, d_swapped, ctx->stream);
            }
            prepare_candidates();
            mg::IxLeaders lead;
            if (lead_ready) {
                lead.grp_of = d_lead_rows;
                lead.key = d_key;
                lead.val = d_val;
                lead.cnt = d_cnt_sub;
                lead.cap_sub = lead_cap;
                lead.nsub = lead_lists;
            }
            // values, rows, groups, statistics, leaders: final behind the bucket sorts.  What the host wants of them is copied
            // back and marked with an event; the images (K5) and the rows' visiting order are queued behind, and the host waits
            // for the EVENT -- it lays out the dense groups while the images are still being written.
            if (e == hipSuccess && ctx->aside_at_sort) ctx->aside_at_sort();      // (a job's fill that waits for the sorts: host_compare.cpp, prefill)
            if (e == hipSuccess)
                e = mg::index_build(plan, H, sp->off, d_lb, d_tcnt, d_start, d_big, d_pk, d_tc, sp->keys_sorted, sp->sorted_rows, sp->gend, gs_of, sp->code_img,
                                    sp->pos_img, d_slots, &d_stat.p->shared, &d_stat.p->max_group, &d_stat.p->groups, d_stat.p->ixf,
                                    lead_ready ? &lead : nullptr, ctx->stream, 2);
            if (e == hipSuccess && lead_ready) {             // the leaders' lists made one; their count comes back with the statistics
                e = mg::dense_join_leaders(d_key, d_val, lead_cap, d_keyj, d_valj, d_cnt_sub, d_off_sub, d_nlead, ctx->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(lead_tot, d_nlead, 8, hipMemcpyDeviceToHost, ctx->stream);
            }
            hipEvent_t ev_stats = nullptr;
            if (e == hipSuccess) e = hipMemcpyAsync(&h_stat, d_stat, sizeof(Stat), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_stats, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(ev_stats, ctx->stream);
            if (e == hipSuccess)
                e = mg::index_build(plan, H, sp->off, d_lb, d_tcnt, d_start, d_big, d_pk, d_tc, sp->keys_sorted, sp->sorted_rows, sp->gend, gs_of, sp->code_img,
                                    sp->pos_img, d_slots, &d_stat.p->shared, &d_stat.p->max_group, &d_stat.p->groups, d_stat.p->ixf,
                                    lead_ready ? &lead : nullptr, ctx->stream, 4);
            if (e == hipSuccess && want_order)
                e = mg::launch_sparse_row_order(sp->off, sp->code_img, sp->gend, sp->rep, (uint32_t)n, sp->rs, temp, temp_bytes, key64_a, key64_b,
                                                sp->order, ctx->stream);
            if (ev_stats) {
                const hipError_t ew = e == hipSuccess ? hipEventSynchronize(ev_stats) : hipStreamSynchronize(ctx->stream);
                if (e == hipSuccess) e = ew;
                (void)hipEventDestroy(ev_stats);
            } else {
                (void)hipStreamSynchronize(ctx->stream);    // (host memory is the target of copies that may be queued)
            }
            built = e == hipSuccess && !h_stat.ixf[mg::IXF_DEGENERATE];
            // (not with buckets beyond the LDS: their second split reuses the partition's array, which the sorts would read again)
            if (built && lead_ready && lead_tot[1] > lead_cap && h_stat.ixf[mg::IXF_NBIG] == 0) {
                // A list of leaders filled beyond its room (the lists are filled by bucket: a table whose near-copies crowd a few
                // buckets).  The leaders are found a second time with the room the fullest list asked for: the bucket sorts
                // again -- the same index into the same arrays, the statistics counted anew (the table used to
                // lose its dense groups here).
                e = hipStreamSynchronize(ctx->stream);
                const uint32_t want_cap = lead_tot[1];
                for (void **q : {(void **)&d_key.p, (void **)&d_val.p, (void **)&d_keyj.p, (void **)&d_valj.p})
                    if (*q) { ctx_free(ctx, *q); *q = nullptr; }
                const uint64_t room = (uint64_t)lead_lists * want_cap;
                if (e == hipSuccess && d_key.alloc(room) == hipSuccess && d_val.alloc(room) == hipSuccess && d_keyj.alloc(room) == hipSuccess &&
                    d_valj.alloc(room) == hipSuccess) {
                    lead_cap = want_cap;
                    lead.key = d_key;
                    lead.val = d_val;
                    lead.cap_sub = lead_cap;
                    e = hipMemsetAsync(d_cnt_sub, 0, lead_lists * 4, ctx->stream);
                    if (e == hipSuccess) e = hipMemsetAsync(d_nlead, 0, 8, ctx->stream);
                    if (e == hipSuccess) e = hipMemsetAsync(d_stat.p, 0, offsetof(Stat, bad), ctx->stream);      // shared, max_group, groups
                    if (e == hipSuccess)
                        e = mg::index_build(plan, H, sp->off, d_lb, d_tcnt, d_start, d_big, d_pk, d_tc, sp->keys_sorted, sp->sorted_rows, sp->gend, gs_of,
                                            sp->code_img, sp->pos_img, d_slots, &d_stat.p->shared, &d_stat.p->max_group, &d_stat.p->groups, d_stat.p->ixf, &lead,
                                            ctx->stream, 2);
                    if (e == hipSuccess) e = mg::dense_join_leaders(d_key, d_val, lead_cap, d_keyj, d_valj, d_cnt_sub, d_off_sub, d_nlead, ctx->stream);
                    if (e == hipSuccess) e = hipMemcpyAsync(lead_tot, d_nlead, 8, hipMemcpyDeviceToHost, ctx->stream);
                    if (e == hipSuccess) e = hipMemcpyAsync(&h_stat, d_stat, sizeof(Stat), hipMemcpyDeviceToHost, ctx->stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                    built = e == hipSuccess && !h_stat.ixf[mg::IXF_DEGENERATE];
                    if (ctx_opt(ctx, "MASHGPU_SPARSE_DBG"))
                        fprintf(stderr, "compare dense: a list of leaders asked for %u entries: found again with that room (%u leaders)\n", want_cap, lead_tot[0]);
                } else {
                    (void)hipGetLastError();                // (no room: the table goes without dense groups, as before)
                }
            }
            lead_done = built && lead_ready;
            if (ctx_opt(ctx, "MASHGPU_SPARSE_DBG"))
                fprintf(stderr, "compare sparse: index by tiles: shift %u, %u buckets (%u per window, %u windows), fullest %u, %u beyond the LDS (%u values streamed)%s\n",
                        plan.g.shift, plan.g.Bp, plan.g.BW, plan.g.NW, h_stat.ixf[mg::IXF_MAXBUCKET], h_stat.ixf[mg::IXF_NBIG], h_stat.ixf[mg::IXF_NSTREAMED],
                        h_stat.ixf[mg::IXF_DEGENERATE] ? " -- clumped values: sorted instead" : "");
        } else {
            (void)hipGetLastError();
        }
    } else if (ix_tiles && ctx_opt(ctx, "MASHGPU_SPARSE_DBG")) {
        fprintf(stderr, "compare sparse: index by tiles refused: %s\n", plan.why);
    }
    sp->by_tiles = built;
    prepare_candidates();                                   // (no tiles: the candidates are still to be laid out)
#ifdef IX_PHASE_CLOCKS
    if (built && ctx_opt(ctx, "MASHGPU_IX_CLOCKS")) mg::index_dump_clocks();
#endif
    if (ix_verify && !built && ok && e == hipSuccess && !ctx_opt(ctx, "MASHGPU_SPARSE_INDEX_MAY_REFUSE"))
        return fail(ctx, MG_ERR_INVALID, std::string("index verify: the tile build refused this table (") + (plan.ok ? "bucket flags" : plan.why) + ")");
    return MG_OK;
}

int SparseIndexBuild::build_by_sort()
{
    if (ok && e == hipSuccess && (!built || ix_verify)) {
        // ---- the general way: every entry through a radix sort on the values' leading bits (compare_sparse.hip)
        const size_t sort_bytes = mg::sparse_sort_temp_bytes(E, end_bit, sort_begin_bit);
        DevBuf<unsigned char> temp_sort(ctx), d_ties(ctx);
        DevBuf<uint64_t> keys_a(ctx), v_keys(ctx);
        DevBuf<uint32_t> idx_a(ctx), idx_sorted(ctx), v_rows(ctx), v_gend(ctx), v_gs(ctx), v_code(ctx), v_pos(ctx);
        DevBuf<unsigned long long> d_diff(ctx);
        bool ok2 = temp_sort.alloc(std::max<size_t>(sort_bytes, 16)) == hipSuccess && keys_a.alloc(E) == hipSuccess && idx_a.alloc(E) == hipSuccess &&
                   idx_sorted.alloc(E) == hipSuccess && d_ties.alloc(mg::sparse_tie_scratch_bytes()) == hipSuccess;
        if (ok2 && built)                                   // (verify: the second build goes into arrays of its own)
            ok2 = v_keys.alloc(E) == hipSuccess && v_rows.alloc(E) == hipSuccess && v_gend.alloc(E) == hipSuccess && v_gs.alloc(E) == hipSuccess &&
                  v_code.alloc((size_t)n * sp->rs + 64) == hipSuccess && v_pos.alloc((size_t)n * sp->rs) == hipSuccess && d_diff.alloc(16) == hipSuccess;
        if (!ok2) {
            (void)hipGetLastError();
            ok = false;
        } else {
            const Stat tiles_stat = h_stat;
            uint64_t *o_keys = built ? v_keys.p : sp->keys_sorted;
            uint32_t *o_rows = built ? v_rows.p : sp->sorted_rows, *o_gend = built ? v_gend.p : sp->gend, *o_gs = built ? v_gs.p : gs_of.p,
                     *o_code = built ? v_code.p : sp->code_img, *o_pos = built ? v_pos.p : sp->pos_img;
            // (the sort looks at the values' leading bits only and repairs the few ties; a table that defeats that is sorted again, on every bit)
            for (uint32_t begin_bit = sort_begin_bit;; begin_bit = 0) {
                if (e == hipSuccess) e = hipMemsetAsync(d_stat, 0, sizeof(Stat), ctx->stream);
                if (e == hipSuccess)
                    e = mg::sparse_build_index(H, t->s, sp->off, (uint32_t)n, E, sp->rs, end_bit, temp_sort, sort_bytes, keys_a, idx_a, o_keys, idx_sorted,
                                               /*head=*/idx_a, o_gs, o_rows, o_gend, o_code, o_pos, d_slots, begin_bit, d_ties, &d_stat.p->shared,
                                               &d_stat.p->max_group, &d_stat.p->groups, &d_stat.p->bad, &d_stat.p->tie_overflow, ctx->stream);
                if (built) {
                    if (e == hipSuccess) e = hipMemcpyAsync(&h_stat, d_stat, sizeof(Stat), hipMemcpyDeviceToHost, ctx->stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                } else {
                    finish_build();
                }
                // (also when the order came out wrong behind a sort on the leading bits -- never seen since the repair kernels
                //  exist, but the remedy is the same and silent downgrades to the tile engine are worse)
                if (e != hipSuccess || (!h_stat.tie_overflow && !h_stat.bad) || begin_bit == 0) break;
                if (ctx_opt(ctx, "MASHGPU_SPARSE_DBG"))
                    fprintf(stderr, "compare sparse: sort on bits [%u, %u) %s: every bit again\n", begin_bit, end_bit,
                            h_stat.bad ? "left values out of order" : "left ties the repair could not take");
            }
            if (built && e == hipSuccess) {
                // every array of the two builds, word by word
                unsigned long long diff[16];
                for (int k = 0; k < 8; k++) { diff[2 * k] = 0; diff[2 * k + 1] = ~0ull; }
                e = hipMemcpyAsync(d_diff, diff, sizeof diff, hipMemcpyHostToDevice, ctx->stream);
                const uint64_t img = (uint64_t)n * sp->rs;
                if (e == hipSuccess) e = mg::index_verify_words((const uint32_t *)sp->keys_sorted, (const uint32_t *)v_keys.p, nullptr, 0, 2ull * E, d_diff.p + 0, ctx->stream);
                if (e == hipSuccess) e = mg::index_verify_words(sp->sorted_rows, v_rows, nullptr, 0, E, d_diff.p + 2, ctx->stream);
                if (e == hipSuccess) e = mg::index_verify_words(gs_of, v_gs, nullptr, 0, E, d_diff.p + 4, ctx->stream);
                if (e == hipSuccess) e = mg::index_verify_words(sp->gend, v_gend, v_gs, 1, E, d_diff.p + 6, ctx->stream);
                if (e == hipSuccess) e = mg::index_verify_words(sp->code_img, v_code, nullptr, 0, img, d_diff.p + 8, ctx->stream);
                if (e == hipSuccess) e = mg::index_verify_words(sp->pos_img, v_pos, v_code, 2, img, d_diff.p + 10, ctx->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(diff, d_diff, sizeof diff, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e == hipSuccess) {
                    static const char *names[6] = {"values (32-bit words)", "rows", "group starts", "group ends", "code image", "position image"};
                    std::string msg;
                    for (int k = 0; k < 6; k++)
                        if (diff[2 * k]) msg += std::string(msg.empty() ? "" : "; ") + names[k] + ": " + std::to_string(diff[2 * k]) + " differ, first at " + std::to_string(diff[2 * k + 1]);
                    if (h_stat.shared != tiles_stat.shared || h_stat.max_group != tiles_stat.max_group || h_stat.groups != tiles_stat.groups)
                        msg += std::string(msg.empty() ? "" : "; ") + "statistics: shared " + std::to_string(tiles_stat.shared) + " / " + std::to_string(h_stat.shared) +
                               ", largest group " + std::to_string(tiles_stat.max_group) + " / " + std::to_string(h_stat.max_group) + ", groups " +
                               std::to_string(tiles_stat.groups) + " / " + std::to_string(h_stat.groups);
                    if (!msg.empty()) {
                        char geom[160];
                        snprintf(geom, sizeof geom, " [n %llu, E %u, shift %u, %u buckets, %u per window, %u windows]", (unsigned long long)n, E, plan.g.shift, plan.g.Bp, plan.g.BW, plan.g.NW);
                        sp->usable = false;
                        return fail(ctx, MG_ERR_INVALID, "index verify: tiles / sort -- " + msg + geom);
                    }
                }
            }
        }
    }
    return MG_OK;
}

int SparseIndexBuild::check_build()
{
    if (!ok) { (void)hipGetLastError(); drop(); return unusable("no device memory for the index"); }
    if (e != hipSuccess) { drop(); return fail(ctx, MG_ERR_HIP, std::string("compare (index build): ") + hipGetErrorString(e)); }
    if (h_stat.bad) { drop(); return unusable("sort order inside a value not by row"); }
    sp->G = h_stat.groups;
    sp->shared = h_stat.shared;
    sp->max_group = h_stat.max_group;
    sp->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    sp->usable = true;
    return MG_OK;
}

int SparseIndexBuild::dense_groups()
{
    // ---- dense groups: runs of at least 8 consecutive rows linked to their predecessors.  Their universes come from the
    // index just built (gs_of is still alive), groups without one (or with one too large for a tile's LDS) are dropped,
    // the rest are encoded and the index's runs clipped for their rows.  Any failure here leaves the index as it is.
    if (!cand_groups.empty()) {
        auto t_dense = std::chrono::steady_clock::now();
        while (!cand_groups.empty()) {                       // (a block to leave with `break`)
            const uint32_t ng = (uint32_t)cand_groups.size();
            std::vector<uint32_t> &grp_of = grp_of_h;
            DevBuf<uint32_t> d_us(ctx), d_ue(ctx);
            DevBuf<unsigned long long> d_key2(ctx);
            DevBuf<unsigned char> d_tmp(ctx);
            if (d_us.alloc(ng) != hipSuccess || d_ue.alloc(ng) != hipSuccess) { (void)hipGetLastError(); break; }
            // leaders: one list entry each, appended to one of a thousand lists (room for a quarter of the entries in all, evenly).
            // The build by tiles has found them already; behind the sort the finished index is searched, a second time with
            // the room the first pass asked for if the lists filled unevenly.
            const uint32_t L = lead_lists;
            uint32_t *tot = lead_tot, cap_sub = lead_cap;
            hipError_t e2 = hipSuccess;
            if (!lead_done) {
                if (built || !lead_ready) break;             // (searching needs every position's group start: the sort wrote it)
                for (int attempt = 0; e2 == hipSuccess; attempt++) {
                    if (attempt > 0) {
                        for (void **q : {(void **)&d_key.p, (void **)&d_val.p, (void **)&d_keyj.p, (void **)&d_valj.p})
                            if (*q) { ctx_free(ctx, *q); *q = nullptr; }
                        const uint64_t room = (uint64_t)L * cap_sub;
                        if (d_key.alloc(room) != hipSuccess || d_val.alloc(room) != hipSuccess || d_keyj.alloc(room) != hipSuccess || d_valj.alloc(room) != hipSuccess) {
                            (void)hipGetLastError();
                            e2 = hipErrorOutOfMemory;
                            break;
                        }
                    }
                    e2 = mg::dense_find_leaders(sp->sorted_rows, gs_of, sp->gend, d_grp_of, d_groups, E, d_key, d_val, cap_sub, d_keyj, d_valj, d_cnt_sub, d_off_sub,
                                                d_nlead, ctx->stream);
                    if (e2 == hipSuccess) e2 = hipMemcpyAsync(tot, d_nlead, 8, hipMemcpyDeviceToHost, ctx->stream);
                    if (e2 == hipSuccess) e2 = hipStreamSynchronize(ctx->stream);
                    if (e2 != hipSuccess || tot[1] <= cap_sub || attempt >= 1) break;
                    cap_sub = tot[1];
                }
            } else if (tot[1] > cap_sub && ctx_opt(ctx, "MASHGPU_SPARSE_DBG")) {
                fprintf(stderr, "compare dense: a list of leaders asked for %u entries, room for %u: no dense groups for this table\n", tot[1], cap_sub);
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
            // (Measured and dropped: this sort -- two dozen small kernels that need the leaders only -- on a second stream beside
            //  the images' kernel.  The large kernel holds every CU: a 4 us memset took 170 us there, and the sort ended when it
            //  would have ended behind the images.)
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
                words += ((m + 127) / 128) * mg::dense_block_words(G.W);
                wmax = std::max(wmax, G.W);
                for (uint32_t r = G.g0; r < G.g1; r++) grp_of[r] = (uint32_t)sp->dgroups_host.size();
                sp->dgroups_host.push_back(G);
            }
            if (sp->dgroups_host.empty()) break;
            sp->dn_wmax = wmax;
            sp->dn_xs = ((s + 7u) & ~7u) + 8u;
            void *p1 = nullptr, *p2 = nullptr, *p3 = nullptr, *p4 = nullptr;
            if (ctx_malloc(ctx, &p1, sp->dgroups_host.size() * sizeof(mg::DenseGroup)) != hipSuccess || ctx_malloc(ctx, &p2, n * 4) != hipSuccess ||
                ctx_malloc(ctx, &p3, words * 8) != hipSuccess || ctx_malloc(ctx, &p4, (size_t)xrows * sp->dn_xs * 2) != hipSuccess) {
                (void)hipGetLastError();
                for (void *q : {p1, p2, p3, p4}) ctx_free(ctx, q);
                sp->dgroups_host.clear();
                break;
            }
            sp->dgroups = static_cast<mg::DenseGroup *>(p1);
            sp->grp_of = static_cast<uint32_t *>(p2);
            sp->gdata = static_cast<unsigned long long *>(p3);
            sp->ext = static_cast<uint16_t *>(p4);
            e2 = hipMemcpyAsync(sp->dgroups, sp->dgroups_host.data(), sp->dgroups_host.size() * sizeof(mg::DenseGroup), hipMemcpyHostToDevice, ctx->stream);
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(sp->grp_of, grp_of.data(), n * 4, hipMemcpyHostToDevice, ctx->stream);
            // masks, extras, and the index's runs clipped for the rows of the groups: discovery sees the partners outside only
            if (e2 == hipSuccess)
                e2 = mg::launch_dense_encode(sp->off, sp->code_img, sp->pos_img, sp->rs, sp->grp_of, sp->dgroups, sp->ulist, sp->upos, sp->gdata,
                                             sp->ext, sp->dn_xs, (uint32_t)n, wmax, ctx->stream,
                                             ctx_opt(ctx, "MASHGPU_DENSE_UL_LDS") ? atoi(ctx_opt(ctx, "MASHGPU_DENSE_UL_LDS")) : -1);
            // (no wait here: what the copies above read -- dgroups_host and the rows' group map -- lives in the index, what
            //  comes next is queued on the same stream, and a fault shows at its wait)
            sp->grp_of_host.swap(grp_of);
            if (e2 != hipSuccess) {
                // the runs may be half clipped: this index is not to be used
                drop();
                for (void **q : {(void **)&sp->dgroups, (void **)&sp->grp_of, (void **)&sp->gdata, (void **)&sp->ext, (void **)&sp->ulist,
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

int SparseIndexBuild::run()
{
    int rc = MG_OK;
    if ((rc = scan_rows()) != MG_OK || stop) return rc;
    if ((rc = order_rows()) != MG_OK || stop) return rc;
    if ((rc = find_copies()) != MG_OK || stop) return rc;
    if ((rc = lay_out()) != MG_OK || stop) return rc;
    if ((rc = build_by_tiles()) != MG_OK || stop) return rc;
    if ((rc = build_by_sort()) != MG_OK || stop) return rc;
    if ((rc = check_build()) != MG_OK || stop) return rc;
    return dense_groups();
}

}  // namespace

int table_sparse_index(mg_ctx *ctx, const mg_table *t, uint32_t s, bool clustered, mg_table::Sparse **out, uint32_t split)
{
    if (!clustered) split = 0;
    for (mg_table::Sparse *sp : t->sparse)
        if (sp->s == s && sp->clustered == clustered && sp->split == split) { *out = sp; return MG_OK; }
    mg_table::Sparse *sp = new mg_table::Sparse;
    sp->s = s;
    sp->clustered = clustered;
    sp->split = split;
    t->sparse.push_back(sp);
    *out = sp;
    auto unusable = [&](const char *why) { sp->usable = false; sp->why = why; return MG_OK; };
    const uint64_t n = t->n;
    if (n == 0) return unusable("empty table");
    if (n >= (1ull << 31)) return unusable("too many rows");
    sp->rs = mg::sparse_img_stride(s);
    if (n * sp->rs >= (1ull << 32)) return unusable("image index beyond 32 bits");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    SparseIndexBuild build(ctx, t, s, clustered, sp, split);
    // (the "index" phase of the library's HIP-event records: everything the build queues -- clustering, digests, the
    //  index, dense groups -- incl. the waits between its steps)
    prof_begin(ctx, ctx->prof_index);
    struct ProfEnd { mg_ctx *c; ~ProfEnd() { prof_end(c, c->prof_index); } } prof_end_guard{ctx};
    return build.run();
}
