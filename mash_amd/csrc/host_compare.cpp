// host_compare.cpp -- the compare entry points: tile engine, inverted-index engine, finishing, thresholded and list outputs
#include "host_internal.h"
#include "index_build.h"
#ifdef IX_PHASE_CLOCKS
namespace mg { void index_dump_clocks(); }
#endif
#include "sort_bits.h"

/* ------------------------------------------------------------------ comparing */

// largest hash of a table (device reduction, cached)
int table_max(mg_ctx *ctx, const mg_table *t, uint64_t *out)
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

static const mg_table *tri_view(mg_ctx *ctx, const mg_table *t, uint64_t rb, uint64_t re);

// ---- inverted-index engine (compare_sparse.hip) ----------------------------------------------
//
// The index of a table is built by host_index.cpp (table_sparse_index: one object, one function per phase).

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
struct SparseJob {
    mg::SparseArgs args;
    uint64_t cand = 0;                                      // candidates: pairs that share a hash and lie in no dense group
    mg_table::Sparse *ix = nullptr;
    // the pairs inside the dense groups of the job's rows (a list job gets them appended to their rows' lists: job_lists)
    const mg::DenseTile *dtiles = nullptr;
    uint32_t ndtiles = 0, dtile_rows = 0;
    uint64_t dense_pairs = 0;
};

// What the dispatch believes things cost: SparseCosts (host_internal.h) -- a context starts from numbers measured on one MI355X
// and corrects them from its OWN launches: the phases of every job seen for the first time are timed (events on the context's
// stream, read when the job has ended) and the per-unit prices move half way towards what was measured, within a factor of
// four of the defaults (SparseJobRun::learn).  MASHGPU_COSTS_FIXED: the defaults, always.


// One job of the inverted-index engine, phase by phase (run()): the index of the column table, the row side (triangle: the
// table itself; rect: the queries located in the index), the job's plan (dense tiles, visiting order; cached per table and
// rows), discovery -- which also counts, so the engine choice needs no pass of its own --, then fill + dense groups, merge
// and scatter.  A phase that finds the job is not this engine's leaves with *handled as it stands (leave()).
struct SparseJobRun {
    mg_ctx *ctx;
    const mg_table *rows, *cols;
    const uint64_t row_begin, row_end;
    const bool triangle;
    const uint32_t s;
    mg_counts *out_dev;
    const bool force;
    const bool force_join;                                  // MASHGPU_COMPARE_KERNEL=join
    bool *handled;
    SparseJob *job;
    const uint64_t nrows, pairs;
    bool stop = false;

    mg_table::Sparse *ix = nullptr;
    bool clustered = false;
    int rc = MG_OK;
    mg::SparseArgs a;
    DevBuf<uint32_t> q_off, q_img, q_lo, q_hi, q_short, q_short_cnt;
    std::vector<uint32_t> qshort_h, qshort_cnt_h;
    const uint32_t *short_rows_dev = nullptr, *short_rcnt_dev = nullptr;
    uint32_t nshort_rows = 0;
    mg_table::Sparse::Plan *plan = nullptr;
    mg_table::Sparse::Plan fresh;
    bool fresh_kept = false;                                // (its buffers belong to the index once it is in ix->plans)
    bool first = false;
    uint64_t want = 0;
    unsigned long long h[3] = {0, 0, 0};
    bool nothing_to_find = false;
    bool prefilled = false;                                 // the fill runs beside the index build (prefill): its constant, its counter
    uint32_t aside_numer = 0, aside_denom = 0, aside_wgs = 0, aside_naps = 0, aside_threads = 256;
    uint32_t *aside_ctr = nullptr;
    uint32_t aside_copies = 0;                              // the table is nothing but copies of a sketch of this many hashes

    SparseJobRun(mg_ctx *c, const mg_table *r, const mg_table *cl, uint64_t rb, uint64_t re, bool tri, uint32_t sketch_size, mg_counts *out, bool forced,
                 bool forced_join, bool *handled_out, SparseJob *list_job)
        : ctx(c), rows(r), cols(cl), row_begin(rb), row_end(re), triangle(tri), s(sketch_size), out_dev(out), force(forced), force_join(forced_join), handled(handled_out),
          job(list_job), nrows(re - rb), pairs(tri ? (re * (re - 1) / 2 - (rb ? rb * (rb - 1) / 2 : 0)) : (re - rb) * cl->n), q_off(c), q_img(c), q_lo(c),
          q_hi(c), q_short(c), q_short_cnt(c)
    {
        fresh.order = nullptr;
        fresh.dtiles = nullptr;
    }
    // a fresh plan's buffers belong to the index once it is in ix->plans -- and to nobody on every other way out
    ~SparseJobRun()
    {
        if (!fresh_kept) {
            if (fresh.order) ctx_free(ctx, fresh.order);
            if (fresh.dtiles) ctx_free(ctx, fresh.dtiles);
        }
    }
    int leave() { stop = true; return MG_OK; }              // not (or no longer) this engine's job

    int open_index();
    int prefill();
    int aside_start();
    int aside_launch(uint32_t numer, uint32_t denom, uint32_t *ctr);
    void aside_other_constant(uint32_t c);
    int end_prefill(bool finish);
    int row_side();
    int find_plan();
    int keep_plan();
    int join();
    // the context's cost table learns from the phases of a job seen for the first time
    enum { CK_FILL, CK_DISCOVER, CK_MERGE, CK_DENSE, CK_JOIN, CK_N };
    bool clk_used[CK_N] = {false, false, false, false, false};
    double join_steps = 0.0;
    // MASHGPU_COSTS_FIXED=1: the defaults decide and nothing is learned (the test suite: one context serves hundreds of tests, and
    // which engine takes a job of borderline size must not depend on what ran before)
    bool costs_fixed() const { const char *e = ctx_opt(ctx, "MASHGPU_COSTS_FIXED"); return e && atoi(e) != 0; }
    const SparseCosts &costs() const { static const SparseCosts defaults; return costs_fixed() ? defaults : ctx->costs; }
    bool learning() const { return first && !ctx->async && !costs_fixed(); }
    void clk_begin(int k);
    void clk_end(int k);
    void learn();
    int ensure_lists(uint64_t want_cand);
    int discover();
    int choose_engine();
    int fill_and_dense();
    int merge_and_scatter();
    int run();
};

int SparseJobRun::open_index()
{
    *handled = false;
    if (pairs == 0) return leave();
    if (!force && pairs < 4000000ull) return leave();        // small jobs: one tile launch beats an index
    if (const char *e = ctx_opt(ctx, "MASHGPU_COMPARE_SPARSE")) { if (atoi(e) == 0 && !force) return leave(); }
    if (nrows >= (1ull << 31) || cols->n >= (1ull << 31)) return leave();
    if (!mg::sparse_discover_supported((uint32_t)(triangle ? row_end : cols->n))) return leave();
    // the plain full triangle takes the CLUSTERED variant of the index (built on the table with related rows next to each
    // other: dense groups whatever the order of the collection); row ranges, rect and list jobs address table rows
    // ... and so does a job over the table's LAST rows [rb, n) -- what a rank of several is given, its table being the view of the
    // rows below its block's end (tri_view): the clustered order then keeps the rows from rb on in a segment of their own, so the
    // job's rows are a range of the index's rows as well.  Not when the table's plain index exists already (a
    // caller walking the table in blocks: its last block ends at n too).
    clustered = triangle && !job && row_end == cols->n;
    if (clustered && row_begin != 0)
        for (const mg_table::Sparse *have : cols->sparse)
            if (have->s == s && !have->clustered) clustered = false;
    if (const char *e = ctx_opt(ctx, "MASHGPU_COMPARE_CLUSTER")) clustered = clustered && atoi(e) != 0;
    bool cold = true;                                       // no index of this sketch size yet: it is built now
    for (const mg_table::Sparse *have : cols->sparse) cold = cold && have->s != s;
    uint64_t aside_min = 100000000ull;                       // (below: the fill is a fraction of a millisecond)
    if (const char *e = ctx_opt(ctx, "MASHGPU_FILL_ASIDE_MIN_PAIRS")) aside_min = strtoull(e, nullptr, 10);
    if (cold && !job && out_dev && pairs >= aside_min && mg::sparse_fill_chunks(pairs) < mg::kFillStop / 2u && (rc = prefill()) != MG_OK) return rc;
    rc = table_sparse_index(ctx, cols, s, clustered, &ix, clustered ? (uint32_t)row_begin : 0u);
    if (rc != MG_OK) return rc;
    if (!ix->usable && clustered) {                         // (whatever stopped it may not stop the plain variant)
        clustered = false;
        rc = table_sparse_index(ctx, cols, s, false, &ix);
        if (rc != MG_OK) return rc;
    }
    if (!ix->usable) return leave();
    // (list mode: two copies of one sketch are a pair at distance 0 that is no candidate, two EMPTY sketches
    //  likewise -- such tables take the matrix path.  The pairs inside dense groups are no candidates either: they are
    //  appended to their rows' lists by the dense kernel itself, see job_lists)
    if (job && (ix->copies || ix->has_empty)) return leave();
    if (job && !triangle && !ix->dgroups_host.empty()) return leave();
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return MG_OK;
}

// The constant of a matrix job -- {0, s} for every pair that shares no hash -- depends on nothing the index says: while the
// index is built it is written on a stream of its own, 64 KB chunks taken from a counter by a few workgroups -- a quarter of
// the CUs have one: a fill at full speed keeps the memory's queues full of writes and the build's loads, kernels that live
// on round trips, wait behind them (round 5 measured exactly that with 2 - 16 workgroups per CU and dropped it; 64
// workgroups write ~4 TB/s and the build takes 10 ms instead of 8, with the fill's 6.7 ms gone).  What is left when the build
// has ended is taken from the same counter at full speed on the context's stream (end_prefill), so a pace that is too slow
// costs little.  C3 per table: 15.4 -> 11.8 ms (tools/aside_sweep.sh).
// MASHGPU_FILL_ASIDE = "<workgroups>[,<naps of 64 cycles per 4 KB>[,<work-items>]]"; 0: off.
int SparseJobRun::prefill()
{
    uint32_t wgs = std::max(16u, (uint32_t)ctx->cu_count / 4u), naps = 1, threads = 256;
    if (const char *e = ctx_opt(ctx, "MASHGPU_FILL_ASIDE")) {
        if (atoi(e) == 0) return MG_OK;
        if (sscanf(e, "%u,%u,%u", &wgs, &naps, &threads) < 1 || wgs == 0 || (threads != 64 && threads != 128 && threads != 256)) return MG_OK;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->aux) {
        if (hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ctx->aux_go, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->aux_done, hipEventDisableTiming) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&ctx->aux_ctr), 256) != hipSuccess) {
            (void)hipGetLastError();                        // (no second stream: the fill runs behind the build as it always did)
            if (ctx->aux) hipStreamDestroy(ctx->aux);
            if (ctx->aux_go) hipEventDestroy(ctx->aux_go);
            if (ctx->aux_done) hipEventDestroy(ctx->aux_done);
            if (ctx->aux_ctr) hipFree(ctx->aux_ctr);
            ctx->aux = nullptr;
            ctx->aux_go = ctx->aux_done = nullptr;
            ctx->aux_ctr = nullptr;
            return MG_OK;
        }
    }
    aside_wgs = wgs; aside_naps = naps; aside_threads = threads;
    ctx->aside_all_copies = [this](uint32_t c) { aside_other_constant(c); };
    // Where the build is long (C5: 10^9 entries, 79 ms) the fill waits for the bucket sorts: the build's first kernels -- the
    // clustered copy, K1, K3 -- are the ones that move bytes, K4 works in LDS for a third of the build.  (MASHGPU_FILL_ASIDE_AT_SORT:
    // entries from which on; tables the tiles refuse never get there and fill behind their build.)
    uint64_t late_from = 300000000ull;
    if (const char *e = ctx_opt(ctx, "MASHGPU_FILL_ASIDE_AT_SORT")) late_from = strtoull(e, nullptr, 10);
    if (cols->n * std::min<uint64_t>(cols->s, s) >= late_from) {
        ctx->aside_at_sort = [this]() { (void)aside_start(); };
        return MG_OK;
    }
    return aside_start();
}

int SparseJobRun::aside_start()
{
    ctx->aside_at_sort = nullptr;
    if (prefilled) return MG_OK;
    HIP_TRY(ctx, hipMemsetAsync(ctx->aux_ctr, 0, 128, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->aux_go, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux, ctx->aux_go, 0));
    const bool copies = triangle && aside_copies != 0;
    return aside_launch(copies ? aside_copies : 0u, copies ? aside_copies : s, ctx->aux_ctr);
}

int SparseJobRun::aside_launch(uint32_t numer, uint32_t denom, uint32_t *ctr)
{
    prof_begin(ctx, ctx->prof_fill_aside, ctx->aux);
    HIP_TRY(ctx, mg::launch_sparse_fill_chunks(reinterpret_cast<uint2 *>(out_dev), pairs, numer, denom, aside_wgs, aside_naps, ctr, ctx->aux, aside_threads));
    prof_end(ctx, ctx->prof_fill_aside, ctx->aux);
    HIP_TRY(ctx, hipEventRecord(ctx->aux_done, ctx->aux));
    aside_numer = numer; aside_denom = denom; aside_ctr = ctr;
    prefilled = true;
    return MG_OK;
}

// The build has found the table to be nothing but copies of one sketch of c hashes (host_index.cpp: find_copies): every pair is
// {c, c}.  The launch beside the build ends at its next chunk and another one starts behind it, with that constant and a
// counter of its own.  (A triangle job's matter: a rect job's queries are not the table's rows.)
void SparseJobRun::aside_other_constant(uint32_t c)
{
    aside_copies = c;                                       // (a fill that has not started yet starts with it)
    if (!prefilled || !triangle) return;
    if (hipMemsetAsync(aside_ctr, 0x80, 4, ctx->stream) != hipSuccess || aside_launch(c, c, ctx->aux_ctr + 16) != MG_OK) (void)hipGetLastError();
}

// The end of the fill beside the build: what is left of it at full speed on the context's stream (finish), or nothing more --
// somebody else writes the whole output, or the constant was the wrong one -- and the stream waits for the chunks in flight.
int SparseJobRun::end_prefill(bool finish)
{
    ctx->aside_all_copies = nullptr;
    ctx->aside_at_sort = nullptr;
    if (!prefilled) return MG_OK;
    prefilled = false;
    hipError_t e;
    if (finish) e = mg::launch_sparse_fill_chunks(reinterpret_cast<uint2 *>(out_dev), pairs, aside_numer, aside_denom, (uint32_t)ctx->cu_count * 16u, 0u, aside_ctr, ctx->stream);
    else e = hipMemsetAsync(aside_ctr, 0x80, 4, ctx->stream);                // (0x80808080 >= kFillStop)
    const hipError_t e2 = hipStreamWaitEvent(ctx->stream, ctx->aux_done, 0);
    if (e != hipSuccess || e2 != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("compare (fill beside the build): ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return MG_OK;
}

int SparseJobRun::row_side()
{
    // ---- row side ----
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
    a.dn_grp_of = (job && triangle && !ix->dgroups_host.empty()) ? ix->grp_of : nullptr;
    a.dn_groups = a.dn_grp_of ? ix->dgroups : nullptr;
    a.res = nullptr;
    a.seg_base = nullptr;
    a.seg_cnt = nullptr;
    a.chunk_inc = nullptr;
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
            if (Eq >= (1ull << 31)) return leave();
            if (c < s) { qshort_h.push_back((uint32_t)q); qshort_cnt_h.push_back((uint32_t)c); }
            if (c == 0 && job) return leave();
            if (c && rows->last[row_begin + q] == MG_HASH_PAD) return leave();
        }
        qoff[nrows] = (uint32_t)Eq;
        const uint32_t rsq = ix->rs;
        if (nrows * rsq >= (1ull << 32)) return leave();
        if (q_off.alloc(nrows + 1) != hipSuccess || q_img.alloc(nrows * rsq) != hipSuccess || q_lo.alloc(nrows * rsq) != hipSuccess ||
            q_hi.alloc(nrows * rsq) != hipSuccess || q_short.alloc(std::max<size_t>(qshort_h.size(), 1)) != hipSuccess ||
            q_short_cnt.alloc(std::max<size_t>(qshort_h.size(), 1)) != hipSuccess) {
            (void)hipGetLastError();
            return leave();                                   // no memory for the query side: tile engine
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
    return MG_OK;
}

int SparseJobRun::find_plan()
{
    // ---- the plan of a (rows, range) job: its slice of the visiting order; candidates, shared hashes and the engine
    // choice are learned from the first discover launch (rect: the query table may change between calls, so every
    // call is a first call)
    if (triangle)
        for (auto &pl : ix->plans)
            if (pl.rows == (const void *)rows && pl.rb == row_begin && pl.re == row_end && pl.triangle == triangle) plan = &pl;
    fresh.order = nullptr;
    fresh.dtiles = nullptr;
    first = plan == nullptr;
    if (first) {
        fresh.rows = rows; fresh.rb = row_begin; fresh.re = row_end; fresh.triangle = triangle;
        fresh.cand = 0; fresh.shared = 0; fresh.use = true; fresh.use_list = true; fresh.join = false; fresh.order = nullptr;
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
                    for (uint64_t ra = a_lo; ra < a_hi; ra++) fresh.dense_pairs += ra - G.g0;
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
    return MG_OK;
}

// lists of the job (grown on demand, kept with the index)
int SparseJobRun::ensure_lists(uint64_t want_cand)
{
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
}

int SparseJobRun::discover()
{
    if (!first && !force && !(job ? plan->use_list : plan->use)) return leave();      // a job the tile engine was found to do faster
    // first sight of a job: room for one candidate per two index entries, at most 2^27 (2 GB of lists from the pool; C3
    // has one per twenty, the clade table one per two); a job that holds more is discovered twice, the second time
    // with the count the first one left
    want = first ? std::max<uint64_t>(ix->cand_cap, std::min<uint64_t>(pairs, std::min<uint64_t>(std::max<uint64_t>((uint64_t)ix->E / 2, 1u << 16), 1ull << 27)))
                          : plan->cand;
    nothing_to_find = triangle && ix->one_class != 0;      // nothing but copies of one sketch: every pair is inside the class
    for (int attempt = 0; !nothing_to_find; attempt++) {
        rc = ensure_lists(want);
        if (rc != MG_OK) return rc;
        a.cand = ix->cand;
        a.res = ix->res;
        a.cand_cap = ix->cand_cap;
        a.seg_base = ix->seg_base;
        a.seg_cnt = ix->seg_cnt;
        a.chunk_inc = ix->chunk_inc;
        HIP_TRY(ctx, hipMemsetAsync(ix->counters, 0, 4 * 8, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ix->seg_cnt, 0, nrows * 4, ctx->stream));
        prof_begin(ctx, ctx->prof_discover);
        if (attempt == 0) clk_begin(CK_DISCOVER);
        hipError_t e = mg::launch_sparse_discover(a, false, ctx->stream);
        if (attempt == 0) clk_end(CK_DISCOVER);
        prof_end(ctx, ctx->prof_discover);
        if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("compare (discover): ") + hipGetErrorString(e));
        if (!first) break;                                  // a job seen before: its list has the size it needed then (checked at the end)
        HIP_TRY(ctx, hipMemcpyAsync(h, ix->counters, 24, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (h[2] == 0) break;
        if (attempt >= 1) return fail(ctx, MG_ERR_INVALID, "compare: the table changed while it was compared");
        want = h[0];
    }
    return MG_OK;
}

int SparseJobRun::choose_engine()
{
    if (first) {
        fresh.cand = h[0];
        fresh.shared = h[1];
        // seconds, one MI355X (measured: profiles/r03_sparse_phases.txt)
        const double np = (double)pairs;
        const SparseCosts &K = costs();
        const double t_sparse = np * 8.0 / K.fill_bytes_s + (double)fresh.shared * K.discover_per_shared + (double)nrows * s * K.discover_per_entry +
                                (double)fresh.cand * K.merge_per_candidate + K.launches + (triangle ? (double)ix->cls_pairs * 8.0 / K.class_bytes_s : 0.0) +
                                (double)fresh.dense_pairs * K.dense_per_pair;
        const double dense_rate = np < 3.0e8 ? K.tiles_rate_small : np < 2.0e9 ? K.tiles_rate_mid : K.tiles_rate_large;
        // (what a shared hash costs the tile engine: 2.2e-12 s between copies of one sketch, 5.5e-12 inside clades -- and
        //  1.2e-11 in a collection of one species, where every pair shares a few hundred values and no two rows the same
        //  ones (round 5's one_species bracket: 1.58 s for 5.4e8 pairs where the model said 0.33); priced at the upper
        //  middle, the copies and clades having engines of their own by now)
        const double t_dense = np / dense_rate + (double)fresh.shared * K.tiles_per_shared;
        fresh.use = t_sparse < t_dense;
        // a list job fills nothing (and its pairs inside dense groups are appended, not written into a matrix): priced
        // without the 8 B per pair -- its alternative is the blocked matrix path, which pays them all
        fresh.use_list = t_sparse - np * 8.0 / K.fill_bytes_s - (triangle ? (double)ix->cls_pairs * 8.0 / K.class_bytes_s : 0.0) < t_dense;
        if (ctx_opt(ctx, "MASHGPU_SPARSE_DBG"))
            fprintf(stderr, "compare sparse: rows [%llu, %llu) %s: %llu pairs, %llu candidates, %llu shared hashes; model sparse %.3f ms, tiles %.3f ms\n",
                    (unsigned long long)row_begin, (unsigned long long)row_end, triangle ? "triangle" : "rect", (unsigned long long)pairs,
                    (unsigned long long)fresh.cand, (unsigned long long)fresh.shared, t_sparse * 1e3, t_dense * 1e3);
        keep_plan();
    }
    if (!force && !(job ? plan->use_list : plan->use)) return leave();
    return MG_OK;
}

// a fresh plan of a triangle job joins the index's list (rect: the query table may change between calls)
int SparseJobRun::keep_plan()
{
    if (!triangle || fresh_kept) return MG_OK;
    if (ix->plans.size() >= 64) {
        if (ix->plans.front().order) ctx_free(ctx, ix->plans.front().order);
        if (ix->plans.front().dtiles) ctx_free(ctx, ix->plans.front().dtiles);
        ix->plans.erase(ix->plans.begin());
    }
    ix->plans.push_back(fresh);
    fresh_kept = true;
    plan = &ix->plans.back();
    return MG_OK;
}

// ---- the join engine (compare_join.hip): a job in the middle of the similarity range -- so many shared hashes per pair that
// reading them all to FIND the candidates (discovery) already costs more than counting them in rank order does.  Decided
// before anything is discovered: the shared hashes of the job are the sum of the run lengths in the images (one small
// kernel, exact), the cost of the join follows from them and the table's shape.  Taken: the lists of the table's blocks are
// built once per table (rect: the queries' per call), the tiles write EVERY pair of the job, nothing else runs.
struct JoinListBufs {
    mg_ctx *ctx;
    void *bufs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    mg::JoinSide side;
    explicit JoinListBufs(mg_ctx *c) : ctx(c) {}
    ~JoinListBufs() { for (void *q : bufs) if (q) ctx_free(ctx, q); }
    void release_to(void **out) { for (int i = 0; i < 6; i++) { out[i] = bufs[i]; bufs[i] = nullptr; } }
};

static int join_make_lists(mg_ctx *ctx, const uint32_t *img, uint32_t rs, const uint32_t *cnt_off, const uint32_t *rep, uint32_t nrows, uint32_t s,
                           uint32_t E, bool only_shared, JoinListBufs &L)
{
    const uint64_t slots = (uint64_t)nrows * s;
    const uint32_t nblocks = (nrows + mg::join_block_rows() - 1u) / mg::join_block_rows();
    const size_t tb = mg::join_build_temp_bytes(slots);
    DevBuf<unsigned long long> key_a(ctx), key_b(ctx);
    DevBuf<unsigned char> temp(ctx);
    bool ok = key_a.alloc(slots) == hipSuccess && key_b.alloc(slots) == hipSuccess && temp.alloc(std::max<size_t>(tb, 16)) == hipSuccess;
    const size_t want[6] = {(size_t)(slots + 1) * sizeof(uint2), (size_t)slots * 4, (size_t)slots * 4, (size_t)(nblocks + 1) * 4, (size_t)(nblocks + 1) * 4,
                            (size_t)nblocks * mg::join_levels() * 4};
    for (int i = 0; ok && i < 6; i++) ok = ctx_malloc(ctx, &L.bufs[i], want[i]) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); return MG_ERR_NOMEM; }
    const uint32_t *ent = nullptr;
    hipError_t e = mg::join_build_lists(img, rs, cnt_off, rep, nrows, s, E, only_shared, temp, tb, key_a, key_b, static_cast<uint32_t *>(L.bufs[1]),
                                        static_cast<uint32_t *>(L.bufs[2]), static_cast<uint2 *>(L.bufs[0]), static_cast<uint32_t *>(L.bufs[3]),
                                        static_cast<uint32_t *>(L.bufs[4]), static_cast<uint32_t *>(L.bufs[5]), &ent, ctx->stream);
    if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("compare (join lists): ") + hipGetErrorString(e));
    L.side.grp = static_cast<const uint2 *>(L.bufs[0]);
    L.side.ent = ent;
    L.side.goff = static_cast<const uint32_t *>(L.bufs[3]);
    L.side.gend = static_cast<const uint32_t *>(L.bufs[4]);
    L.side.thr = static_cast<const uint32_t *>(L.bufs[5]);
    return MG_OK;
}

int SparseJobRun::join()
{
    // (a LIST job -- thresholded results, the sparse matrix -- wants candidates, not a matrix: the engine does not make lists.  But
    //  where it would take the matrix of the same rows, the list engine is the wrong one too -- every pair a candidate, a merge
    //  each --, so the job is handed back and its caller takes the matrix path in blocks, which reaches this engine per block
    //  and filters on the device)
    if (job && force_join) return MG_OK;
    if (const char *e = ctx_opt(ctx, "MASHGPU_COMPARE_JOIN")) { if (atoi(e) == 0 && !force_join) return MG_OK; }
    const uint32_t B = mg::join_block_rows();
    const uint64_t nside = triangle ? cols->n : nrows;     // rows of the row side's index space
    const bool can = s >= 1 && s <= 65535u && nside * s < (1ull << 32) && cols->n * (uint64_t)s < (1ull << 32) && ix->E >= 1;
    if (!can) return force_join ? fail(ctx, MG_ERR_UNSUPPORTED, "compare: the join engine cannot take this job") : MG_OK;
    bool take = force_join || (plan && !first && plan->join);
    if (!take && first && !force) {
        const SparseCosts &K = costs();
        const double table_pairs = triangle ? (double)cols->n * (double)(cols->n - 1) / 2.0 : (double)pairs;
        // (the index's own statistic -- every value's holders choose 2, before any clipping -- rules most tables out for free;
        //  rect: the queries are not part of it, the count below decides)
        if (triangle && (double)ix->shared < K.join_min_shared_per_pair * table_pairs) return MG_OK;
        DevBuf<unsigned long long> d_sum(ctx);
        if (d_sum.alloc(1) != hipSuccess) { (void)hipGetLastError(); return MG_OK; }
        unsigned long long shared_job = 0;
        HIP_TRY(ctx, mg::launch_join_shared(a.lo_img, a.hi_img, a.lo_shift, a.rs_row, a.off, a.row_begin, a.row_end, d_sum, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&shared_job, d_sum, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const double np = (double)pairs, I = (double)shared_job;
        const double nbr = (double)((nrows + B - 1) / B), nbc = (double)((cols->n + B - 1) / B);
        const double tiles = triangle ? nbr * (double)((row_begin + row_end) / 2 / B + 1) : nbr * nbc;
        const double per_block = (double)ix->E / nbc;                                     // entries of a block bound its groups
        const double build = ix->jn.built ? 0.0 : (double)cols->n * s * K.join_per_slot;
        const double t_join = np * 8.0 / K.fill_bytes_s + I * K.join_per_shared + tiles * 2.0 * per_block / 64.0 * K.join_per_step + build + K.launches;
        // what the inverted index pays at the very least: the fill, and discovery reading every shared hash
        const double t_sparse = np * 8.0 / K.fill_bytes_s + I * K.discover_per_shared + (double)nrows * s * K.discover_per_entry + K.launches;
        take = t_join < 0.8 * t_sparse;
        if (ctx_opt(ctx, "MASHGPU_SPARSE_DBG"))
            fprintf(stderr, "compare join: rows [%llu, %llu) %s: %llu pairs, %llu shared hashes; model join %.3f ms, inverted index at least %.3f ms -> %s\n",
                    (unsigned long long)row_begin, (unsigned long long)row_end, triangle ? "triangle" : "rect", (unsigned long long)pairs,
                    (unsigned long long)shared_job, t_join * 1e3, t_sparse * 1e3, take ? "join" : "no");
        fresh.shared = shared_job;
    }
    if (!take) return MG_OK;
    if ((rc = end_prefill(false)) != MG_OK) return rc;       // (the join engine writes every pair itself)
    if (job) {
        if (first) { fresh.join = true; keep_plan(); }
        return leave();
    }
    // ---- the lists
    const bool only_shared = triangle && ix->copies == 0;   // (a value of ONE row of the index and that row's copies has no shared bit)
    // the whole triangle: the lists on the rows in an order of their own, relatives side by side (compare_join.hip: jn_labels_kernel);
    // a range of rows is a range of the lists' blocks only in the index's order
    // (a job over the table's last rows [rb, n) -- a rank's, on the view of the rows below its block's end -- likewise: those rows
    //  form a segment of their own behind the others, so the job is a range of the lists' rows)
    bool ordered = triangle && row_end == cols->n;
    if (const char *e = ctx_opt(ctx, "MASHGPU_JOIN_ORDER")) ordered = ordered && atoi(e) != 0;
    const uint32_t jsplit = ordered ? a.row_begin : 0u;
    if (ix->jn.built && (ix->jn.only_shared != only_shared || ix->jn.ordered != ordered || ix->jn.split != jsplit)) {
        for (void *&q : ix->jn.bufs) { if (q) ctx_free(ctx, q); q = nullptr; }
        ix->jn.built = false;
    }
    if (!ix->jn.built) {
        prof_begin(ctx, ctx->prof_index);
        const uint32_t n32 = (uint32_t)cols->n;
        void *order_bufs[3] = {nullptr, nullptr, nullptr};                            // perm, src, map
        struct OrderGuard { mg_ctx *c; void **b; ~OrderGuard() { for (int i = 0; i < 3; i++) if (b[i]) ctx_free(c, b[i]); } } og{ctx, order_bufs};
        const uint32_t *src = ix->rep;
        if (ordered) {
            const size_t tb = mg::join_order_temp_bytes(n32);
            DevBuf<unsigned char> temp(ctx);
            DevBuf<uint32_t> lab(ctx), val_a(ctx);
            DevBuf<unsigned long long> key_a(ctx), key_b(ctx);
            bool ok = temp.alloc(std::max<size_t>(tb, 16)) == hipSuccess && lab.alloc(6ull * n32) == hipSuccess && val_a.alloc(n32) == hipSuccess &&
                      key_a.alloc(n32) == hipSuccess && key_b.alloc(n32) == hipSuccess;
            for (int i = 0; ok && i < 3; i++) ok = ctx_malloc(ctx, &order_bufs[i], (size_t)n32 * 4) == hipSuccess;
            if (!ok) { (void)hipGetLastError(); if (force_join) return fail(ctx, MG_ERR_NOMEM, "compare: no device memory for the join order"); return MG_OK; }
            hipError_t e = mg::join_order_rows(ix->code_img, ix->rs, ix->off, ix->rep, ix->inv, ix->gend, ix->sorted_rows, n32, temp, tb, lab, key_a, key_b, val_a,
                                               static_cast<uint32_t *>(order_bufs[0]), static_cast<uint32_t *>(order_bufs[1]),
                                               static_cast<uint32_t *>(order_bufs[2]), ctx->stream, jsplit);
            if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("compare (join order): ") + hipGetErrorString(e));
            src = static_cast<const uint32_t *>(order_bufs[1]);
        }
        JoinListBufs L(ctx);
        rc = join_make_lists(ctx, ix->code_img, ix->rs, ix->off, src, n32, s, ix->E, only_shared, L);
        prof_end(ctx, ctx->prof_index);
        if (rc == MG_ERR_NOMEM) { if (force_join) return fail(ctx, rc, "compare: no device memory for the join lists"); return MG_OK; }
        if (rc != MG_OK) return rc;
        ix->jn.side = L.side;
        L.release_to(ix->jn.bufs);
        for (int i = 0; i < 3; i++) { ix->jn.bufs[6 + i] = order_bufs[i]; order_bufs[i] = nullptr; }
        ix->jn.src = ordered ? static_cast<const uint32_t *>(ix->jn.bufs[7]) : nullptr;
        ix->jn.map = ordered ? static_cast<const uint32_t *>(ix->jn.bufs[8]) : nullptr;
        ix->jn.only_shared = only_shared;
        ix->jn.ordered = ordered;
        ix->jn.split = jsplit;
        ix->jn.built = true;
    }
    JoinListBufs Q(ctx);
    mg::JoinArgs j;
    j.cols = ix->jn.side;
    if (triangle) {
        j.rows = ix->jn.side;
    } else {
        // the queries' lists: their codes are 2 x (position in the table's sorted values) + 1 where the table holds the value
        rc = join_make_lists(ctx, a.row_img, a.rs_row, a.off, nullptr, (uint32_t)nrows, s, ix->E, true, Q);
        if (rc == MG_ERR_NOMEM) { if (force_join) return fail(ctx, rc, "compare: no device memory for the join lists"); return MG_OK; }
        if (rc != MG_OK) return rc;
        j.rows = Q.side;
    }
    if (ctx_opt(ctx, "MASHGPU_JOIN_NO_EARLY_STOP")) j.rows.thr = j.cols.thr = nullptr;
    j.row_cnt_off = a.off;
    j.col_cnt_off = ix->off;
    j.col_rep = ix->jn.ordered ? ix->jn.src : ix->rep;
    j.rep = triangle ? j.col_rep : nullptr;
    j.inv = triangle ? (ix->jn.ordered ? ix->jn.map : ix->inv) : nullptr;
    j.out = reinterpret_cast<uint2 *>(out_dev);
    j.out_base = a.out_base;
    j.ncols = (uint32_t)cols->n;
    j.row_begin = a.row_begin;
    j.row_end = a.row_end;
    j.bi0 = a.row_begin / B;
    const uint64_t bi1 = ((uint64_t)a.row_end + B - 1) / B;
    j.ncb = (uint32_t)((cols->n + B - 1) / B);
    j.triangle = triangle ? 1u : 0u;
    j.s = s;
    if (const char *e = ctx_opt(ctx, "MASHGPU_JOIN_TILES_PER_WG")) j.tiles_per_wg = atoi(e) == 4 ? 4u : 1u;      // (A/B)
    j.ntiles = triangle ? bi1 * (bi1 + 1) / 2 - (uint64_t)j.bi0 * (j.bi0 + 1) / 2 : (bi1 - j.bi0) * (uint64_t)j.ncb;
    prof_begin(ctx, ctx->prof_join);
    clk_begin(CK_JOIN);
    hipError_t e = mg::launch_join_tiles(j, ctx->stream);
    clk_end(CK_JOIN);
    prof_end(ctx, ctx->prof_join);
    join_steps = (double)j.ntiles * 2.0 * ((double)ix->E / (double)std::max<uint32_t>(j.ncb, 1u)) / 64.0;
    if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("compare (join): ") + hipGetErrorString(e));
    if (first) { fresh.join = true; fresh.use = true; keep_plan(); }
    *handled = true;
    if (!ctx->async || !triangle) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // (rect: the queries' lists go back to the pool)
    return leave();
}

int SparseJobRun::fill_and_dense()
{
    // ---- fill.  The candidates' results are kept in list order and scattered into the output after it.
    // (Beside the index build the fill runs on a stream of its own, see prefill: then what is left of it ends here.)
    // (Side by side with discover + merge on a second stream the fill was MEASURED to gain nothing -- discover's
    // loads queue behind 40 GB of writes, and a kernel that merely ends under the fill waits milliseconds for the
    // L2's write-back, profiles/r03_sparse_phases.json, r03_overlap_trace.txt -- so the phases run one after the other.)
    if (!job) {
        prof_begin(ctx, ctx->prof_fill);
        // a table of n copies of one sketch: the fill IS the answer, {c, c} in every slot, written once
        const bool all_copies = triangle && ix->one_class != 0;
        hipError_t e = hipSuccess;
        // (beside the index build the fill has run already, with the right constant unless the job's rows are not the table's)
        const bool beside = prefilled && (all_copies ? aside_numer == ix->one_class && aside_denom == ix->one_class : aside_numer == 0u);
        if ((rc = end_prefill(beside)) != MG_OK) return rc;
        if (!beside) {
            clk_begin(CK_FILL);
            e = all_copies ? mg::launch_sparse_fill_value(a.out, pairs, ix->one_class, ix->one_class, 16u, (uint32_t)ctx->cu_count, ctx->stream)
                           : mg::launch_sparse_fill_value(a.out, pairs, 0u, s, 16u, (uint32_t)ctx->cu_count, ctx->stream);
            clk_end(CK_FILL);
        }
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
            clk_begin(CK_DENSE);
            e = mg::launch_dense_pairs(plan->dtiles, plan->ndtiles, plan->dtile_rows, ix->dgroups, ix->gdata, ix->dn_lists, ix->ext, ix->dn_xs, s, ix->dn_wmax,
                                       a.row_begin, a.row_end, a.out_base, a.inv, a.out, ctx->stream);
            clk_end(CK_DENSE);
            prof_end(ctx, ctx->prof_dense);
            if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("compare (dense groups): ") + hipGetErrorString(e));
        }
    }
    *handled = true;
    if (job) {
        job->ix = ix;
        job->cand = plan->cand;
        job->args = a;
        job->dtiles = plan->dtiles;
        job->ndtiles = plan->ndtiles;
        job->dtile_rows = plan->dtile_rows;
        job->dense_pairs = plan->ndtiles ? plan->dense_pairs : 0;
    }
    if (plan->cand == 0) return leave();
    return MG_OK;
}

int SparseJobRun::merge_and_scatter()
{
    // ---- merge ----
    bool by_rows = mg::sparse_merge_rows_supported(a.rs_row);
    if (const char *ev = ctx_opt(ctx, "MASHGPU_SPARSE_MERGE")) by_rows = by_rows && strcmp(ev, "lanes") != 0;
    prof_begin(ctx, ctx->prof_merge);
    clk_begin(CK_MERGE);
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
    clk_end(CK_MERGE);
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

void SparseJobRun::clk_begin(int k)
{
    if (!learning()) return;
    mg_ctx::CostClock &c = ctx->cost_clk[k];
    if (!c.a && (hipEventCreate(&c.a) != hipSuccess || hipEventCreate(&c.b) != hipSuccess)) { (void)hipGetLastError(); c.a = c.b = nullptr; return; }
    if (hipEventRecord(c.a, ctx->stream) == hipSuccess) clk_used[k] = true;
}

void SparseJobRun::clk_end(int k)
{
    if (clk_used[k] && hipEventRecord(ctx->cost_clk[k].b, ctx->stream) != hipSuccess) clk_used[k] = false;
}

// prices per unit from the phases just timed: half way from what the context believed to what it measured, never further than
// a factor of four from the defaults (one odd table must not turn the dispatch over); phases too short to say anything are skipped
void SparseJobRun::learn()
{
    bool any = false;
    for (bool u : clk_used) any = any || u;
    if (!any || !plan) return;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipGetLastError(); return; }
    const SparseCosts D;                                    // the defaults
    SparseCosts &K = ctx->costs;
    auto secs = [&](int k) -> double {
        float ms = 0.f;
        if (!clk_used[k] || hipEventElapsedTime(&ms, ctx->cost_clk[k].a, ctx->cost_clk[k].b) != hipSuccess) { (void)hipGetLastError(); return 0.0; }
        return (double)ms * 1e-3;
    };
    auto move = [](double &c, double measured, double def) {
        const double v = 0.5 * c + 0.5 * measured;
        c = std::min(std::max(v, def / 4.0), def * 4.0);
    };
    // (a phase below a millisecond is its launch, its first wave's round trips and a tail: on a table of a few thousand rows
    //  every price came out three to four times too high, and the list engine lost a job it does in half the time --
    //  test_survivor_lists_from_candidates_equal_the_matrix_path; the phases that decide anything are the long ones)
    const double floor_s = 1.0e-3;
    double t;
    if ((t = secs(CK_FILL)) > floor_s) { double bps = K.fill_bytes_s; move(bps, (double)pairs * 8.0 / t, D.fill_bytes_s); K.fill_bytes_s = bps; }
    if ((t = secs(CK_DISCOVER)) > floor_s) {
        const double model = (double)plan->shared * K.discover_per_shared + (double)nrows * s * K.discover_per_entry;
        if (model > 0) { const double r = t / model; move(K.discover_per_shared, K.discover_per_shared * r, D.discover_per_shared); move(K.discover_per_entry, K.discover_per_entry * r, D.discover_per_entry); }
    }
    if ((t = secs(CK_MERGE)) > floor_s && plan->cand >= 10000) move(K.merge_per_candidate, t / (double)plan->cand, D.merge_per_candidate);
    if ((t = secs(CK_DENSE)) > floor_s && plan->dense_pairs >= 100000) move(K.dense_per_pair, t / (double)plan->dense_pairs, D.dense_per_pair);
    if ((t = secs(CK_JOIN)) > floor_s) {
        const double model = (double)plan->shared * K.join_per_shared + join_steps * K.join_per_step;
        if (model > 0) { const double r = t / model; move(K.join_per_shared, K.join_per_shared * r, D.join_per_shared); move(K.join_per_step, K.join_per_step * r, D.join_per_step); }
    }
    ctx->cost_updates++;
    if (ctx_opt(ctx, "MASHGPU_SPARSE_DBG"))
        fprintf(stderr, "compare costs (context, after %llu jobs): fill %.3g B/s, discover %.3g s per shared hash + %.3g per entry, merge %.3g per candidate, dense %.3g per pair, join %.3g per shared hash + %.3g per step\n",
                (unsigned long long)ctx->cost_updates, K.fill_bytes_s, K.discover_per_shared, K.discover_per_entry, K.merge_per_candidate, K.dense_per_pair,
                K.join_per_shared, K.join_per_step);
}

int SparseJobRun::run()
{
    struct Learn { SparseJobRun *r; ~Learn() { r->learn(); } } learn_at_exit{this};
    // (whoever takes the job from here -- another engine, an error -- finds no write to the output in flight)
    struct Aside { SparseJobRun *r; ~Aside() { (void)r->end_prefill(false); } } aside_at_exit{this};
    if ((rc = open_index()) != MG_OK || stop) return rc;
    if ((rc = row_side()) != MG_OK || stop) return rc;
    if ((rc = find_plan()) != MG_OK || stop) return rc;
    if ((rc = join()) != MG_OK || stop) return rc;
    if (force_join) return fail(ctx, MG_ERR_UNSUPPORTED, "compare: the join engine cannot take this job");
    if ((rc = discover()) != MG_OK || stop) return rc;
    if ((rc = choose_engine()) != MG_OK || stop) return rc;
    if ((rc = fill_and_dense()) != MG_OK || stop) return rc;
    return merge_and_scatter();
}

static int run_compare_sparse(mg_ctx *ctx, const mg_table *rows, const mg_table *cols, uint64_t row_begin, uint64_t row_end,
                              bool triangle, uint32_t s, mg_counts *out_dev, bool force, bool *handled, SparseJob *job = nullptr, bool force_join = false)
{
    *handled = false;
    SparseJobRun run(ctx, rows, cols, row_begin, row_end, triangle, s, out_dev, force, force_join, handled, job);
    return run.run();
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
    if (!force || strcmp(force, "sparse") == 0 || strcmp(force, "join") == 0) {
        bool handled = false;
        const int rcs = run_compare_sparse(ctx, rows, cols, row_begin, row_end, triangle, a.s, out_dev, force != nullptr, &handled, nullptr,
                                           force && strcmp(force, "join") == 0);
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

uint64_t tri_pairs(uint64_t row_begin, uint64_t row_end)
{
    // sum_{i=row_begin}^{row_end-1} i
    auto tri = [](uint64_t x) { return x ? x * (x - 1) / 2 : 0; };
    return tri(row_end) - tri(row_begin);
}

// A triangle call over rows [rb, re) looks at rows below re only (CommandTriangle.cpp:200-214: row i against the rows j < i): what
// the engines derive from the table -- the inverted index above all, 8 of C3's 15 ms -- is derived from the first re rows.  One of
// G ranks with equal-area blocks thereby builds an index of sqrt((g + 1) / G) of the table instead of all of it (rank 0 of 8:
// 35 %; the mean: 70 %), and rank 0's job is the WHOLE triangle of its view: the clustered index, dense groups whatever the
// collection's order, the join engine's ordered lists.  Per CALL, not per internal block of a call: the blocks of one call
// share one view.  (MASHGPU_TRI_PREFIX=0: off.)
static const mg_table *tri_view(mg_ctx *ctx, const mg_table *t, uint64_t rb, uint64_t re)
{
    if (re > t->n) re = t->n;
    if (rb >= re || re == t->n || tri_pairs(rb, re) < 4000000ull) return t;
    if (const char *e = ctx_opt(ctx, "MASHGPU_TRI_PREFIX")) { if (atoi(e) == 0) return t; }
    // One view per table: a rank's calls come with the same range every time, but a caller that walks the table in blocks
    // of rows (the CLI: 2^24 pairs a call) would pay an index per block -- its first block makes a (small) view, the second
    // finds no view that covers it and takes the whole table, whose index then serves every later block.
    if (!t->sparse.empty()) return t;                       // (the whole table's index exists already)
    const mg_table *best = nullptr;
    for (const mg_table *v : t->prefix_views)
        if (v->n >= re && (!best || v->n < best->n)) best = v;
    if (best) return best;
    if (!t->prefix_views.empty()) return t;
    return table_prefix_view(t, re);
}

int mg_compare_tri_dev(mg_ctx *ctx, const mg_table *t, uint64_t row_begin, uint64_t row_end, mg_counts *out_dev)
{
    if (!ctx) return MG_ERR_INVALID;
    if (!t || !out_dev) return fail(ctx, MG_ERR_INVALID, "mg_compare_tri_dev: NULL argument");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    t = tri_view(ctx, t, row_begin, row_end);
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
    if (triangle && rows == cols) rows = cols = tri_view(ctx, cols, rb, re);
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

// kmer_size == 0: no distance filter -- every pair that shares a hash inside its first s union elements (numer >= 1) passes
static int compare_filter(mg_ctx *ctx, const mg_table *rows, const mg_table *cols, uint64_t rb, uint64_t re,
                          bool triangle, int kmer_size, double max_distance, mg_edge *out_host, uint64_t capacity,
                          uint64_t *count_out)
{
    *count_out = 0;
    if (re > rows->n) re = rows->n;
    if (triangle && rows == cols) rows = cols = tri_view(ctx, cols, rb, re);     // (the first re rows are all this call looks at)
    if (rb >= re) return MG_OK;
    if (kmer_size < 0) return fail(ctx, MG_ERR_INVALID, "compare filter: bad k-mer size");
    const uint64_t s64 = std::min(rows->s, cols->s);
    if (s64 > 0xFFFFFFFEull) return fail(ctx, MG_ERR_INVALID, "compare: sketch size too large");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> min_numer;
    if (kmer_size == 0) min_numer.assign((size_t)s64 + 1, 1u);
    else build_min_numer((uint32_t)s64, kmer_size, max_distance, min_numer);

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
    if (kmer_size < 1) return fail(ctx, MG_ERR_INVALID, "compare filter: bad k-mer size");
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
    if (kmer_size < 1) return fail(ctx, MG_ERR_INVALID, "compare filter: bad k-mer size");
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
    if (triangle && rows == cols) rows = cols = tri_view(ctx, cols, rb, re);     // (the first re rows are all this call looks at)
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
// The lists of a finished list job in REFERENCE order (rows ascending, a row's pairs by column): the candidates' {row, col}
// and {common, denom} gathered row by row, and behind a grouped row's candidates -- they all lie below its group -- its pairs
// inside the group, computed into their places by the dense kernel.  K = job.cand + job.dense_pairs entries.
static int job_lists(mg_ctx *ctx, const SparseJob &job, uint32_t nrows, uint32_t row_add, uint32_t s, uint32_t *d_byrow, uint32_t *d_base, void *d_temp,
                     size_t tb, uint2 *d_rc, uint2 *d_cnt)
{
    HIP_TRY(ctx, hipMemsetAsync(d_byrow, 0, (size_t)nrows * 4, ctx->stream));
    HIP_TRY(ctx, mg::launch_sparse_gather_rows(job.args, d_byrow, d_base, d_temp, tb, row_add, d_rc, d_cnt, ctx->stream));
    if (job.ndtiles) {
        mg_table::Sparse *ix = job.ix;
        mg::DenseList L;
        L.row_base = d_base;
        L.row_cnt = d_byrow;
        L.rc = d_rc;
        L.counts = d_cnt;
        L.row_first = job.args.row_begin;
        HIP_TRY(ctx, mg::launch_dense_pairs(job.dtiles, job.ndtiles, job.dtile_rows, ix->dgroups, ix->gdata, ix->dn_lists, ix->ext, ix->dn_xs, s, ix->dn_wmax,
                                            job.args.row_begin, job.args.row_end, job.args.out_base, nullptr, nullptr, ctx->stream, &L));
    }
    return MG_OK;
}

static int compare_results(mg_ctx *ctx, const mg_table *rows, const mg_table *cols, uint64_t rb, uint64_t re, bool triangle,
                           int kmer_size, double kmer_space, double max_d, double max_p, mg_result *out_host, uint64_t capacity,
                           uint64_t *count_out)
{
    *count_out = 0;
    if (re > rows->n) re = rows->n;
    if (triangle && rows == cols) rows = cols = tri_view(ctx, cols, rb, re);     // (the first re rows are all this call looks at)
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
        const uint64_t K = handled ? job.cand + job.dense_pairs : 0;
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
            rc = job_lists(ctx, job, nrows, triangle ? 0u : (uint32_t)rb, s, d_byrow, d_base, d_temp, tb, d_rc, d_cnt);
            if (rc != MG_OK) return rc;
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

// ---- the whole matrix as the pairs that are NOT {0, min(s, |A| + |B|)} (include/mashgpu.h: mg_compare_tri_sparse_host)
static int compare_sparse_matrix(mg_ctx *ctx, const mg_table *rows, const mg_table *cols, uint64_t rb, uint64_t re, bool triangle, mg_edge *out_host,
                                 uint64_t capacity, uint64_t *count_out)
{
    *count_out = 0;
    if (re > rows->n) re = rows->n;
    if (triangle && rows == cols) rows = cols = tri_view(ctx, cols, rb, re);     // (the first re rows are all this call looks at)
    if (rb >= re) return MG_OK;
    const uint64_t s64 = std::min(rows->s, cols->s);
    if (s64 > 0xFFFFFFFEull) return fail(ctx, MG_ERR_INVALID, "compare: sketch size too large");
    const uint32_t s = (uint32_t)s64;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t all_pairs = triangle ? tri_pairs(rb, re) : (re - rb) * cols->n;
    if (all_pairs == 0) return MG_OK;
    const char *force_kernel = ctx_opt(ctx, "MASHGPU_COMPARE_KERNEL");
    if ((!force_kernel || strcmp(force_kernel, "sparse") == 0) && !ctx_opt(ctx, "MASHGPU_RESULTS_MATRIX")) {
        // the index engine's lists ARE the answer: its candidates (pairs that share a hash) and the pairs inside its dense
        // groups, in reference order, minus the few that share a hash only behind their first s union elements
        SparseJob job;
        bool handled = false;
        int rc = run_compare_sparse(ctx, rows, cols, rb, re, triangle, s, nullptr, force_kernel != nullptr, &handled, &job);
        if (rc != MG_OK) return rc;
        const uint64_t K = handled ? job.cand + job.dense_pairs : 0;
        if (handled && K == 0) return MG_OK;
        const uint32_t nrows = (uint32_t)(re - rb);
        DevBuf<uint2> d_rc(ctx), d_cnt(ctx);
        DevBuf<uint32_t> d_byrow(ctx), d_base(ctx), d_bc(ctx), d_bo(ctx);
        DevBuf<uint4> d_edges(ctx);
        DevBuf<unsigned long long> d_total(ctx);
        DevBuf<unsigned char> d_temp(ctx);
        const size_t tb = std::max(mg::sparse_gather_temp_bytes(nrows), mg::sparse_edges_temp_bytes(K));
        if (handled && K < (1ull << 32) && d_rc.alloc(K) == hipSuccess && d_cnt.alloc(K) == hipSuccess && d_byrow.alloc(nrows) == hipSuccess &&
            d_base.alloc(nrows) == hipSuccess && d_temp.alloc(std::max<size_t>(tb, 16)) == hipSuccess && d_bc.alloc(mg::sparse_edges_blocks(K)) == hipSuccess &&
            d_bo.alloc(mg::sparse_edges_blocks(K)) == hipSuccess && d_edges.alloc(K) == hipSuccess && d_total.alloc(1) == hipSuccess) {
            rc = job_lists(ctx, job, nrows, triangle ? 0u : (uint32_t)rb, s, d_byrow, d_base, d_temp, tb, d_rc, d_cnt);
            if (rc != MG_OK) return rc;
            unsigned long long total = 0;
            HIP_TRY(ctx, mg::launch_sparse_list_edges(d_rc, d_cnt, K, d_bc, d_bo, d_temp, tb, d_edges, d_total, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            *count_out = total;
            if (total > capacity) return fail(ctx, MG_ERR_NOMEM, "compare: more pairs that share a hash than `capacity` (see *count_out)");
            if (total) {
                HIP_TRY(ctx, hipMemcpyAsync(out_host, d_edges, total * sizeof(mg_edge), hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            }
            return MG_OK;
        }
        (void)hipGetLastError();
    }
    // tables the list engine leaves alone (copies of rows, empty rows, nearly every pair related): the matrix in blocks, filtered on the device
    return compare_filter(ctx, rows, cols, rb, re, triangle, 0, 0.0, out_host, capacity, count_out);
}

int mg_compare_tri_sparse_host(mg_ctx *ctx, const mg_table *t, uint64_t row_begin, uint64_t row_end, mg_edge *out_host, uint64_t capacity,
                               uint64_t *count_out)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!t || !count_out || (!out_host && capacity)) return fail(ctx, MG_ERR_INVALID, "mg_compare_tri_sparse_host: null argument");
    return compare_sparse_matrix(ctx, t, t, row_begin, row_end, true, out_host, capacity, count_out);
}

int mg_compare_rect_sparse_host(mg_ctx *ctx, const mg_table *ref, const mg_table *qry, uint64_t q_begin, uint64_t q_end, mg_edge *out_host,
                                uint64_t capacity, uint64_t *count_out)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!ref || !qry || !count_out || (!out_host && capacity)) return fail(ctx, MG_ERR_INVALID, "mg_compare_rect_sparse_host: null argument");
    return compare_sparse_matrix(ctx, qry, ref, q_begin, q_end, false, out_host, capacity, count_out);
}

int mg_expand_tri_sparse(const mg_edge *edges, uint64_t count, const uint32_t *nhash, uint64_t sketch_size, uint64_t row_begin, uint64_t row_end,
                         mg_counts *out_host)
{
    if ((!edges && count) || !nhash || !out_host || row_begin > row_end) return MG_ERR_INVALID;
    const uint64_t base = tri_pairs(0, row_begin);
    finish_parallel(row_end - row_begin, [=](uint64_t kb, uint64_t ke) {
        for (uint64_t i = row_begin + kb; i < row_begin + ke; i++) {
            const uint64_t ni = std::min<uint64_t>(nhash[i], sketch_size);
            mg_counts *row = out_host + ((i ? i * (i - 1) / 2 : 0) - base);
            for (uint64_t j = 0; j < i; j++) {
                row[j].numer = 0;
                row[j].denom = (uint32_t)std::min<uint64_t>(sketch_size, ni + std::min<uint64_t>(nhash[j], sketch_size));
            }
        }
    });
    for (uint64_t e = 0; e < count; e++) {
        const mg_edge &x = edges[e];
        if (x.row < row_begin || x.row >= row_end || x.col >= x.row) return MG_ERR_INVALID;
        mg_counts &c = out_host[(uint64_t)x.row * (x.row - 1) / 2 - base + x.col];
        c.numer = x.numer;
        c.denom = x.denom;
    }
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

