// host_sketch.cpp -- the sketching entry points of include/mashgpu.h: batch, packed input, streamed sessions, reads mode
#include "host_internal.h"

/* ------------------------------------------------------------------ sketching */

bool alphabet_is_dna(const mg_params *p)
{
    if (p->alphabet_size != 4) return false;
    return p->alphabet['A'] && p->alphabet['C'] && p->alphabet['G'] && p->alphabet['T'];
}

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
    mg::SketchArgs a{};
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

int sketch_dev_impl(mg_ctx *ctx, const mg_params *p, const uint8_t *bases_dev, uint64_t nbases,
                           const uint64_t *sketch_off, uint64_t nsketch, uint64_t *hashes_out_dev,
                           uint32_t *nhash_out_dev, uint32_t *counts_out_dev, const ProbeHook *probe, const PackedSrc *pk)
{
    if (!ctx) return MG_ERR_INVALID;
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (!p || !sketch_off || !hashes_out_dev || !nhash_out_dev || (!bases_dev && nbases && !pk))
        return fail(ctx, MG_ERR_INVALID, "mg_sketch: NULL argument");
    if (pk && (counts_out_dev || probe || p->min_copies > 1 || !alphabet_is_dna(p)))
        return fail(ctx, MG_ERR_UNSUPPORTED, "mg_sketch: packed bases in the kernel serve plain nucleotide sketches only");
    // (what the unpack kernel's launcher checked before the sketch kernel read packed bases itself: ADVICE r5)
    if (pk && (!pk->packed || pk->skip >= 16u || pk->mskip >= 32u || ((uintptr_t)pk->packed & 3) != 0 || ((uintptr_t)pk->mask & 3) != 0))      // (mask: nullable)
        return fail(ctx, MG_ERR_INVALID, "mg_sketch: packed bases must be 4-byte aligned words, base skip < 16, mask skip < 32");
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
    a.packed = pk ? pk->packed : nullptr;
    a.pmask = pk ? pk->mask : nullptr;
    a.pskip = pk ? pk->skip : 0;
    a.pmskip = pk ? pk->mskip : 0;
    if (pk && range_path) return fail(ctx, MG_ERR_UNSUPPORTED, "mg_sketch: packed bases in the kernel need the LDS selector's sketch sizes");
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
    // plain sketches: the kernel reads the packed arrays itself; multiplicities, min_copies and sketch sizes beyond the LDS
    // selector go through an ASCII copy (unpack_bases_kernel) and the ordinary path.  MASHGPU_PACKED_UNPACK=1: always the copy.
    int nt_unused = 0;
    uint32_t cap_unused = 0;
    const bool direct = !d_counts && p->min_copies <= 1 && alphabet_is_dna(p) && mg::sketch_geometry(p->sketch_size, &nt_unused, &cap_unused) &&
                        !ctx_opt(ctx, "MASHGPU_PACKED_UNPACK");
    DevBuf<uint8_t> d_ascii(ctx), d_pk[2] = {DevBuf<uint8_t>(ctx), DevBuf<uint8_t>(ctx)}, d_mk[2] = {DevBuf<uint8_t>(ctx), DevBuf<uint8_t>(ctx)};
    if (!direct && d_ascii.alloc(((longest + 15u) & ~15ull) + 64u) != hipSuccess) return fail(ctx, MG_ERR_NOMEM, "mg_sketch_packed: device allocation failed");
    hipStream_t copy_stream = nullptr;
    struct StreamGuard { hipStream_t *s; ~StreamGuard() { if (*s) hipStreamDestroy(*s); } } stream_guard{&copy_stream};
    if (host_input) {
        for (int k = 0; k < 2; k++)
            if (d_pk[k].alloc(longest / 4u + 32u) != hipSuccess || (mask && d_mk[k].alloc(longest / 8u + 32u) != hipSuccess))
                return fail(ctx, MG_ERR_NOMEM, "mg_sketch_packed: device allocation failed");
        HIP_TRY(ctx, hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        // The staging buffers come from the context's pool, whose rule is stream order: a block handed back while work on it
        // is still queued on the context's stream (mg_ctx_set_async: a compare call may have returned with kernels pending on
        // its scratch) is only safe for work queued BEHIND that work.  The copy stream is another stream, so it is put behind
        // everything the context's stream holds right now.
        hipEvent_t ev = nullptr;
        HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(copy_stream, ev, 0);
        (void)hipEventDestroy(ev);                          // (released once the wait has been served)
        if (e != hipSuccess) return fail(ctx, MG_ERR_HIP, std::string("mg_sketch_packed: ") + hipGetErrorString(e));
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
        bool next_copied_here = false;
        if (host_input && c + 1 < pieces.size()) {
            try {
                next = std::thread(copy_piece, std::cref(pieces[c + 1]), (int)((c + 1) & 1), &next_err);
            } catch (const std::system_error &) {
                next_copied_here = true;                     // no thread to be had: the copy follows this piece, unoverlapped
            }
        }
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
        off.resize(q.i1 - q.i0 + 1);
        for (uint64_t i = q.i0; i <= q.i1; i++) off[i - q.i0] = sketch_off[i] - q.b0;
        int rc;
        if (direct) {
            // the sketch kernel expands the codes itself while it stages its tiles (round 5): no ASCII copy in HBM
            const PackedSrc pks{reinterpret_cast<const uint32_t *>(src_pk), reinterpret_cast<const uint32_t *>(src_mk), skip, mskip};
            rc = sketch_dev_impl(ctx, p, nullptr, len, off.data(), q.i1 - q.i0, d_hashes + q.i0 * s, d_nhash + q.i0, nullptr, nullptr, &pks);
        } else {
            HIP_TRY(ctx, mg::launch_unpack_bases(src_pk, src_mk, skip, mskip, len, d_ascii, ctx->stream));
            rc = sketch_dev_impl(ctx, p, d_ascii, len, off.data(), q.i1 - q.i0, d_hashes + q.i0 * s, d_nhash + q.i0,
                                 d_counts ? d_counts + q.i0 * s : nullptr, nullptr);
        }
        if (rc != MG_OK) return rc;
        if (next.joinable()) next.join();
        if (next_copied_here) copy_piece(pieces[c + 1], (int)((c + 1) & 1), &next_err);
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

