// host_screen.cpp -- screening (mash screen), on one GPU and on several
#include "host_internal.h"

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

