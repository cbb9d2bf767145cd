// dense_emu_main.cpp -- runs the dense-group kernels of mash_amd/csrc/compare_dense.hip (dn_encode_kernel, dn_pairs_kernel) on
// host threads (tools/hipemu) and compares every pair inside every group with the loop of compareSketches
// (CommandDistance.cpp:347-385, restated below on the rows' hashes).  The index the kernels read (code and position images,
// the groups' universes and leaders) is made here by a std::stable_sort, as index_emu_main.cpp states it.
// TEST INFRASTRUCTURE (tests/test_dense_emu.py); built with g++.
//
//   dense_emu <case> ...      cases: see main(); exit status 0 = every case agrees
#include "../../tools/hipemu/hipemu.h"

#include <algorithm>
#include <random>
#include <string>
#include <vector>

#include "../../mash_amd/csrc/compare_dense.hip"

using namespace mg;

struct Table {
    uint32_t n = 0, s = 0;
    std::vector<std::vector<uint64_t>> rows;          // ascending, distinct, at most s values each
    std::vector<std::pair<uint32_t, uint32_t>> groups; // row intervals [g0, g1)
};

static std::vector<uint64_t> finish_row(std::vector<uint64_t> v, uint32_t keep)
{
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    if (v.size() > keep) v.resize(keep);
    return v;
}

// kind: "near" (rows keep 96 % of a pool, 3 % private), "loose" (70 % of a wider pool + 30 % private: dozens of extras per
// word), "clumped" / "clumped12" / "clumped5" (20 / 12 / 5 private values between two pool values: more than fifteen extras
// in one gap -- the flagged words -- or counts that need the higher bit planes),
// "short" (rows of a third to all of s values: unions that end before s), "wide" (a universe of several words beyond s)
static Table make_table(uint32_t s, uint64_t seed, const std::string &kind)
{
    Table t;
    t.s = s;
    std::mt19937_64 rng(seed);
    const uint64_t top = 1ull << 54;
    auto single = [&]() {
        std::vector<uint64_t> v;
        for (uint32_t k = 0; k < s + 8; k++) v.push_back(rng() % top);
        t.rows.push_back(finish_row(v, s));
    };
    auto group = [&](uint32_t m, double pool_factor, double keep, double priv, uint32_t clump, bool shorten) {
        std::vector<uint64_t> pool;
        for (uint32_t k = 0; k < (uint32_t)(pool_factor * s) + 8; k++) pool.push_back(rng() % top);
        std::sort(pool.begin(), pool.end());
        const uint32_t g0 = (uint32_t)t.rows.size();
        for (uint32_t i = 0; i < m; i++) {
            std::vector<uint64_t> v;
            for (uint64_t x : pool)
                if ((double)(rng() % 10000) < keep * 10000.0) v.push_back(x);
            const uint32_t np = std::max<uint32_t>(1, (uint32_t)(priv * s));
            if (clump && i % 3 == 0) {
                // (the gap lies in the word in which a pair of these rows reaches its s-th union element: only there are the
                //  extras of a word counted gap by gap)
                const size_t at = pool.size() * 4 / 5;
                const uint64_t lo = pool[at], hi = pool[at + 1];
                // (every row its own values: shared ones would belong to the universe and be no extras)
                const uint64_t step = std::max<uint64_t>((hi - lo - 1) / ((uint64_t)(clump + 1) * 64u), 1);
                for (uint32_t k = 0; k < clump; k++) v.push_back(lo + 1 + ((uint64_t)k * 64u + i) * step);
            }
            for (uint32_t k = 0; k < np; k++) v.push_back(rng() % top);
            uint32_t keep_n = s;
            if (shorten && i % 2 == 1) keep_n = s / 3 + (uint32_t)(rng() % (s - s / 3));
            t.rows.push_back(finish_row(v, keep_n));
        }
        t.groups.push_back({g0, (uint32_t)t.rows.size()});
    };
    for (int k = 0; k < 3; k++) single();
    if (kind == "near") {
        group(40, 1.06, 0.96, 0.03, 0, false);
        single();
        group(9, 1.06, 0.96, 0.03, 0, false);
    } else if (kind == "loose") {
        group(24, 1.3, 0.70, 0.30, 0, false);
        single();
        group(12, 1.3, 0.70, 0.30, 0, false);
    } else if (kind == "clumped") {
        group(20, 1.06, 0.90, 0.05, 20, false);
    } else if (kind == "clumped12") {
        group(20, 1.06, 0.90, 0.05, 12, false);
    } else if (kind == "clumped5") {
        group(20, 1.06, 0.90, 0.05, 5, false);
    } else if (kind == "short") {
        group(16, 1.06, 0.95, 0.03, 0, true);
        group(10, 1.3, 0.70, 0.30, 0, true);
    } else if (kind == "fuzz") {
        const uint32_t ng = 1 + (uint32_t)(rng() % 3);
        for (uint32_t g = 0; g < ng; g++) {
            const uint32_t m = 2 + (uint32_t)(rng() % (rng() % 4 == 0 ? 150 : 40));
            const double pool_factor = 1.0 + (double)(rng() % 250) / 100.0, keep = 0.3 + (double)(rng() % 69) / 100.0, priv = (double)(rng() % 40) / 100.0;
            const uint32_t clump = rng() % 3 == 0 ? (uint32_t)(rng() % 26) : 0u;
            group(m, pool_factor, keep, priv, clump, rng() % 3 == 0);
            if (rng() % 2) single();
        }
    } else if (kind == "wide") {
        group(140, 3.0, 0.33, 0.02, 0, false);       // more rows than a block of 128: two blocks, several column blocks
    }
    single();
    t.n = (uint32_t)t.rows.size();
    return t;
}

// the loop of compareSketches on two rows (CommandDistance.cpp:347-385)
static void reference_pair(const std::vector<uint64_t> &A, const std::vector<uint64_t> &B, uint32_t s, uint32_t &common, uint32_t &denom)
{
    size_t i = 0, j = 0;
    uint64_t c = 0, d = 0;
    while (d < s && i < A.size() && j < B.size()) {
        if (A[i] < B[j]) i++;
        else if (B[j] < A[i]) j++;
        else { i++; j++; c++; }
        d++;
    }
    if (d < s) {
        if (i < A.size()) d += A.size() - i;
        if (j < B.size()) d += B.size() - j;
        if (d > s) d = s;
    }
    common = (uint32_t)c;
    denom = (uint32_t)d;
}

template <uint32_t EK> static void run_encode(uint32_t n, size_t smem, const uint32_t *off, const uint32_t *code, uint32_t *pos, uint32_t rs,
                                              const uint32_t *grp_of, const DenseGroup *groups, const uint32_t *ulist, const uint32_t *upos,
                                              unsigned long long *gdata, uint32_t wmax, uint16_t *ext, uint32_t xs)
{
    hipLaunchKernelGGL(dn_encode_kernel<EK>, dim3((n + 63u) & ~63u), dim3(256), smem, nullptr, off, code, pos, rs, grp_of, groups, ulist, upos, gdata, wmax, ext,
                       xs, n, 1u);
}

template <uint32_t R, uint32_t IL> static void run_pairs(const std::vector<DenseTile> &tiles, size_t smem, const DenseGroup *groups,
                                                         const unsigned long long *gdata, uint32_t wmax,
                                                         uint32_t use_lists, const uint16_t *ext, uint32_t xs, uint32_t s, uint32_t n, uint2 *out)
{
    hipLaunchKernelGGL((dn_pairs_kernel<R, IL>), dim3((uint32_t)tiles.size()), dim3(128), smem, nullptr, tiles.data(), groups, gdata,
                       use_lists, ext, xs, s, 0u, n, (uint64_t)0, (const uint32_t *)nullptr, out, DenseList());
}

static int run_case(const std::string &name, uint32_t s, uint64_t seed, const std::string &kind)
{
    const Table t = make_table(s, seed, kind);
    const uint32_t n = t.n;
    const uint32_t rs = ((s + 3u) & ~3u) + 4u;
    // ---- the index: entries in value order, rows ascending inside a value
    std::vector<uint32_t> off(n + 1, 0);
    for (uint32_t r = 0; r < n; r++) off[r + 1] = off[r] + (uint32_t)t.rows[r].size();
    const uint32_t E = off[n];
    struct Ent { uint64_t v; uint32_t row, p; };
    std::vector<Ent> e;
    for (uint32_t r = 0; r < n; r++)
        for (uint32_t p = 0; p < t.rows[r].size(); p++) e.push_back({t.rows[r][p], r, p});
    std::stable_sort(e.begin(), e.end(), [](const Ent &a, const Ent &b) { return a.v < b.v; });
    std::vector<uint32_t> code((size_t)n * rs + 64, 0xFFFFFFFFu), pos((size_t)n * rs, 0), grp_of(n, 0xFFFFFFFFu);
    std::vector<DenseGroup> groups;
    for (auto &g : t.groups) {
        DenseGroup G{};
        G.g0 = g.first;
        G.g1 = g.second;
        for (uint32_t r = G.g0; r < G.g1; r++) grp_of[r] = (uint32_t)groups.size();
        groups.push_back(G);
    }
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> uni(groups.size());      // per group {run start, leader's position}
    for (uint32_t q = 0; q < E;) {
        uint32_t z = q;
        while (z < E && e[z].v == e[q].v) z++;
        for (uint32_t x = q; x < z; x++) {
            code[(size_t)e[x].row * rs + e[x].p] = (q << 1) | (z - q > 1 ? 1u : 0u);
            pos[(size_t)e[x].row * rs + e[x].p] = x;
        }
        for (uint32_t x = q; x < z;) {                      // holders inside one group stand side by side
            const uint32_t g = grp_of[e[x].row];
            uint32_t y = x + 1;
            while (y < z && grp_of[e[y].row] == g) y++;
            if (g != 0xFFFFFFFFu && y - x >= 2) uni[g].push_back({q, x});
            x = y;
        }
        q = z;
    }
    std::vector<uint32_t> ulist, upos;
    uint32_t xrows = 0, wmax = 0;
    uint64_t words = 0;
    for (size_t g = 0; g < groups.size(); g++) {
        DenseGroup &G = groups[g];
        G.ustart = (uint32_t)ulist.size();
        G.u = (uint32_t)uni[g].size();
        for (auto &x : uni[g]) { ulist.push_back(x.first); upos.push_back(x.second); }
        G.W = (G.u >> 6) + 1u;
        G.xrow0 = xrows;
        G.data_off = words;
        const uint64_t m = G.g1 - G.g0;
        xrows += (uint32_t)m;
        words += ((m + 127) / 128) * dense_block_words(G.W);
        wmax = std::max(wmax, G.W);
    }
    if (ulist.empty()) { ulist.push_back(0); upos.push_back(0); }
    const uint32_t xs = ((s + 7u) & ~7u) + 8u;
    const std::vector<uint32_t> pos0 = pos;
    int bad = 0;
    auto report = [&](const char *what, size_t i, unsigned long long got, unsigned long long want) {
        if (bad++ < 8) printf("  %s[%zu] = %llu, expected %llu\n", what, i, got, want);
    };
    char info[160];
    snprintf(info, sizeof info, "n %u s %u E %u groups %zu widest universe %u words", n, s, E, groups.size(), wmax);
    const size_t smem_enc = ((size_t)28 * wmax + 9) * 4 + (size_t)wmax * 64 * 4;
    const uint64_t npairs = (uint64_t)n * (n - 1) / 2;
    std::vector<unsigned long long> gdata_first;
    for (int variant = 0; variant < 2; variant++) {        // an entry per work-item; eight entries per work-item
        std::vector<unsigned long long> gdata(words + 1, 0xABABABABABABABABull);
        std::vector<uint16_t> ext((size_t)xrows * xs + 1, 0xEEEE);
        pos = pos0;
        if (variant == 0) run_encode<1>(n, smem_enc, off.data(), code.data(), pos.data(), rs, grp_of.data(), groups.data(), ulist.data(), upos.data(), gdata.data(), wmax, ext.data(), xs);
        else run_encode<8>(n, smem_enc, off.data(), code.data(), pos.data(), rs, grp_of.data(), groups.data(), ulist.data(), upos.data(), gdata.data(), wmax, ext.data(), xs);
        // the clipped runs: an entry whose value is in its group's universe ends at the leader's position
        for (uint32_t r = 0; r < n; r++) {
            const uint32_t g = grp_of[r];
            for (uint32_t p = 0; p < t.rows[r].size(); p++) {
                const size_t img = (size_t)r * rs + p;
                uint32_t want = pos0[img];
                if (g != 0xFFFFFFFFu) {
                    const uint32_t gs = code[img] >> 1;
                    auto &U = uni[g];
                    auto it = std::lower_bound(U.begin(), U.end(), std::make_pair(gs, 0u));
                    if (it != U.end() && it->first == gs) want = it->second;
                }
                if (pos[img] != want) report(variant ? "pos_img (8 per item)" : "pos_img", img, pos[img], want);
            }
        }
        if (variant == 0) gdata_first = gdata;
        else if (gdata != gdata_first) report("gdata of the two encode kernels differ", 0, 1, 0);
        // ---- the pairs: every tile shape, rows side by side, masks and lists
        for (uint32_t R : {8u, 32u}) {
            std::vector<DenseTile> tiles;
            for (uint32_t g = 0; g < groups.size(); g++) {
                const DenseGroup &G = groups[g];
                for (uint32_t row0 = G.g0; row0 < G.g1; row0 += R) {
                    const uint64_t a_lo = std::max<uint64_t>(row0, (uint64_t)G.g0 + 1), a_hi = std::min<uint64_t>(row0 + R, G.g1);
                    if (a_lo >= a_hi) continue;
                    const uint32_t cb_last = (uint32_t)((a_hi - 2 - G.g0) >> 7);
                    for (uint32_t cb = 0; cb <= cb_last; cb++) tiles.push_back({g, row0, cb});
                }
            }
            const size_t smem = dense_pairs_lds(wmax, R);
            for (uint32_t il : {4u, 8u, 16u}) {
                if (R == 8 && il == 16) continue;
                if (variant == 1 && il != 8) continue;     // (the second encode's data through one shape of the pairs kernel)
                for (uint32_t lists = 0; lists < 2; lists++) {
                    std::vector<uint2> out(npairs + 1, make_uint2(0xDEADu, 0xDEADu));
                    if (R == 8 && il == 4) run_pairs<8, 4>(tiles, smem, groups.data(), gdata.data(), wmax, lists, ext.data(), xs, s, n, out.data());
                    else if (R == 8) run_pairs<8, 8>(tiles, smem, groups.data(), gdata.data(), wmax, lists, ext.data(), xs, s, n, out.data());
                    else if (il == 4) run_pairs<32, 4>(tiles, smem, groups.data(), gdata.data(), wmax, lists, ext.data(), xs, s, n, out.data());
                    else if (il == 8) run_pairs<32, 8>(tiles, smem, groups.data(), gdata.data(), wmax, lists, ext.data(), xs, s, n, out.data());
                    else run_pairs<32, 16>(tiles, smem, groups.data(), gdata.data(), wmax, lists, ext.data(), xs, s, n, out.data());
                    char what[96];
                    snprintf(what, sizeof what, "pair (tile %u rows, %u side by side, %s)", R, il, lists ? "lists" : "planes");
                    for (uint32_t a = 1; a < n; a++)
                        for (uint32_t b = 0; b < a; b++) {
                            const uint2 got = out[(uint64_t)a * (a - 1) / 2 + b];
                            const bool inside = grp_of[a] != 0xFFFFFFFFu && grp_of[a] == grp_of[b];
                            if (!inside) {
                                if (got.x != 0xDEADu) report("a pair outside the groups was written", (size_t)a * n + b, got.x, 0xDEAD);
                                continue;
                            }
                            uint32_t c = 0, d = 0;
                            reference_pair(t.rows[a], t.rows[b], s, c, d);
                            if (got.x != c || got.y != d) report(what, (size_t)a * n + b, ((unsigned long long)got.x << 32) | got.y, ((unsigned long long)c << 32) | d);
                        }
                }
            }
        }
    }
    printf("%-12s %s: %s\n", name.c_str(), info, bad ? "MISMATCH" : "ok");
    return bad ? 1 : 0;
}

int main(int argc, char **argv)
{
    struct Case { const char *name; uint32_t s; uint64_t seed; const char *kind; };
    const Case cases[] = {
        {"near", 150, 1, "near"},          // near-copies: a few extras per row
        {"loose", 150, 2, "loose"},        // loose clusters: dozens of extras per word -- the bit planes
        {"clumped", 150, 3, "clumped"},    // more than fifteen extras in one gap: flagged words, the rows' lists
        {"clumped12", 150, 7, "clumped12"},// twelve extras in one gap: all four bit planes
        {"clumped5", 150, 8, "clumped5"},  // five: planes 0 and 2
        {"short", 150, 4, "short"},        // rows shorter than s: unions that end before s
        {"wide", 64, 5, "wide"},           // universe three times s, 140 rows: two blocks of rows, the loop's early exit
        {"one_word", 40, 6, "near"},       // universes of one word
    };
    int rc = 0, ran = 0;
    if (argc > 1 && std::string(argv[1]) == "fuzz") {      // dense_emu fuzz <seed> <cases>: groups of random shape
        std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 1);
        const int count = argc > 3 ? atoi(argv[3]) : 30;
        for (int i = 0; i < count; i++) {
            const uint32_t s2 = 8 + (uint32_t)(rng() % (rng() % 3 == 0 ? 250 : 90));
            char name[48];
            snprintf(name, sizeof name, "fuzz %d", i);
            rc |= run_case(name, s2, rng(), "fuzz");
        }
        printf(rc ? "FAILED\n" : "all cases agree\n");
        return rc;
    }
    for (const Case &c : cases) {
        bool want = argc < 2;
        for (int i = 1; i < argc; i++) want = want || std::string(argv[i]) == c.name || std::string(argv[i]) == "all";
        if (!want) continue;
        rc |= run_case(c.name, c.s, c.seed, c.kind);
        ran++;
    }
    if (!ran) { printf("no such case\n"); return 2; }
    printf(rc ? "FAILED\n" : "all cases agree\n");
    return rc;
}
