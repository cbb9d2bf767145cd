// join_emu_main.cpp -- runs the join engine of mash_amd/csrc/compare_join.hip (the list kernels jn_emit / jn_heads / jn_groups /
// jn_gend / jn_levels, the count of shared hashes, and jn_tile_kernel) on host threads (tools/hipemu) and compares EVERY pair
// with the loop of compareSketches (CommandDistance.cpp:347-385, restated below on the rows' hashes).  The code image the
// kernels read is made here by a std::stable_sort, as index_emu_main.cpp states the index.
// TEST INFRASTRUCTURE (tests/test_join_emu.py); built with g++.
//
//   join_emu <case> [seed]      cases: see main(); exit status 0 = every pair agrees
#include "../../tools/hipemu/hipemu.h"

#include <algorithm>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "../../mash_amd/csrc/compare_join.hip"

using namespace mg;

typedef std::vector<std::vector<uint64_t>> Rows;

static std::vector<uint64_t> finish_row(std::vector<uint64_t> v, uint32_t keep)
{
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    if (v.size() > keep) v.resize(keep);
    return v;
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
        d += (A.size() - i) + (B.size() - j);
        if (d > s) d = s;
    }
    common = (uint32_t)c;
    denom = (uint32_t)d;
}

// a tree of descent (workloads/synth.py: species_sketches in small): slot k of a row holds a value of stratum k; on every
// edge a share of the slots is replaced
static Rows species(uint32_t n, uint32_t s, std::mt19937_64 &rng, double q_inner, double q_leaf, bool ragged)
{
    uint32_t L = 1;
    while ((1u << L) < n) L++;
    Rows rows;
    for (uint32_t i = 0; i < n; i++) {
        std::vector<uint64_t> v(s);
        for (uint32_t k = 0; k < s; k++) {
            uint64_t origin = 0;
            for (uint32_t lev = 1; lev <= L; lev++) {
                const uint64_t a = i >> (L - lev);
                std::mt19937_64 h(((uint64_t)k * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)lev << 50) ^ (a * 0xC2B2AE3D27D4EB4Full));
                const double u = (double)(h() >> 11) / 9007199254740992.0;
                if (u < (lev == L ? q_leaf : q_inner)) origin = ((uint64_t)lev << 40) | a;
            }
            std::mt19937_64 h2(((uint64_t)k * 0xC2B2AE3D27D4EB4Full) ^ (origin * 0x9E3779B97F4A7C15ull));
            v[k] = ((uint64_t)k << 40) + (h2() & ((1ull << 40) - 1));
        }
        uint32_t keep = s;
        if (ragged && rng() % 3 == 0) keep = (uint32_t)(rng() % (s + 1));
        rows.push_back(finish_row(v, keep));
    }
    std::shuffle(rows.begin(), rows.end(), rng);
    return rows;
}

static Rows pool_rows(uint32_t n, uint32_t s, std::mt19937_64 &rng, double pool_factor, double keep, bool ragged)
{
    std::vector<uint64_t> pool;
    for (uint32_t k = 0; k < (uint32_t)(pool_factor * s) + 4; k++) pool.push_back(rng() >> 10);
    Rows rows;
    for (uint32_t i = 0; i < n; i++) {
        std::vector<uint64_t> v;
        for (uint64_t x : pool)
            if ((double)(rng() % 10000) < keep * 10000.0) v.push_back(x);
        for (uint32_t k = 0; k < s / 10 + 1; k++) v.push_back(rng() >> 10);
        uint32_t kn = s;
        if (ragged && rng() % 2) kn = (uint32_t)(rng() % (s + 1));
        rows.push_back(finish_row(v, kn));
    }
    return rows;
}

// the code image of an index over `rows` (copies of an earlier row stay out: rep), entries as the index states them: code =
// 2 x (first sorted position of the value's group) + (another row of the index holds it too)
struct Index {
    uint32_t n = 0, rs = 0, E = 0;
    std::vector<uint32_t> off, code, rep, gend, sorted_rows;
    std::vector<uint64_t> keys_sorted;
    bool copies = false;
};

static Index build_index(const Rows &rows, uint32_t s, bool dedup)
{
    Index ix;
    ix.n = (uint32_t)rows.size();
    ix.rs = (s + 3u) / 4u * 4u + 4u;
    ix.rep.resize(ix.n);
    std::map<std::vector<uint64_t>, uint32_t> firsts;
    for (uint32_t i = 0; i < ix.n; i++) {
        ix.rep[i] = i;
        if (dedup && !rows[i].empty()) {
            auto it = firsts.find(rows[i]);
            if (it == firsts.end()) firsts[rows[i]] = i;
            else { ix.rep[i] = it->second; ix.copies = true; }
        }
    }
    ix.off.assign(ix.n + 1, 0);
    std::vector<std::pair<uint64_t, uint32_t>> ent;         // (value, image index)
    for (uint32_t i = 0; i < ix.n; i++) {
        ix.off[i] = (uint32_t)ent.size();
        if (ix.rep[i] == i)
            for (uint32_t p = 0; p < rows[i].size(); p++) ent.push_back({rows[i][p], i * ix.rs + p});
    }
    ix.off[ix.n] = ix.E = (uint32_t)ent.size();
    std::stable_sort(ent.begin(), ent.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
    ix.code.assign((size_t)ix.n * ix.rs + 64, 0xFFFFFFFFu);
    ix.keys_sorted.resize(ix.E);
    ix.gend.assign(ix.E, 0xFFFFFFFFu);                      // (defined at the first position of a value's group only)
    ix.sorted_rows.resize(ix.E);
    for (uint32_t e = 0; e < ix.E;) {
        uint32_t f = e;
        while (f < ix.E && ent[f].first == ent[e].first) f++;
        ix.gend[e] = f;
        for (uint32_t g = e; g < f; g++) {
            ix.code[ent[g].second] = 2u * e + (f - e >= 2 ? 1u : 0u);
            ix.keys_sorted[g] = ent[g].first;
            ix.sorted_rows[g] = ent[g].second / ix.rs;
        }
        e = f;
    }
    return ix;
}

struct Lists {
    std::vector<unsigned long long> key_a, key_b;
    std::vector<uint32_t> val_a, val_b, goff, gend, thr;
    std::vector<uint2> grp;
    JoinSide side;
};

static void make_lists(const uint32_t *img, uint32_t rs, const uint32_t *off, const uint32_t *rep, uint32_t nrows, uint32_t s, uint32_t E, bool only_shared, Lists &L)
{
    const uint64_t slots = (uint64_t)nrows * s;
    const uint32_t nb = (nrows + 63u) / 64u;
    L.key_a.resize(slots); L.key_b.resize(slots); L.val_a.resize(slots); L.val_b.resize(slots);
    L.grp.resize(slots + 1); L.goff.assign(nb + 1, 0); L.gend.assign(nb + 1, 0); L.thr.assign((size_t)nb * 16, 0);
    const uint32_t *ent = nullptr;
    unsigned char temp[16];
    const hipError_t e = join_build_lists(img, rs, off, rep, nrows, s, E, only_shared, temp, 16, L.key_a.data(), L.key_b.data(), L.val_a.data(), L.val_b.data(),
                                          L.grp.data(), L.goff.data(), L.gend.data(), L.thr.data(), &ent, nullptr);
    if (e != hipSuccess) { fprintf(stderr, "join_build_lists failed\n"); exit(2); }
    L.side.grp = L.grp.data(); L.side.ent = ent; L.side.goff = L.goff.data(); L.side.gend = L.gend.data(); L.side.thr = L.thr.data();
}

static uint64_t g_pairs = 0, g_shared = 0;

// triangle over rows [rb, re) of the table; perm: the index is built on the table in another order (index row a = table row inv[a])
static int run_triangle(const Rows &table, uint32_t s, uint32_t rb, uint32_t re, bool permute, bool early, std::mt19937_64 &rng, const char *what,
                        bool order = false)
{
    const uint32_t n = (uint32_t)table.size();
    std::vector<uint32_t> inv(n);
    for (uint32_t i = 0; i < n; i++) inv[i] = i;
    if (permute) { std::shuffle(inv.begin(), inv.end(), rng); rb = 0; re = n; }
    Rows rows(n);
    for (uint32_t a = 0; a < n; a++) rows[a] = table[inv[a]];
    Index ix = build_index(rows, s, true);
    if (ix.E == 0) return 0;
    Lists L;
    const uint32_t *src = ix.copies ? ix.rep.data() : nullptr, *map = permute ? inv.data() : nullptr;
    std::vector<uint32_t> o_perm(n), o_src(n), o_map(n);
    // the lists on the rows in the order of their families: the whole triangle, or a job over the table's LAST rows [rb, n), which
    // then form a segment of their own behind the others (not together with a permuted index: the range is one of index rows)
    uint32_t split = 0;
    if (order) {
        if (!permute && re == n) split = rb; else { rb = 0; re = n; }
        std::vector<uint32_t> lab(6ull * n), val_a(n);
        std::vector<unsigned long long> key_a(n), key_b(n);
        unsigned char temp[16];
        if (join_order_rows(ix.code.data(), ix.rs, ix.off.data(), src, map, ix.gend.data(), ix.sorted_rows.data(), n, temp, 16, lab.data(), key_a.data(),
                            key_b.data(), val_a.data(), o_perm.data(), o_src.data(), o_map.data(), nullptr, split) != hipSuccess) { fprintf(stderr, "join_order_rows failed\n"); return 1; }
        std::vector<uint32_t> seen(n, 0);
        for (uint32_t x : o_perm) seen[x]++;
        for (uint32_t x : seen) if (x != 1) { fprintf(stderr, "%s: the order is no permutation\n", what); return 1; }
        for (uint32_t a2 = 0; a2 < n; a2++)
            if ((o_perm[a2] >= split) != (a2 >= split)) { fprintf(stderr, "%s: the rows from %u on are not the order's last\n", what, split); return 1; }
        src = o_src.data();
        map = o_map.data();
    }
    make_lists(ix.code.data(), ix.rs, ix.off.data(), src, n, s, ix.E, !ix.copies, L);
    JoinArgs a;
    a.rows = a.cols = L.side;
    if (!early) a.rows.thr = a.cols.thr = nullptr;
    a.row_cnt_off = a.col_cnt_off = ix.off.data();
    a.rep = a.col_rep = src;
    a.inv = map;
    permute = permute || order;
    const uint64_t out_base = rb ? (uint64_t)rb * (rb - 1) / 2 : 0, npairs = (uint64_t)re * (re - 1) / 2 - out_base;
    std::vector<uint2> out(npairs + 1, make_uint2(0xDEADBEEFu, 0xDEADBEEFu));
    a.out = out.data();
    a.out_base = out_base;
    a.ncols = n;
    a.row_begin = rb;                                       // (a permuted index serves the whole triangle only: rb = 0, re = n there)
    a.row_end = re;
    a.bi0 = a.row_begin / 64u;
    const uint64_t bi1 = ((uint64_t)a.row_end + 63u) / 64u;
    a.ncb = (n + 63u) / 64u;
    a.triangle = 1;
    a.s = s;
    a.ntiles = bi1 * (bi1 + 1) / 2 - (uint64_t)a.bi0 * (a.bi0 + 1) / 2;
    if (launch_join_tiles(a, nullptr) != hipSuccess) { fprintf(stderr, "launch failed\n"); return 1; }
    int bad = 0;
    for (uint32_t i = rb; i < re; i++)
        for (uint32_t j = 0; j < i; j++) {
            uint32_t c, d;
            reference_pair(table[i], table[j], s, c, d);
            const uint2 got = out[(uint64_t)i * (i - 1) / 2 + j - out_base];
            g_pairs++;
            g_shared += c;
            if (got.x != c || got.y != d) {
                if (bad++ < 8) fprintf(stderr, "%s: pair (%u, %u): got {%u, %u}, reference {%u, %u}\n", what, i, j, got.x, got.y, c, d);
            }
        }
    if (out[npairs].x != 0xDEADBEEFu) { fprintf(stderr, "%s: wrote past the end\n", what); bad++; }
    return bad;
}

// rect: queries located in the reference table's sorted values (sp_locate_kernel's scheme: 2 lo + 1 where found, else 2 lo)
static int run_rect(const Rows &ref, const Rows &qry, uint32_t s, bool early, const char *what)
{
    Index ix = build_index(ref, s, true);                 // (copies among the reference rows stay out of the index)
    if (ix.E == 0) return 0;
    const uint32_t nq = (uint32_t)qry.size(), nr = (uint32_t)ref.size();
    std::vector<uint32_t> qoff(nq + 1, 0), qimg((size_t)nq * ix.rs + 64, 0xFFFFFFFFu);
    for (uint32_t q = 0; q < nq; q++) {
        qoff[q + 1] = qoff[q] + (uint32_t)qry[q].size();
        for (uint32_t p = 0; p < qry[q].size(); p++) {
            const uint32_t lo = (uint32_t)(std::lower_bound(ix.keys_sorted.begin(), ix.keys_sorted.end(), qry[q][p]) - ix.keys_sorted.begin());
            const bool found = lo < ix.E && ix.keys_sorted[lo] == qry[q][p];
            qimg[(size_t)q * ix.rs + p] = 2u * lo + (found ? 1u : 0u);
        }
    }
    Lists LC, LQ;
    make_lists(ix.code.data(), ix.rs, ix.off.data(), ix.copies ? ix.rep.data() : nullptr, nr, s, ix.E, false, LC);
    make_lists(qimg.data(), ix.rs, qoff.data(), nullptr, nq, s, ix.E, true, LQ);
    JoinArgs a;
    a.rows = LQ.side;
    a.cols = LC.side;
    if (!early) a.rows.thr = a.cols.thr = nullptr;
    a.row_cnt_off = qoff.data();
    a.col_cnt_off = ix.off.data();
    a.rep = nullptr;
    a.col_rep = ix.copies ? ix.rep.data() : nullptr;
    a.inv = nullptr;
    std::vector<uint2> out((size_t)nq * nr + 1, make_uint2(0xDEADBEEFu, 0xDEADBEEFu));
    a.out = out.data();
    a.out_base = 0;
    a.ncols = nr;
    a.row_begin = 0;
    a.row_end = nq;
    a.bi0 = 0;
    a.ncb = (nr + 63u) / 64u;
    a.triangle = 0;
    a.s = s;
    a.ntiles = (uint64_t)((nq + 63u) / 64u) * a.ncb;
    if (launch_join_tiles(a, nullptr) != hipSuccess) { fprintf(stderr, "launch failed\n"); return 1; }
    int bad = 0;
    for (uint32_t q = 0; q < nq; q++)
        for (uint32_t r = 0; r < nr; r++) {
            uint32_t c, d;
            reference_pair(qry[q], ref[r], s, c, d);          // (rect: the shorter sketch size is s already)
            const uint2 got = out[(size_t)q * nr + r];
            g_pairs++;
            g_shared += c;
            if (got.x != c || got.y != d) {
                if (bad++ < 8) fprintf(stderr, "%s: pair (q %u, r %u): got {%u, %u}, reference {%u, %u}\n", what, q, r, got.x, got.y, c, d);
            }
        }
    if (out[(size_t)nq * nr].x != 0xDEADBEEFu) { fprintf(stderr, "%s: wrote past the end\n", what); bad++; }
    return bad;
}

// the count of shared hashes against its definition (pairs x common values of the WHOLE rows, rows below only)
static int check_shared(const Rows &rows, uint32_t s)
{
    Index ix = build_index(rows, s, false);
    if (ix.E == 0) return 0;
    // position image: an entry's own sorted position; lo = code >> 1
    std::vector<uint32_t> pos((size_t)ix.n * ix.rs, 0);
    {
        std::vector<std::pair<uint64_t, uint32_t>> ent;
        for (uint32_t i = 0; i < ix.n; i++)
            for (uint32_t p = 0; p < rows[i].size(); p++) ent.push_back({rows[i][p], i * ix.rs + p});
        std::stable_sort(ent.begin(), ent.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
        for (uint32_t e = 0; e < ent.size(); e++) pos[ent[e].second] = e;
    }
    unsigned long long sum = 0, want = 0;
    launch_join_shared(ix.code.data(), pos.data(), 1, ix.rs, ix.off.data(), 0, ix.n, &sum, nullptr);
    for (uint32_t i = 0; i < ix.n; i++)
        for (uint32_t j = 0; j < i; j++) {
            std::vector<uint64_t> both;
            std::set_intersection(rows[i].begin(), rows[i].end(), rows[j].begin(), rows[j].end(), std::back_inserter(both));
            want += both.size();
        }
    if (sum != want) { fprintf(stderr, "shared hashes: kernel %llu, definition %llu\n", sum, want); return 1; }
    return 0;
}

int main(int argc, char **argv)
{
    const std::string which = argc > 1 ? argv[1] : "species";
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    std::mt19937_64 rng(seed * 0x9E3779B97F4A7C15ull + 7);
    int bad = 0;
    if (which == "species") {
        // 150 rows = three blocks (the last one ragged), every pair shares a tenth to a half; with and without the early stop
        Rows t = species(150, 96, rng, 0.06, 0.2, false);
        bad += run_triangle(t, 96, 0, 150, false, true, rng, "species");
        bad += run_triangle(t, 96, 0, 150, false, false, rng, "species, no early stop");
        bad += run_triangle(t, 96, 0, 150, false, true, rng, "species, rows by family", true);
        bad += run_triangle(t, 96, 0, 150, true, true, rng, "species, rows by family over a permuted index", true);
        bad += check_shared(t, 96);
    } else if (which == "ranges") {
        // row ranges that start and end inside blocks; the permuted index
        Rows t = species(200, 64, rng, 0.05, 0.15, false);
        bad += run_triangle(t, 64, 70, 131, false, true, rng, "rows [70, 131)");
        bad += run_triangle(t, 64, 128, 200, false, true, rng, "rows [128, 200)");
        bad += run_triangle(t, 64, 0, 65, false, true, rng, "rows [0, 65)");
        bad += run_triangle(t, 64, 0, 200, true, true, rng, "permuted index");
        bad += run_triangle(t, 64, 70, 200, false, true, rng, "rows [70, 200), rows by family in two segments", true);
        bad += run_triangle(t, 64, 128, 200, false, true, rng, "rows [128, 200), rows by family in two segments", true);
    } else if (which == "ragged") {
        // rows of every length (empty ones too) over one pool and over a tree: positions differ between the rows of a pair
        Rows t = pool_rows(90, 50, rng, 2.5, 0.4, true);
        bad += run_triangle(t, 50, 0, 90, false, true, rng, "ragged pool");
        Rows u = species(100, 80, rng, 0.05, 0.2, true);
        bad += run_triangle(u, 80, 0, 100, false, true, rng, "ragged species");
    } else if (which == "copies") {
        // identical rows (kept out of the index: their representative's entries speak for them), among them short ones
        Rows t = pool_rows(80, 40, rng, 1.5, 0.7, true);
        for (uint32_t k = 0; k < 12; k++) t[(size_t)(rng() % 80)] = t[(size_t)(rng() % 80)];
        t[70] = t[3]; t[71] = t[3]; t[5] = t[3];
        bad += run_triangle(t, 40, 0, 80, false, true, rng, "copies");
        bad += run_triangle(t, 40, 0, 80, true, true, rng, "copies, permuted");
        bad += run_triangle(t, 40, 0, 80, false, true, rng, "copies, rows by family", true);
    } else if (which == "near") {
        // near-copies: every value held by nearly every row (holder lists of 64 on both sides), s reached long before the lists end
        Rows t = pool_rows(130, 64, rng, 1.05, 0.95, false);
        bad += run_triangle(t, 64, 0, 130, false, true, rng, "near-copies");
        bad += run_triangle(t, 64, 0, 130, false, false, rng, "near-copies, no early stop");
    } else if (which == "random") {
        Rows t;
        for (uint32_t i = 0; i < 70; i++) {
            std::vector<uint64_t> v;
            for (uint32_t k = 0; k < 40; k++) v.push_back(rng() >> 10);
            t.push_back(finish_row(v, 32));
        }
        bad += run_triangle(t, 32, 0, 70, false, true, rng, "random");
        if (g_shared != 0) { fprintf(stderr, "random rows share values?\n"); bad++; }
    } else if (which == "rect") {
        Rows ref = species(140, 64, rng, 0.05, 0.2, false);
        ref[100] = ref[30]; ref[101] = ref[30]; ref[7] = ref[139];      // copies among the reference rows
        Rows qry(ref.begin() + 20, ref.begin() + 95);       // 75 queries: some rows of the table itself
        Rows more = pool_rows(10, 64, rng, 1.5, 0.6, true); // and strangers of every length
        qry.insert(qry.end(), more.begin(), more.end());
        bad += run_rect(ref, qry, 64, true, "rect");
        bad += run_rect(ref, qry, 64, false, "rect, no early stop");
    } else if (which == "fuzz") {
        const int cases = argc > 3 ? atoi(argv[3]) : 10;
        for (int c = 0; c < cases; c++) {
            const uint32_t n = 2 + (uint32_t)(rng() % 200), s = 1 + (uint32_t)(rng() % 120);
            const int kind = (int)(rng() % 3);
            Rows t = kind == 0 ? species(n, s, rng, 0.01 + (double)(rng() % 20) / 100.0, (double)(rng() % 40) / 100.0, rng() % 2)
                               : pool_rows(n, s, rng, 1.0 + (double)(rng() % 300) / 100.0, 0.2 + (double)(rng() % 79) / 100.0, rng() % 2);
            if (kind == 2) for (uint32_t k = 0; k < n / 8; k++) t[(size_t)(rng() % n)] = t[(size_t)(rng() % n)];
            uint32_t rb = 0, re = n;
            if (rng() % 2) { rb = (uint32_t)(rng() % n); re = rb + 1 + (uint32_t)(rng() % (n - rb)); }
            char what[96];
            snprintf(what, sizeof what, "fuzz %d (n %u, s %u, kind %d, rows [%u, %u))", c, n, s, kind, rb, re);
            if (rng() % 3 == 0) re = n;
            bad += run_triangle(t, s, rb, re, rb == 0 && re == n && rng() % 2, rng() % 4 != 0, rng, what, re == n && rng() % 2);
            if (bad) break;
        }
    } else {
        fprintf(stderr, "unknown case %s\n", which.c_str());
        return 2;
    }
    printf("%s: %llu pairs, %llu shared hashes counted by the reference, %d differ\n", which.c_str(), (unsigned long long)g_pairs, (unsigned long long)g_shared, bad);
    if (bad == 0) printf("all cases agree\n");
    return bad ? 1 : 0;
}
