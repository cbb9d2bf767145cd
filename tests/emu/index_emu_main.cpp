// index_emu_main.cpp -- runs the index-build kernels of mash_amd/csrc/index_build.hip on host threads (tools/hipemu) and
// compares every array they produce with a plain statement of the index: all (value, row) entries in a std::stable_sort by
// value (rows ascend inside a value because the entries are generated row-major).  TEST INFRASTRUCTURE (tests/test_index_emu.py);
// built with g++, optionally with -fsanitize=thread: a missing barrier in a kernel is then a reported data race.
//
//   index_emu <case> ...      cases: see main(); exit status 0 = every case agrees
#include "../../tools/hipemu/hipemu.h"

#include <algorithm>
#include <random>
#include <string>

#include "../../mash_amd/csrc/index_build.hip"

using namespace mg;

struct Table {
    uint32_t n = 0, s = 0;
    uint64_t stride = 0;
    std::vector<uint64_t> H;          // n x stride, ascending rows, padding ~0
    std::vector<uint32_t> cnt;        // entries of every row that enter the index
};

static Table make_table(uint32_t n, uint32_t s, uint64_t seed, const std::string &kind)
{
    Table t;
    t.n = n;
    t.s = s;
    t.stride = s;
    t.H.assign((size_t)n * s, ~0ull);
    t.cnt.assign(n, 0);
    std::mt19937_64 rng(seed);
    std::vector<uint64_t> pool;
    for (uint32_t r = 0; r < n; r++) {
        uint32_t c = s;
        uint64_t top = 1ull << 54;                       // hashes of a 1 Mbp genome at s = 1000 reach about here
        if (kind == "ragged") {
            c = (uint32_t)(rng() % (s + 1u));            // short and empty rows
            if (r % 7 == 3) c = 0;
        } else if (kind == "sizes") {
            top = 1ull << (50 + rng() % 10);             // genomes of many sizes: rows of very different density
        } else if (kind == "top") {
            top = ~0ull;                                 // values up to the top bit
        }
        std::vector<uint64_t> v;
        if (kind == "clusters" || kind == "clade" || kind == "twins") {
            // rows of a cluster draw most of their values from a shared pool: values held by several rows
            const uint32_t per = kind == "clusters" ? 8 : n;
            if (r % per == 0) {
                pool.clear();
                for (uint32_t k = 0; k < s + s / 4; k++) pool.push_back(rng() % top);
                if (kind == "twins")                         // pairs of neighbouring values: no number of leading bits tells them apart
                    for (uint32_t k = 1; k < pool.size(); k += 2) pool[k] = pool[k - 1] + 1;
            }
            std::vector<uint64_t> p2 = pool;
            std::shuffle(p2.begin(), p2.end(), rng);
            for (uint32_t k = 0; k < c; k++) v.push_back(k % 10 == 9 ? rng() % top : p2[k]);
        } else {
            for (uint32_t k = 0; k < c; k++) v.push_back(rng() % top);
        }
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        t.cnt[r] = (uint32_t)v.size();
        std::copy(v.begin(), v.end(), t.H.begin() + (size_t)r * s);
    }
    if (kind == "copies")                                 // (a copy of an earlier row stays out of the index: count 0)
        for (uint32_t r = 1; r < n; r += 3) t.cnt[r] = 0;
    return t;
}

struct Ref {
    std::vector<uint64_t> keys;
    std::vector<uint32_t> rows, gend, gs, code, pos;
    unsigned long long inc = 0;
    uint32_t max_group = 0, groups = 0;
};

static Ref reference(const Table &t, const std::vector<uint32_t> &off, uint32_t rs)
{
    Ref R;
    const uint32_t E = off[t.n];
    std::vector<std::pair<uint64_t, uint32_t>> e;          // {value, image index}
    e.reserve(E);
    for (uint32_t r = 0; r < t.n; r++)
        for (uint32_t p = 0; p < off[r + 1] - off[r]; p++) e.push_back({t.H[(size_t)r * t.stride + p], r * rs + p});
    std::stable_sort(e.begin(), e.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    R.keys.resize(E);
    R.rows.resize(E);
    R.gend.assign(E, 0);
    R.gs.resize(E);
    R.code.assign((size_t)t.n * rs, 0xFFFFFFFFu);
    R.pos.assign((size_t)t.n * rs, 0);
    for (uint32_t q = 0; q < E;) {
        uint32_t z = q;
        while (z < E && e[z].first == e[q].first) z++;
        R.gend[q] = z;
        R.groups++;
        R.max_group = std::max(R.max_group, z - q);
        for (uint32_t x = q; x < z; x++) {
            R.keys[x] = e[x].first;
            R.rows[x] = e[x].second / rs;
            R.gs[x] = q;
            R.code[e[x].second] = (q << 1) | (z - q > 1 ? 1u : 0u);
            R.pos[e[x].second] = x;
            R.inc += x - q;
        }
        q = z;
    }
    if (R.max_group < 2) R.max_group = 0;                 // (as the round-4 kernels report it: the largest group of a SHARED value)
    return R;
}

static int run_case(const std::string &name, uint32_t n, uint32_t s, uint64_t seed, const std::string &kind, bool expect_fallback = false,
                    double dens_scale = 1.0)
{
    const Table t = make_table(n, s, seed, kind);
    std::vector<uint32_t> off(n + 1, 0);
    uint64_t maxv = 0;
    double dens0 = 0;
    for (uint32_t r = 0; r < n; r++) {
        off[r + 1] = off[r] + t.cnt[r];
        if (t.cnt[r]) {
            const uint64_t last = t.H[(size_t)r * t.stride + t.cnt[r] - 1];
            maxv = std::max(maxv, last);
            dens0 += (double)t.cnt[r] / ((double)last + 1.0);
        }
    }
    const uint32_t E = off[n];
    const uint32_t rs = ((s + 3u) & ~3u) + 4u;
    if (E == 0) { printf("%-28s skipped (no entries)\n", name.c_str()); return 0; }
    const IxPlan plan = index_plan(n, E, s, rs, t.stride, maxv, dens0 * dens_scale, true);
    if (!plan.ok) {
        printf("%-28s plan refused: %s%s\n", name.c_str(), plan.why, expect_fallback ? " (expected)" : "");
        return expect_fallback ? 0 : 1;
    }
    const IxGeom &g = plan.g;
    std::vector<unsigned char> lb(plan.lb_bytes + 16, 0xAB), cnt(plan.cnt_bytes + 16, 0xAB), start(plan.start_bytes + 16, 0xAB), pk(plan.pk_bytes + 16, 0xAB),
        tc(plan.tc_bytes + 16, 0xAB), big(plan.big_bytes + 16, 0xAB), stat(index_stat_scratch_bytes());
    std::vector<uint64_t> keys(E, 0xDEADull);
    std::vector<uint32_t> rows(E, 0xDEADu), gend(E, 0xDEADu), gs(E, 0xDEADu), code((size_t)n * rs + 64, 0xDEADu), pos((size_t)n * rs, 0xDEADu);
    unsigned long long inc = 0;
    uint32_t max_group = 0, groups = 0, flags[4] = {0, 0, 0, 0};
    // groups of rows for the leader search (any grouping is a valid input): runs of 8 rows over the first two thirds of the table
    std::vector<uint32_t> grp_of(n, 0xFFFFFFFFu), groups32;
    for (uint32_t r0 = 0; r0 + 8 <= n * 2 / 3; r0 += 8) {
        const uint32_t gi = (uint32_t)(groups32.size() / 8);
        for (uint32_t r = r0; r < r0 + 8; r++) grp_of[r] = gi;
        groups32.insert(groups32.end(), {r0, r0 + 8, 0, 0, 0, 0, 0, 0});
    }
    if (groups32.empty()) groups32.assign(8, 0);
    IxLeaders lead;
    const uint32_t nsub = 1024, cap_sub = E / 2 + 64;
    std::vector<unsigned long long> lkey((size_t)nsub * cap_sub, ~0ull);
    std::vector<uint32_t> lval((size_t)nsub * cap_sub, ~0u), lcnt(nsub, 0);
    std::vector<uint32_t> lead_rows(4 * (size_t)n, 0);
    for (uint32_t r = 0; r < n; r++) {
        lead_rows[4 * r] = grp_of[r];
        if (grp_of[r] != 0xFFFFFFFFu) { lead_rows[4 * r + 1] = groups32[8 * grp_of[r]]; lead_rows[4 * r + 2] = groups32[8 * grp_of[r] + 1]; }
    }
    lead.grp_of = lead_rows.data();
    lead.key = lkey.data();
    lead.val = lval.data();
    lead.cnt = lcnt.data();
    lead.cap_sub = cap_sub;
    lead.nsub = nsub;
    const hipError_t e = index_build(plan, t.H.data(), off.data(), lb.data(), cnt.data(), start.data(), big.data(), pk.data(), tc.data(), keys.data(), rows.data(),
                                     gend.data(), gs.data(), code.data(), pos.data(), stat.data(), &inc, &max_group, &groups, flags, &lead, nullptr);
    if (e != hipSuccess) { printf("%-28s index_build failed\n", name.c_str()); return 1; }
    char geom[200];
    snprintf(geom, sizeof geom, "n %u s %u E %u shift %u B %u BW %u NW %u passes %u fullest %u big %u streamed %u", n, s, E, g.shift, g.Bp, g.BW, g.NW,
             g.npass, flags[IXF_MAXBUCKET], flags[IXF_NBIG], flags[IXF_NSTREAMED]);
    if (flags[IXF_DEGENERATE]) {
        printf("%-28s %s: flagged degenerate%s\n", name.c_str(), geom, expect_fallback ? " (expected)" : " UNEXPECTED");
        return expect_fallback ? 0 : 1;
    }
    if (expect_fallback) { printf("%-28s %s: expected a flag, none raised\n", name.c_str(), geom); return 1; }
    const Ref R = reference(t, off, rs);
    int bad = 0;
    auto report = [&](const char *what, size_t i, unsigned long long got, unsigned long long want) {
        if (bad++ < 8) printf("  %s[%zu] = %llu, expected %llu\n", what, i, got, want);
    };
    for (uint32_t q = 0; q < E; q++) {
        if (keys[q] != R.keys[q]) report("keys_sorted", q, keys[q], R.keys[q]);
        if (rows[q] != R.rows[q]) report("sorted_rows", q, rows[q], R.rows[q]);
        if (gs[q] != R.gs[q]) report("gs_of", q, gs[q], R.gs[q]);
        if (R.gs[q] == q && gend[q] != R.gend[q]) report("gend", q, gend[q], R.gend[q]);
    }
    for (size_t i = 0; i < (size_t)n * rs; i++) {
        if (code[i] != R.code[i]) report("code_img", i, code[i], R.code[i]);
        if (R.code[i] != 0xFFFFFFFFu && pos[i] != R.pos[i]) report("pos_img", i, pos[i], R.pos[i]);
    }
    {   // the leaders: a value's first holder inside a group of rows, if a second one follows (dn_leaders_kernel's statement)
        std::vector<std::pair<unsigned long long, uint32_t>> want, got;
        for (uint32_t q = 0; q < E; q++) {
            const uint32_t gi = grp_of[R.rows[q]];
            if (gi == 0xFFFFFFFFu) continue;
            const uint32_t g0 = groups32[8 * gi], g1 = groups32[8 * gi + 1], gsq = R.gs[q];
            const bool first = q == gsq || R.rows[q - 1] < g0;
            const bool more = q + 1 < R.gend[gsq] && R.rows[q + 1] < g1;
            if (first && more) want.push_back({((unsigned long long)gi << 32) | gsq, q});
        }
        for (uint32_t sub = 0; sub < nsub; sub++) {
            if (lcnt[sub] > cap_sub) report("leader list overflow", sub, lcnt[sub], cap_sub);
            for (uint32_t k = 0; k < lcnt[sub] && k < cap_sub; k++) got.push_back({lkey[(size_t)sub * cap_sub + k], lval[(size_t)sub * cap_sub + k]});
        }
        std::sort(want.begin(), want.end());
        std::sort(got.begin(), got.end());
        if (want.size() != got.size()) report("leaders (count)", 0, got.size(), want.size());
        for (size_t i = 0; i < std::min(want.size(), got.size()); i++)
            if (want[i] != got[i]) report("leaders", i, got[i].first, want[i].first);
    }
    if (inc != R.inc) report("incidences", 0, inc, R.inc);
    if (max_group != R.max_group) report("max_group", 0, max_group, R.max_group);
    if (groups != R.groups) report("groups", 0, groups, R.groups);
    {   // K0 inside the copy of the table (index_gather_rows): a table whose rows stand in REVERSE order is copied back into
        // this one's order; the copy must be the table and the window offsets those K0 made above
        std::vector<uint64_t> Hrev((size_t)n * t.stride), Hcopy((size_t)n * t.stride, 0x1111ull);
        std::vector<uint32_t> inv(n), cnt_rev(n);
        for (uint32_t r = 0; r < n; r++) {
            inv[r] = n - 1 - r;
            cnt_rev[n - 1 - r] = off[r + 1] - off[r];
            std::copy(t.H.begin() + (size_t)r * t.stride, t.H.begin() + (size_t)(r + 1) * t.stride, Hrev.begin() + (size_t)(n - 1 - r) * t.stride);
        }
        std::vector<unsigned char> lb2(plan.lb_bytes + 16, 0xCD);
        if (index_gather_rows(plan, Hrev.data(), inv.data(), cnt_rev.data(), Hcopy.data(), lb2.data(), nullptr) != hipSuccess) report("index_gather_rows", 0, 1, 0);
        if (Hcopy != t.H) report("the copy of the table", 0, 1, 0);
        if (memcmp(lb2.data(), lb.data(), plan.lb_bytes) != 0) report("window offsets made by the copy", 0, 1, 0);
    }
    printf("%-28s %s: %s\n", name.c_str(), geom, bad ? "MISMATCH" : "ok");
    return bad ? 1 : 0;
}

int main(int argc, char **argv)
{
    const std::string which = argc > 1 ? argv[1] : "all";
    int rc = 0;
    struct Case { const char *name; uint32_t n, s; uint64_t seed; const char *kind; bool fallback; double dens_scale; };
    const Case cases[] = {
        {"random_small", 40, 64, 1, "random", false, 1.0},
        {"random_two_blocks", 700, 48, 2, "random", false, 1.0},
        {"clusters", 600, 96, 3, "clusters", false, 1.0},
        {"clusters_windows", 600, 96, 13, "clusters", false, 12.0},   // the same with a dozen times the buckets: windows of several buckets
        {"ragged", 530, 80, 4, "ragged", false, 1.0},
        {"ragged_windows", 1030, 80, 14, "ragged", false, 25.0},
        {"copies_out", 300, 64, 5, "copies", false, 1.0},
        {"sizes", 520, 64, 6, "sizes", false, 1.0},
        {"top_bit", 200, 64, 7, "top", false, 1.0},
        {"one_row", 1, 1000, 8, "random", false, 1.0},
        {"pieces", 512, 40, 9, "random", false, 0.25},            // full buckets, one block of rows: tiles beyond one piece
        {"many_buckets", 513, 128, 10, "random", false, 60.0},    // sparse buckets, two sort passes
        {"three_passes", 100, 128, 15, "random", false, 2000.0},  // windows of 512 buckets
        {"clade", 600, 64, 11, "clade", false, 1.0},              // every value held by hundreds of rows: the stable counting sort takes them as they arrive
        {"big_random", 2000, 32, 12, "random", false, 0.0005},    // buckets beyond the LDS capacity: the two-level sort, parts of many sub-buckets
        {"big_clade", 10000, 16, 13, "clade", false, 1.0},        // values held by 7 000 rows: more than the LDS takes, streamed out in row order
        {"big_clade_fit", 7000, 16, 14, "clade", false, 1.0},     // values held by 5 000 rows in buckets beyond the capacity: a part of one sub-bucket
        {"twins", 2500, 16, 16, "twins", false, 1.0},             // neighbouring values with 1 800 holders each: ranked by comparison in LDS
        {"big_twins", 10000, 16, 15, "twins", true, 1.0},         // two values, thousands of holders each, that differ in the last bit: the flag
    };
    if (which == "fuzz") {
        // index_emu fuzz <seed> <cases>: tables of random shape and kind, bucket widths from a twentieth to twenty times the plan's
        // (full buckets, several pieces per tile, buckets beyond the LDS on one side; windows of hundreds of buckets on the other)
        std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 1);
        const int count = argc > 3 ? atoi(argv[3]) : 50;
        const char *kinds[] = {"random", "ragged", "clusters", "copies", "sizes", "top", "clade"};
        const double scales[] = {1.0, 1.0, 0.25, 0.05, 4.0, 20.0, 300.0};
        for (int i = 0; i < count; i++) {
            const char *kind = kinds[rng() % 7];
            const uint32_t n = 1 + (uint32_t)(rng() % (rng() % 4 == 0 ? 1600 : 400)), s2 = 1 + (uint32_t)(rng() % (rng() % 3 == 0 ? 220 : 60));
            double scale = scales[rng() % 7];
            const uint64_t seed = rng();
            // (a workgroup per bucket: keep their number where the emulator finishes in seconds)
            while (scale > 1.0 && (double)n * s2 * scale / 2560.0 > 3000.0) scale /= 2.0;
            char name[96];
            snprintf(name, sizeof name, "fuzz %d: %s x%g seed %llu", i, kind, scale, (unsigned long long)seed);
            rc |= run_case(name, n, s2, seed, kind, false, scale);
        }
        printf(rc ? "FAILED\n" : "all cases agree\n");
        return rc;
    }
    for (const Case &c : cases)
        if (which == "all" || which == c.name) rc |= run_case(c.name, c.n, c.s, c.seed, c.kind, c.fallback, c.dens_scale);
    printf(rc ? "FAILED\n" : "all cases agree\n");
    return rc;
}
