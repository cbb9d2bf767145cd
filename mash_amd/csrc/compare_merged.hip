// compare_merged.hip — gfx950 pairwise comparison, "merged rows" tile kernel.
//
// Same contract as compare.hip (the merge loop of compareSketches,
// CommandDistance.cpp:347-385, rank formulation described there), different tile engine:
// ALL R rows of a tile share ONE bucketed table in LDS, so a column element is probed once
// for the whole tile instead of once per row.
//
//   tile  : up to 16 rows listed explicitly (rows of one hash-density class, see
//           host_compare.cpp::run_compare) x a range of columns;
//   table : R*s entries {32-bit prefix, 16-bit tag = row<<idx_bits | index-in-row}, grouped
//           by bucket = mulhi(prefix, scale_tile) (CSR layout, load ~0.65); inside a bucket
//           one entry of every DISTINCT prefix comes first;
//   dir   : u16 per bucket = first entry of the bucket, bit 15 set when the bucket holds
//           more than MR_W distinct prefixes.
//
// Every value occurring in any row of the tile sits in the bucket its prefix maps to, so
// "b occurs in some row" <=> one of the bucket's distinct prefixes equals b's prefix (then
// verified on the 64-bit value in HBM/L2).  Unrelated sketches never tie, so for them a
// column costs one probe per element for all R rows plus one rank test per row per 64*KU
// elements:  rank(B[q] in row r) >= s-1  <=>  A_r[s-1-q+c_r - 1] < B[q]  — a single
// load per row (tested conservatively on prefixes).  Columns are streamed as 32-bit
// prefixes (mg_table keeps padded u32 images of the table, one per density class's shift),
// halving the kernel's dominant HBM traffic; 64-bit values are fetched only for tied lanes.
// Matches (ties) are ranked exactly with ballot + mbcnt using the index stored in the tag
// (it IS the lower bound of the matched value in its row).
//
// WIN = true (large sketches): the same kernel over ONE VALUE WINDOW of the hash range per launch.
// R*s <= ~16 000 entries would leave 16000/s rows per tile; instead a tile always has 16 rows and
// its table holds only their hashes with prefix in [win_lo, win_hi) (tags: index inside the row's
// window, lo_of(r) restores the index in the row), a column streams only its elements of the
// window, and a pair keeps {common, 0x80000000 | matches} in its output slot from launch to launch
// until the rank test, the end of the column or the last window decides it.  All indices in the
// rank arithmetic stay global, so the decision rule is the one above.  Host side: run_compare.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "compare_internal.h"

namespace mg {

constexpr int MR_NT = 1024;
constexpr int MR_NW = MR_NT / 64;
constexpr int MR_W = 4;                      // window: entries read per probe
constexpr int MR_EPT = 20;                   // table entries built per thread (R*s <= 20480)
constexpr int MR_CB = 8;                     // consecutive columns per wave batch (64 B of output per row)
constexpr int MR_KU_DEFAULT = 3;             // 64-element blocks of a column between rank tests (template KU)
constexpr uint32_t MR_OVF = 0x8000u;         // dir flag: bucket has more than MR_W entries

struct MergedHdr {
    uint32_t collide;    // set by the build when two DIFFERENT values of the tile share a prefix
    uint32_t scale;      // bucket = mulhi(prefix, scale)
    uint32_t xmax;       // largest prefix present in the tile
    uint32_t nent;       // total entries E
    uint32_t row_n[32];
    uint32_t row_base[32];   // first entry id of row r in (row, index) enumeration order
    uint32_t row_id[32];     // table row index of slot r (0xFFFFFFFF = unused)
};

// tag = row << idx_bits | index-in-row ; idx_bits = bits needed for an index < s (>= 10)
__host__ __device__ inline uint32_t merged_idx_bits(uint32_t s)
{
    uint32_t b = 10;
    while ((1u << b) < s) b++;
    return b;
}

constexpr size_t MR_LDS_LIMIT = 160 * 1024 - 512;       // dynamic part; 512 B left for the static arrays

// LDS bytes of a tile table of `entries` entries in nb buckets
__host__ __device__ constexpr size_t merged_lds_bytes_e(uint32_t R, size_t entries, uint32_t nb)
{
    const size_t ecap = entries + MR_W;
    return 512 + ((size_t)nb + 8) * 2 + ((ecap * 4 + 15) & ~(size_t)15) + ((ecap * 2 + 15) & ~(size_t)15) +
           (size_t)MR_NW * R * MR_CB * 8;                             // + per-wave output staging
}

__host__ __device__ inline size_t merged_lds_bytes_nb(uint32_t R, uint32_t s, uint32_t nb)
{
    return merged_lds_bytes_e(R, (size_t)R * s, nb);
}

// Window mode (large sketches): a tile holds 16 rows' hashes of ONE value window, at most
// MR_WIN_ENTRIES of them, in MR_WIN_BUCKETS buckets -- the geometry of an s = 1000 tile.
// Up to 32 rows share a window tile: unrelated pairs are decided by the lower ~54 % of the hash
// range (the union of two sketches reaches s elements there), so a window holds ~0.54 s entries
// of a row and 29 rows fit where 16 whole rows did -- every probe then serves 29 pairs.
constexpr uint32_t MR_WIN_ROWS = 32;
constexpr uint32_t MR_WIN_ENTRIES = 16000;
constexpr uint32_t MR_WIN_BUCKETS = 24576;
constexpr uint32_t MR_WIN_IDX_BITS = 11;               // index of an entry inside its row's window (< 2048)
// per-wave output staging of a window tile: 4 B per pair ({common, denom} as two u16) while
// s < 32768, else 8 B and at most 16 rows
static_assert(merged_lds_bytes_e(MR_WIN_ROWS / 2, MR_WIN_ENTRIES, MR_WIN_BUCKETS) <= 160 * 1024 - 512, "window tile exceeds LDS");

// Bucket count: at least 0.8 buckets per entry (power of two), then as many more as LDS holds
// up to two per entry -- buckets with more than MR_W entries cost an extra scan per probe that
// lands in them, and their share falls quickly with the load factor.
__host__ __device__ inline uint32_t merged_buckets(uint32_t R, uint32_t s)
{
    uint32_t nb = 1024;
    while (nb * 5 < R * s * 4) nb <<= 1;
    while (nb + 1024 <= 2 * R * s && nb + 1024 <= 65536 - 16 && merged_lds_bytes_nb(R, s, nb + 1024) <= MR_LDS_LIMIT) nb += 1024;
    return nb;
}

__host__ __device__ inline size_t merged_lds_bytes(uint32_t R, uint32_t s)
{
    return merged_lds_bytes_nb(R, s, merged_buckets(R, s));
}

bool compare_merged_supported(uint32_t s) { return s >= 1 && s <= 16384 && merged_lds_bytes(1, s) <= MR_LDS_LIMIT; }

uint32_t compare_merged_rows(uint32_t s)
{
    uint32_t r = 16;
    const uint32_t rcap = 1u << (16 - merged_idx_bits(s));
    if (r > rcap) r = rcap;
    while (r > 1 && (merged_lds_bytes(r, s) > MR_LDS_LIMIT || (uint64_t)r * s > 32767u ||
                     (uint64_t)r * s > (uint64_t)MR_NT * MR_EPT)) r--;
    return r;
}

// values beyond the image's range saturate to 0xFFFFFFFD (0xFFFFFFFE = table sentinel,
// 0xFFFFFFFF = padding); the rows a shift was chosen for stay below all three
__device__ __forceinline__ uint32_t mr_prefix(uint64_t v, uint32_t shr)
{
    const uint64_t t = v >> shr;
    return t >= 0xFFFFFFFDull ? 0xFFFFFFFDu : (uint32_t)t;
}

// W0 (window mode): this launch is the FIRST window -- no pair carries state yet and every column is
// live, so the state loads, the live-column masks and their bookkeeping are compiled out of the
// launch that does nine tenths of the work.
template <int KU, bool WIN, bool W0 = false>
__global__ __launch_bounds__(MR_NT) void compare_merged_kernel(CompareArgs a)
{
    constexpr int MR_KU = KU;
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t s = a.s;
    const uint32_t R = a.rows_per_tile;
    const uint32_t NB = a.nbuckets;
    const uint32_t ecap = (WIN ? a.win_ecap : R * s) + MR_W;
    MergedHdr *hdr = reinterpret_cast<MergedHdr *>(smem);
    uint16_t *dir = reinterpret_cast<uint16_t *>(smem + 512);                       // [NB + 8]
    uint32_t *cnt32 = reinterpret_cast<uint32_t *>(dir);                            // build-time view
    uint32_t *pfx = reinterpret_cast<uint32_t *>(smem + 512 + (size_t)(NB + 8) * 2);   // [ecap]
    uint16_t *tag = reinterpret_cast<uint16_t *>(reinterpret_cast<unsigned char *>(pfx) + (((size_t)ecap * 4 + 15) & ~(size_t)15));
    unsigned char *stage_all = reinterpret_cast<unsigned char *>(tag) + (((size_t)ecap * 2 + 15) & ~(size_t)15);
    __shared__ uint32_t s_wsum[MR_NW + 2];
    __shared__ uint64_t s_rowmax[32];
    __shared__ uint32_t s_rowlo[32], s_rowlen[32];      // WIN: first index of the row's window; full row length
    constexpr uint32_t MAXR = WIN ? 32u : 16u;          // rows a tile may list
    // WIN: results are staged as {common, denom} u16 pairs while s < 32768 (bit 15 of denom = in progress)
    const bool pack = WIN && a.stage_pack != 0;

    // (fields are read individually: indexing a by-value copy of rows[] would put the tile in scratch)
    // Workgroups are dealt to the 8 XCDs round-robin by index, and each XCD has its own L2.  With
    // xcd_remap, XCD x works through the x-th contiguous eighth of the tile list (which is ordered
    // column chunk major), so a column chunk is streamed into one L2 instead of all eight.
    // Measured on C3: 18.4 -> 15.5e9 pairs/s -- 32 workgroups walking the same columns at the same
    // time queue up on the same L2 channels, and HBM is not this kernel's limit -- so it is off
    // (MASHGPU_COMPARE_XCD=1 turns it on for experiments).
    uint32_t tile_index = blockIdx.x;
    if (a.xcd_remap) {
        const uint32_t nt = gridDim.x, x = blockIdx.x & 7u, q = blockIdx.x >> 3;
        uint32_t start = 0;
        for (uint32_t y = 0; y < x; y++) start += (nt - 1u - y) / 8u + 1u;      // blocks with index % 8 == y (nt > y)
        tile_index = start + q;
    }
    const MergedTile *tile_p = a.mtiles + tile_index;
    struct { uint32_t col0, col1; } tile = {tile_p->col0, tile_p->col1};
    const int tid = threadIdx.x;
#ifdef MASHGPU_TILE_CLOCKS
    if (a.dbg && tid == 0) a.dbg[3 * (uint64_t)blockIdx.x] = __builtin_readcyclecounter();
#endif
    const uint32_t lane = tid & 63;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);

    if (WIN && !W0) {
        // later windows: a tile none of whose columns still has a pair in progress has nothing to do
        const uint8_t *m = a.win_mask + (uint64_t)blockIdx.x * MR_NW * a.win_kmax;
        const uint32_t nbytes = MR_NW * a.win_kmax;
        uint32_t any = 0;
        for (uint32_t b = tid; b < nbytes; b += MR_NT) any |= m[b];
        if (__syncthreads_or((int)any) == 0) return;
    }
    // ------------------------------------------------------------------ build the tile table
    if (tid < 32) {
        uint32_t n = 0;
        uint64_t mx = 0;
        uint32_t rid = 0xFFFFFFFFu;
        if ((uint32_t)tid < R && (uint32_t)tid < MAXR) rid = tile_p->rows[tid];
        if (rid != 0xFFFFFFFFu) {
            const uint64_t i = rid;
            n = a.row_nhash[i];
            if (n > s) n = s;
            if (n > 0) mx = a.row_pfx[i * a.row_pfx_stride + n - 1];
        }
        if (WIN) {
            uint32_t lo = 0, hi = 0;
            if (rid != 0xFFFFFFFFu) {
                lo = a.row_win[(uint64_t)rid * (a.nwin + 1) + a.win];
                hi = a.row_win[(uint64_t)rid * (a.nwin + 1) + a.win + 1];
            }
            s_rowlo[tid] = lo;
            s_rowlen[tid] = n;
            n = hi - lo;                                  // entries of this row inside the window
        }
        hdr->row_n[tid] = n;
        hdr->row_id[tid] = rid;
        s_rowmax[tid] = mx;
    }
    for (uint32_t b = tid; b < (NB + 8) / 2; b += MR_NT) cnt32[b] = 0;
    __syncthreads();
    if (tid == 0) {
        uint32_t e = 0;
        for (uint32_t r = 0; r < 32; r++) {
            hdr->row_base[r] = e;
            e += hdr->row_n[r];
        }
        uint64_t mxall = 0;
        for (uint32_t r = 0; r < 32; r++) mxall = s_rowmax[r] > mxall ? s_rowmax[r] : mxall;
        const uint32_t xmax = (uint32_t)mxall;                    // largest prefix of the tile's rows
        // bucket = mulhi(prefix - origin, scale): the whole value range, or the launch's window
        const uint64_t sc = ((uint64_t)NB << 32) / (WIN ? (uint64_t)(a.win_hi - a.win_lo) : (uint64_t)xmax + 1ULL);
        hdr->collide = 0;
        hdr->xmax = xmax;
        hdr->scale = sc > 0xFFFFFFFFULL ? 0xFFFFFFFFu : (uint32_t)sc;
        hdr->nent = e;
    }
    __syncthreads();
    const uint32_t scale = hdr->scale, E = hdr->nent;

    // pass 1: bucket histogram; entries are enumerated row-major, thread tid owns
    // e = tid, tid + NT, ... (coalesced loads of the rows' prefix images)
    const uint32_t idx_bits = WIN ? MR_WIN_IDX_BITS : merged_idx_bits(s);
    const uint32_t idx_mask = (1u << idx_bits) - 1u;
    const uint32_t origin = WIN ? a.win_lo : 0u;
    // WIN: tags hold the index inside the row's window; lo_of(r) turns it into the index in the row
    auto lo_of = [&](uint32_t r) -> uint32_t { return WIN ? s_rowlo[r & 31u] : 0u; };
    uint32_t e_pfx[MR_EPT], e_bs[MR_EPT];    // prefix; bucket | slot << 16, kept in registers
    uint16_t e_tag[MR_EPT];
#pragma unroll
    for (int t = 0; t < MR_EPT; t++) {
        const uint32_t e = (uint32_t)tid + (uint32_t)t * MR_NT;
        uint32_t r, idx;
        bool have;
        if (WIN) {
            // entries are the concatenated window ranges of the rows: find the row of entry e
            r = 0;                                      // last row whose base is <= e (row_base is non-decreasing)
#pragma unroll
            for (uint32_t step = 16; step >= 1; step >>= 1) r += hdr->row_base[r + step] <= e ? step : 0u;
            idx = e - hdr->row_base[r];
            have = e < E;
        } else {
            r = e / s;
            idx = e - r * s;
            have = r < R && idx < hdr->row_n[r];
        }
        e_bs[t] = 0xFFFFFFFFu;
        e_pfx[t] = 0;
        e_tag[t] = 0;
        if (have) {
            const uint32_t x = a.row_pfx[(uint64_t)hdr->row_id[r] * a.row_pfx_stride + lo_of(r) + idx];
            uint32_t bk = __umulhi(x - origin, scale);
            if (WIN && bk >= NB) bk = NB - 1;
            const uint32_t old = atomicAdd(&cnt32[bk >> 1], (bk & 1u) ? 0x10000u : 1u);
            const uint32_t slot = (bk & 1u) ? (old >> 16) : (old & 0xFFFFu);
            e_pfx[t] = x;
            e_tag[t] = (uint16_t)((r << idx_bits) | idx);
            e_bs[t] = bk | (slot << 16);             // bk < 32768
        }
    }
    __syncthreads();
    // exclusive scan of the u16 counters -> dir (bit 15: bucket larger than the probe window)
    {
        const uint32_t per = (NB + MR_NT - 1) / MR_NT;                 // buckets per thread (16)
        const uint32_t b0 = tid * per;
        uint32_t sum = 0;
        for (uint32_t b = b0; b < b0 + per && b < NB; b++) sum += dir[b];
        // block exclusive scan of `sum`
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(inc, d);
            if (lane >= (uint32_t)d) inc += t;
        }
        if (lane == 63) s_wsum[wid] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wid; w++) woff += s_wsum[w];
        uint32_t run = woff + inc - sum;
        for (uint32_t b = b0; b < b0 + per && b < NB; b++) {
            const uint32_t c = dir[b];
            dir[b] = (uint16_t)run;                        // the oversize flag is set by pass 3
            run += c;
        }
        if (tid == MR_NT - 1) {
            for (uint32_t b = NB; b < NB + 8; b++) dir[b] = (uint16_t)E;
        }
    }
    __syncthreads();
    // pass 2: scatter
#pragma unroll
    for (int t = 0; t < MR_EPT; t++) {
        if (e_bs[t] != 0xFFFFFFFFu) {
            const uint32_t bk = e_bs[t] & 0xFFFFu, slot = e_bs[t] >> 16;
            const uint32_t pos = (dir[bk] & 0x7FFFu) + slot;
            pfx[pos] = e_pfx[t];
            tag[pos] = e_tag[t];
        }
    }
    if (tid < MR_W) { pfx[E + tid] = 0xFFFFFFFEu; tag[E + tid] = 0xFFFFu; }   // equals no real prefix, nor the padding
    __syncthreads();
    // pass 3, per bucket: (a) move one entry of every DISTINCT prefix to the front, so the probe
    // window (the first MR_W entries) decides "is this prefix in the tile" no matter how many rows
    // share a value -- related rows in one tile would otherwise turn every bucket of theirs into
    // an oversize one; a bucket is oversize only if it holds more than MR_W distinct prefixes.
    // (b) Check that equal prefix means equal value inside the tile: every duplicate compares its
    // 64-bit value with the front entry of its prefix (loads happen only where rows share
    // values).  Tiles that pass -- practically all -- let the exact path trust prefix equality
    // between table entries and verify one representative per column element.
    {
        const uint32_t per = (NB + MR_NT - 1) / MR_NT;
        const uint32_t b0 = tid * per;
        for (uint32_t b = b0; b < b0 + per && b < NB; b++) {
            const uint32_t st = dir[b] & 0x7FFFu, en = dir[b + 1] & 0x7FFFu;
            const uint32_t c = en - st;
            uint32_t nd = c;
            if (c >= 2) {
                nd = 1;
                for (uint32_t i = 1; i < c; i++) {
                    const uint32_t pi = pfx[st + i];
                    const uint16_t ti = tag[st + i];
                    uint32_t d = 0;
                    while (d < nd && pfx[st + d] != pi) d++;
                    if (d == nd) {                                       // new prefix: swap to the front region
                        if (i != nd) {
                            pfx[st + i] = pfx[st + nd]; tag[st + i] = tag[st + nd];
                            pfx[st + nd] = pi; tag[st + nd] = ti;
                        }
                        nd++;
                    } else {
                        const uint32_t tg0 = tag[st + d], tg1 = ti;
                        const uint64_t v0 = a.row_hashes[(uint64_t)hdr->row_id[tg0 >> idx_bits] * a.row_stride + lo_of(tg0 >> idx_bits) + (tg0 & idx_mask)];
                        const uint64_t v1 = a.row_hashes[(uint64_t)hdr->row_id[tg1 >> idx_bits] * a.row_stride + lo_of(tg1 >> idx_bits) + (tg1 & idx_mask)];
                        if (v0 != v1) hdr->collide = 1;
                    }
                }
            }
            // (c) order the duplicates by (prefix, row): the rows sharing a value then form one run,
            // and the exact path reads row r's entry at run start + popcount(lower rows)
            for (uint32_t i = st + nd + 1; i < st + c; i++) {
                const uint32_t pi = pfx[i];
                const uint16_t ti = tag[i];
                uint32_t q = i;
                while (q > st + nd && (pfx[q - 1] > pi || (pfx[q - 1] == pi && tag[q - 1] > ti))) {
                    pfx[q] = pfx[q - 1]; tag[q] = tag[q - 1];
                    q--;
                }
                pfx[q] = pi; tag[q] = ti;
            }
            if (nd > (uint32_t)MR_W) dir[b] = (uint16_t)(st | MR_OVF);
        }
    }
    __syncthreads();
    const bool clean = hdr->collide == 0;

#ifdef MASHGPU_TILE_CLOCKS
    if (a.dbg && tid == 0) a.dbg[3 * (uint64_t)blockIdx.x + 1] = __builtin_readcyclecounter();
#endif
    // ------------------------------------------------------------------ stream columns
    const uint32_t my_n = WIN ? (lane < 32 ? s_rowlen[lane] : 0) : (lane < 32 ? hdr->row_n[lane] : 0);   // row `lane`
    // WIN: index in row `lane` of its first hash at or above the window's end
    const uint32_t my_lo = WIN && lane < 32 ? s_rowlo[lane] : 0u;
    const uint32_t my_pend = WIN && lane < 32 ? my_lo + hdr->row_n[lane] : 0u;
    const uint32_t my_id = lane < 32 ? hdr->row_id[lane] : 0xFFFFFFFFu;     // table row of slot `lane`
    const uint32_t *my_row = a.row_pfx + (uint64_t)(my_id != 0xFFFFFFFFu ? my_id : hdr->row_id[0]) * a.row_pfx_stride;

    // Column streaming is software-pipelined with UNCONDITIONAL loads (the image's rows are
    // padded, no index needs clamping):
    //   in flight while group g is probed:  rank-test operand of g, data of group g+1,
    //   and (issued during group 0) the first group + length of the wave's next column.
    // VMEM returns in order, so the small L2-resident rank-test operand is always issued
    // BEFORE the streaming loads that follow it.
    // (src and qbase are uniform: the group's base goes into scalar registers and every load is
    //  base + lane * 4 + an immediate, no per-load address arithmetic on the vector unit)
    auto load_group = [&](const uint32_t *src, uint32_t qbase, uint32_t (&dst)[MR_KU]) {
        const uint32_t *gp = src + (uint32_t)__builtin_amdgcn_readfirstlane((int)qbase);
#pragma unroll
        for (int u = 0; u < MR_KU; u++) dst[u] = gp[u * 64 + lane];   // rows of the image are padded (0xFFFFFFFF)
    };
    const uint32_t rows_all = (uint32_t)__ballot(my_id != 0xFFFFFFFFu);       // slots in use
    // Per row, once per tile: which columns it is compared with (j < my_lim: its own index in the
    // triangle, every column in a rect; 0 for unused slots) and where its results start in the
    // output (pair index of column 0) -- the column loop then needs neither the triangle flag nor
    // row_begin / ncols / out_base, which are scalar registers it does not have.
    const uint32_t my_lim = my_id == 0xFFFFFFFFu ? 0u : (a.triangle ? my_id : 0xFFFFFFFFu);
    uint64_t my_obase = 0;
    if (my_id != 0xFFFFFFFFu) {
        const uint64_t i = my_id;
        my_obase = a.triangle ? i * (i - 1) / 2 - a.out_base : (i - a.row_begin) * a.ncols;
    }
    // Wave w owns batches of MR_CB consecutive columns: batch k -> columns
    // col0 + (k*NW + w)*CB ... +CB-1.  Results of a batch are staged in LDS and written
    // as 64-B row segments (nontemporal), instead of 8-B scattered stores that thrash L2.
    unsigned char *stage_b = stage_all + (size_t)wid * R * MR_CB * (pack ? 4u : 8u);   // [R rows][CB]
    uint2 *stage = reinterpret_cast<uint2 *>(stage_b);
    uint32_t *stage_p = reinterpret_cast<uint32_t *>(stage_b);            // packed form
    auto col_of = [&](uint32_t t) -> uint32_t {
        return tile.col0 + ((t / MR_CB) * MR_NW + wid) * MR_CB + (t % MR_CB);
    };
    auto flush_batch = [&](uint32_t jb, uint32_t ncols_done, uint32_t procmask) {
        // lane -> (row = lane/4, two columns 2*(lane%4), +1) : 16 B per lane, 64 B per row
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t c0 = (lane & 3u) * 2;
        for (uint32_t r = lane >> 2; r < ((R + 15u) & ~15u); r += 16) {   // uniform trip count: 16 rows per pass
        const uint32_t lim = (uint32_t)__shfl((int)my_lim, (int)(r & 31u));                // columns of slot r's row
        const uint64_t obase = (uint64_t)(uint32_t)__shfl((int)(uint32_t)my_obase, (int)(r & 31u)) |
                               ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(my_obase >> 32), (int)(r & 31u)) << 32);
        if (r < R && lim != 0) {
            uint4 v;
            if (pack) {
                const uint2 w = *reinterpret_cast<const uint2 *>(&stage_p[r * MR_CB + c0]);
                const uint32_t d0 = w.x >> 16, d1 = w.y >> 16;
                v.x = w.x & 0xFFFFu;
                v.y = (d0 & 0x8000u) ? (0x80000000u | (d0 & 0x7FFFu)) : d0;
                v.z = w.y & 0xFFFFu;
                v.w = (d1 & 0x8000u) ? (0x80000000u | (d1 & 0x7FFFu)) : d1;
            } else {
                v = *reinterpret_cast<const uint4 *>(&stage[r * MR_CB + c0]);
            }
            const uint64_t j0 = (uint64_t)jb + c0;
            // WIN: only the columns this launch worked on (the others keep their earlier result)
            const bool in0 = WIN ? ((procmask >> c0) & 1u) != 0 : c0 < ncols_done;
            const bool in1 = WIN ? ((procmask >> (c0 + 1)) & 1u) != 0 : c0 + 1 < ncols_done;
            const bool ok0 = in0 && j0 < lim;
            const bool ok1 = in1 && j0 + 1 < lim;
            uint2 *dst = a.out + (obase + j0);
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            if (ok0 && ok1 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                u32x4 w = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(dst));
            } else {
                u32x2 w0 = {v.x, v.y}, w1 = {v.z, v.w};
                if (ok0) __builtin_nontemporal_store(w0, reinterpret_cast<u32x2 *>(dst));
                if (ok1) __builtin_nontemporal_store(w1, reinterpret_cast<u32x2 *>(dst + 1));
            }
        }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // WIN: the column range of this launch's window, [col_win[j][win], col_win[j][win+1]), and
    // the state a pair carries from window to window, kept in its output slot until it is final:
    // {common, 0x80000000 | matches so far} while in progress, {common, denom} once decided.
    auto win_range = [&](uint32_t jj, uint32_t &lo, uint32_t &hi) {
        const uint32_t *wp = a.col_win + (uint64_t)jj * (a.nwin + 1) + a.win;
        lo = wp[0];
        hi = wp[1];
    };
    auto load_state = [&](uint32_t jj) -> uint2 {
        uint2 v = make_uint2(0u, 0u);
        if (lane < R && jj < my_lim) v = a.out[my_obase + jj];
        return v;
    };
    uint32_t ncol[MR_KU];
    uint32_t nB_next = 0;
    uint32_t nx_lo = 0, nx_hi = 0, n2_lo = 0, n2_hi = 0;   // WIN: window ranges of the next two columns
    uint2 st_next = make_uint2(0u, 0u);                    // WIN: carried state of the next column (lane r <-> row r)
    // WIN: a column whose 16 pairs are all decided is not touched again.  Every wave keeps one
    // byte per batch of its columns (bit c: column c of the batch still has a pair in progress),
    // written when the batch is flushed and read by the same wave of the next launch; lane l of
    // mk_lane holds the byte of batch mk_base + l.
    uint8_t *wmask = WIN ? a.win_mask + ((uint64_t)blockIdx.x * MR_NW + wid) * a.win_kmax : nullptr;
    uint32_t mk_lane = 0xFFu, mk_base = 0;
    auto mk_load = [&](uint32_t kb) {
        mk_base = kb;
        const uint32_t k = kb + lane;
        mk_lane = k < a.win_kmax ? (uint32_t)wmask[k] : 0u;
    };
    // first position t' >= t of the wave's column sequence whose column is still live (or beyond the tile)
    auto next_col = [&](uint32_t t) -> uint32_t {
        if (!WIN) return t;
        if (W0) return t;
        for (;;) {
            const uint32_t k = t / MR_CB;
            if (col_of(k * MR_CB) >= tile.col1) return t;
            if (k - mk_base >= 64u) mk_load(k & ~63u);
            const uint32_t rel = (uint32_t)__builtin_amdgcn_readfirstlane((int)(k - mk_base));
            const uint32_t m = ((uint32_t)__builtin_amdgcn_readlane((int)mk_lane, (int)rel) & 0xFFu) >> (t % MR_CB);
            if (m != 0) return t + (uint32_t)__builtin_ctz(m);
            uint64_t nz = __ballot(mk_lane != 0);
            nz = rel + 1 < 64u ? nz >> (rel + 1) : 0ull;
            t = nz != 0 ? (k + 1 + (uint32_t)__builtin_ctzll(nz)) * MR_CB : (mk_base + 64u) * MR_CB;
        }
    };
    if (WIN && !W0) mk_load(0);
    uint32_t procmask = 0, progmask = 0;                   // WIN: columns of the open batch worked on / still in progress
    uint32_t tcol = next_col(0);
    uint32_t t1 = next_col(tcol + 1), t2 = next_col(t1 + 1);   // the next two live columns
    uint32_t j = col_of(tcol);
    if (j < tile.col1) {
        if (WIN) {
            win_range(j, nx_lo, nx_hi);
            const uint32_t j2 = col_of(t1);
            win_range(j2 < tile.col1 ? j2 : j, n2_lo, n2_hi);
            if (!W0) st_next = load_state(j);
        }
        load_group(a.col_pfx + (uint64_t)j * a.col_pfx_stride, nx_lo, ncol);
        nB_next = a.col_nhash[j];
    }
    while (j < tile.col1) {
        uint32_t nB = nB_next < s ? nB_next : s;
        const uint32_t *bsrc = a.col_pfx + (uint64_t)j * a.col_pfx_stride;
        uint32_t cur[MR_KU], nxt[MR_KU];
#pragma unroll
        for (int u = 0; u < MR_KU; u++) cur[u] = ncol[u];
        // rows of the tile this column is compared with (triangle: only rows i > j)
        const uint32_t valid = (uint32_t)__ballot(j < my_lim);
        uint32_t active = valid, brokem = 0;
        uint32_t st_call = 0, st_common = 0;                             // lane r <-> row r
        const uint32_t qlo = WIN ? nx_lo : 0u;                           // this launch's part of the column
        const uint32_t qhi = WIN ? nx_hi : nB;
        uint32_t fin_denom = 0;
        if (WIN && !W0) {
            // pairs decided in an earlier window keep their result; the others resume
            const bool inprog = (st_next.y & 0x80000000u) != 0;
            st_common = st_next.x;
            st_call = inprog ? (st_next.y & 0x7FFFFFFFu) : 0u;
            fin_denom = st_next.y;
            active &= (uint32_t)__ballot(lane < R && inprog);
        }
        const uint32_t started = active;                                 // rows this launch works on
        const uint32_t ngroups = active == 0 ? 0 : (qhi - qlo + 64 * MR_KU - 1) / (64 * MR_KU);
        // prologue: rank-test operand of group 0, data of group 1, then the next column
        // (WIN: no rank test between groups -- the pair is decided exactly at the end of the window's part)
        int32_t t_chk = (int32_t)s - (int32_t)(64 * MR_KU) - (int32_t)qlo + (int32_t)st_call;   // s-1-qlast+c, qlast = qlo+64*KU-1
        uint32_t a_chk = WIN ? 0u : my_row[t_chk >= 1 ? (uint32_t)t_chk - 1 : 0];
        load_group(bsrc, qlo + 64 * MR_KU, nxt);
        {
            const uint32_t jnx = col_of(t1);
            const uint32_t jn = jnx < tile.col1 ? jnx : j;
            if (WIN) { nx_lo = n2_lo; nx_hi = n2_hi; }                   // jn's range (clamped duplicates are never used)
            load_group(a.col_pfx + (uint64_t)jn * a.col_pfx_stride, WIN ? nx_lo : 0u, ncol);
            nB_next = a.col_nhash[jn];
            if (WIN) {
                if (!W0) st_next = load_state(jn);
                const uint32_t j2x = col_of(t2);
                win_range(j2x < tile.col1 ? j2x : jn, n2_lo, n2_hi);
            }
        }
        // directory entries are fetched one group ahead (s0c: this group's, read while the previous
        // group was compared), so a group waits for ONE LDS round trip -- its window reads -- not two
        uint32_t s0c[MR_KU];
        if (ngroups != 0) {
#pragma unroll
            for (int u = 0; u < MR_KU; u++) {
                const uint32_t bk = __umulhi(cur[u] - origin, scale);
                s0c[u] = dir[bk < NB ? bk : NB];
            }
        }
        for (uint32_t g = 0; g < ngroups && active != 0; g++) {
            const uint32_t q0 = qlo + g * 64 * MR_KU;
            const bool col_end = q0 + 64 * MR_KU >= qhi;                 // WIN: end of the window's part (no rank test there)
            const uint32_t qlast = q0 + 64 * MR_KU - 1;

            // ---- one probe per element for ALL rows of the tile ----
            uint32_t x[MR_KU], s0[MR_KU], h[MR_KU][MR_W];
            uint64_t tiem[MR_KU];
            uint64_t anytie = 0;
#pragma unroll
            for (int u = 0; u < MR_KU; u++) {
                x[u] = cur[u];
                s0[u] = s0c[u];                              // dir[bucket of x]; prefixes above the tile's maximum (and padding) -> sentinel
            }
#pragma unroll
            for (int u = 0; u < MR_KU; u++)
#pragma unroll
                for (int w = 0; w < MR_W; w++) h[u][w] = pfx[(s0[u] & 0x7FFFu) + w];
            if (!col_end) {
#pragma unroll
                for (int u = 0; u < MR_KU; u++) {            // next group's directory entries, behind this group's window reads
                    const uint32_t bk = __umulhi(nxt[u] - origin, scale);
                    s0c[u] = dir[bk < NB ? bk : NB];
                }
            }
            uint64_t anyovf = 0;
#pragma unroll
            for (int u = 0; u < MR_KU; u++) {
                uint64_t t = 0;
#pragma unroll
                for (int w = 0; w < MR_W; w++) t |= __ballot(h[u][w] == x[u]);
                tiem[u] = t;
                anyovf |= __ballot(s0[u] > 0x7FFFu);
            }
            if (anyovf != 0) {
                // some element fell into a bucket with more than MR_W entries: compare it with the
                // rest of that bucket right here (a few LDS reads) instead of sending the whole
                // block down the exact path
#pragma unroll
                for (int u = 0; u < MR_KU; u++) {
                    bool hit = false;
                    if (s0[u] > 0x7FFFu) {
                        const uint32_t bk = __umulhi(x[u] - origin, scale);   // < NB: the sentinel bucket is never oversize
                        const uint32_t e1 = dir[bk + 1] & 0x7FFFu;
                        for (uint32_t e = (s0[u] & 0x7FFFu) + MR_W; e < e1; e++) hit |= pfx[e] == x[u];
                    }
                    tiem[u] |= __ballot(hit);
                }
            }
#pragma unroll
            for (int u = 0; u < MR_KU; u++) anytie |= tiem[u];

            bool c_changed = false;
            if (anytie != 0) {
                // (the window entries are made opaque here, so the compares below are recomputed rather
                //  than kept as twelve live 64-bit masks across the fast path -- scalar registers are
                //  what this kernel runs out of)
#pragma unroll
                for (int u = 0; u < MR_KU; u++)
#pragma unroll
                    for (int w = 0; w < MR_W; w++) asm volatile("" : "+v"(h[u][w]));
                // ---- exact path: which rows contain which elements, ranked in column order ----
                if (clean) {
                    // every table entry with b's prefix holds the same value (build pass 3).  Phase A,
                    // for all blocks of the group: collect the rows and where b's index in each is found
                    // (the first entry with b's prefix is its representative; further ones are the
                    // duplicates, one run ordered by row -- build pass 3c) and ISSUE the two 64-bit loads
                    // that verify one representative per element; phase B consumes them, so a group
                    // waits for one round trip to memory, not one per block.
                    uint32_t dupm[MR_KU], repv[MR_KU], firste[MR_KU];
                    uint64_t bv[MR_KU], vv[MR_KU];
#pragma unroll
                    for (int u = 0; u < MR_KU; u++) {
                        dupm[u] = 0; repv[u] = 0xFFFFFFFFu; firste[u] = 0; bv[u] = 0; vv[u] = 0;
                        if (tiem[u] == 0) continue;                      // uniform
                        const bool mine = (tiem[u] >> lane) & 1ULL;
                        if (mine) {
                            bv[u] = a.col_hashes[(uint64_t)j * a.col_stride + q0 + u * 64 + lane];   // 64-bit value only for tied lanes
                            uint32_t dupmask = 0, rep = 0xFFFFFFFFu, first_e = 0;
                            auto take = [&](uint32_t e) {
                                const uint32_t tg = tag[e];
                                if (rep == 0xFFFFFFFFu) rep = tg;
                                else {
                                    if (dupmask == 0) first_e = e;
                                    dupmask |= 1u << (tg >> idx_bits);
                                }
                            };
                            const uint32_t st = s0[u] & 0x7FFFu;
#pragma unroll
                            for (int w = 0; w < MR_W; w++)
                                if (h[u][w] == x[u]) take(st + w);
                            {
                                const uint32_t bk = __umulhi(x[u] - origin, scale);
                                const uint32_t e1 = dir[bk + 1] & 0x7FFFu;
                                for (uint32_t e = st + MR_W; e < e1; e++)
                                    if (pfx[e] == x[u]) take(e);
                            }
                            if (rep != 0xFFFFFFFFu)
                                vv[u] = a.row_hashes[(uint64_t)hdr->row_id[rep >> idx_bits] * a.row_stride + lo_of(rep >> idx_bits) + (rep & idx_mask)];
                            dupm[u] = dupmask; repv[u] = rep; firste[u] = first_e;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < MR_KU; u++) {
                        if (tiem[u] == 0) continue;                      // uniform
                        const uint32_t qb = q0 + u * 64;
                        const uint32_t rep = repv[u], first_e = firste[u];
                        uint32_t dupmask = dupm[u], rowmask = 0;
                        if (rep != 0xFFFFFFFFu) {
                            if (vv[u] == bv[u]) rowmask = dupmask | (1u << (rep >> idx_bits));   // else: same prefix, different value
                            else dupmask = 0;
                        }
                        const bool anydup = __ballot(dupmask != 0) != 0;
                        uint32_t any = rowmask;                           // rows involved anywhere in this block
#pragma unroll
                        for (int d = 32; d > 0; d >>= 1) any |= __shfl_xor(any, d);
                        const uint32_t rows_any = (uint32_t)__builtin_amdgcn_readfirstlane((int)any) & active;
                        for (uint32_t rest = rows_any; rest != 0; rest &= rest - 1) {
                            const int r = __builtin_ctz(rest);            // uniform
                            const bool mt = (rowmask >> r) & 1u;
                            uint32_t tg = rep;
                            if (anydup) {                                 // uniform
                                if (mt && (rep >> idx_bits) != (uint32_t)r)
                                    tg = tag[first_e + (uint32_t)__popc(dupmask & ((1u << r) - 1u))];
                            }
                            // (row r's window start comes from lane r's register)
                            const uint32_t idx = (tg & idx_mask) +
                                                 (WIN ? (uint32_t)__builtin_amdgcn_readlane((int)my_lo, r) : 0u);
                            uint32_t c_all = (uint32_t)__builtin_amdgcn_readlane((int)st_call, r);
                            uint32_t common = (uint32_t)__builtin_amdgcn_readlane((int)st_common, r);
                            const uint64_t mm = __ballot(mt);
                            const uint32_t before = c_all + __builtin_amdgcn_mbcnt_hi(
                                (uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0));
                            const uint32_t rank = qb + lane + idx - before;
                            common += (uint32_t)__popcll(__ballot(mt && rank < s));
                            c_all += (uint32_t)__popcll(mm);
                            st_call = (lane == (uint32_t)r) ? c_all : st_call;
                            st_common = (lane == (uint32_t)r) ? common : st_common;
                            c_changed = true;
                        }
                    }
                } else
#pragma unroll
                for (int u = 0; u < MR_KU; u++) {
                    if (tiem[u] == 0) continue;                          // uniform
                    const uint32_t qb = q0 + u * 64;
                    const bool mine = (tiem[u] >> lane) & 1ULL;
                    const uint64_t b = mine ? a.col_hashes[(uint64_t)j * a.col_stride + qb + lane] : 0;   // 64-bit value only for tied lanes
                    // scan my bucket: rows whose value equals b (verified on 64 bits), remember
                    // the index (= lower bound of b in that row) of up to the first 4 hits in regs
                    uint32_t rowmask = 0;
                    uint32_t hit_tag[MR_W];
#pragma unroll
                    for (int w = 0; w < MR_W; w++) hit_tag[w] = 0xFFFFFFFFu;
                    uint32_t extra_lo = 0, extra_hi = 0;                 // entries beyond the window (oversize bucket)
                    if (mine) {
                        const uint32_t st = s0[u] & 0x7FFFu;
#pragma unroll
                        for (int w = 0; w < MR_W; w++) {
                            if (h[u][w] == x[u] && st + w < E) {
                                const uint32_t tg = tag[st + w];
                                const uint32_t r = tg >> idx_bits, idx = (tg & idx_mask) + lo_of(tg >> idx_bits);
                                const uint64_t v = a.row_hashes[(uint64_t)hdr->row_id[r] * a.row_stride + idx];
                                if (v == b) { rowmask |= 1u << r; hit_tag[w] = tg; }
                            }
                        }
                        {
                            const uint32_t bk = __umulhi(x[u] - origin, scale);
                            extra_lo = st + MR_W;
                            extra_hi = dir[bk + 1] & 0x7FFFu;
                            for (uint32_t e = extra_lo; e < extra_hi; e++) {
                                if (pfx[e] == x[u]) {
                                    const uint32_t tg = tag[e];
                                    const uint32_t r = tg >> idx_bits, idx = (tg & idx_mask) + lo_of(tg >> idx_bits);
                                    const uint64_t v = a.row_hashes[(uint64_t)hdr->row_id[r] * a.row_stride + idx];
                                    if (v == b) rowmask |= 1u << r;
                                }
                            }
                        }
                    }
                    // rows involved anywhere in this block
                    uint32_t rows_any = 0;
                    for (uint32_t r = 0; r < R; r++)
                        if (__ballot((rowmask >> r) & 1u) != 0) rows_any |= 1u << r;
                    rows_any &= active;
                    while (rows_any != 0) {
                        const uint32_t r = (uint32_t)__builtin_ctz(rows_any);
                        rows_any &= rows_any - 1;
                        const bool mt = (rowmask >> r) & 1u;
                        // index of b in row r
                        uint32_t idx = 0;
                        bool have = false;
#pragma unroll
                        for (int w = 0; w < MR_W; w++) {
                            if (hit_tag[w] != 0xFFFFFFFFu && (hit_tag[w] >> idx_bits) == r) { idx = (hit_tag[w] & idx_mask) + lo_of(r); have = true; }
                        }
                        if (mt && !have) {
                            // hit came from beyond the window of an oversize bucket (rare): rescan
                            for (uint32_t e = extra_lo; e < extra_hi; e++) {
                                const uint32_t tg = tag[e];
                                if (pfx[e] == x[u] && (tg >> idx_bits) == r) {
                                    const uint64_t v = a.row_hashes[(uint64_t)hdr->row_id[r] * a.row_stride + lo_of(r) + (tg & idx_mask)];
                                    if (v == b) idx = lo_of(r) + (tg & idx_mask);
                                }
                            }
                        }
                        uint32_t c_all = (uint32_t)__builtin_amdgcn_readlane((int)st_call, (int)r);
                        uint32_t common = (uint32_t)__builtin_amdgcn_readlane((int)st_common, (int)r);
                        const uint64_t mm = __ballot(mt);
                        const uint32_t before = c_all + __builtin_amdgcn_mbcnt_hi(
                            (uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0));
                        const uint32_t rank = qb + lane + idx - before;
                        common += (uint32_t)__popcll(__ballot(mt && rank < s));
                        c_all += (uint32_t)__popcll(mm);
                        st_call = (lane == r) ? c_all : st_call;
                        st_common = (lane == r) ? common : st_common;
                        c_changed = true;
                    }
                }
            }

            // ---- rank test per row: has the union reached s elements at the group's last element?
            if (!col_end) {
                if (!WIN) {
                    if (c_changed) {                                     // uniform; counts moved: re-read operand
                        t_chk = (int32_t)s - 1 - (int32_t)qlast + (int32_t)st_call;
                        a_chk = my_row[t_chk >= 1 ? (uint32_t)t_chk - 1 : 0];
                    }
                    // prefix compare is conservative: prefix(A) < prefix(B) => A < B (a later exit is harmless)
                    const uint32_t xlast = (uint32_t)__builtin_amdgcn_readlane((int)cur[MR_KU - 1], 63);
                    const bool done = lane < R && (t_chk <= 0 || ((uint32_t)t_chk <= my_n && a_chk < xlast));
                    const uint32_t dm = (uint32_t)__ballot(done) & active;
                    active &= ~dm;
                    brokem |= dm;
                }
                // advance the pipeline: data of group g+1 becomes current; issue operand of g+1, data of g+2
#pragma unroll
                for (int u = 0; u < MR_KU; u++) cur[u] = nxt[u];
                if (!WIN) {
                    t_chk = (int32_t)s - 1 - (int32_t)(qlast + 64 * MR_KU) + (int32_t)st_call;
                    a_chk = my_row[t_chk >= 1 ? (uint32_t)t_chk - 1 : 0];
                }
                load_group(bsrc, q0 + 2 * 64 * MR_KU, nxt);
            }
        }
        bool prog = false;                                                   // WIN: my pair stays in progress
        if (lane < R) {
            uint32_t denom = s;
            if (!((brokem >> lane) & 1u)) {
                const uint32_t uni = my_n + nB - st_call;
                denom = uni < s ? uni : s;
                if (WIN) {
                    // Decided exactly at the end of the window's part: all union elements below the
                    // window's end are known -- my_pend of the row, qhi of the column, st_call shared.
                    // If they number s or more, the first s union elements (and every match among them,
                    // counted with its rank) lie behind us: denom = s.  Else the pair goes on to the
                    // next window unless the row or the column is exhausted (nothing left that could
                    // match: denom = |union| capped at s) or this was the last window.
                    if (!((started >> lane) & 1u)) denom = fin_denom;                      // decided earlier (or not ours)
                    else if (my_pend + qhi - st_call >= s) denom = s;
                    else if (a.win + 1 < a.nwin && qhi < nB && my_pend < my_n) { denom = 0x80000000u | st_call; prog = true; }
                }
            }
            if (pack) stage_p[lane * MR_CB + (tcol % MR_CB)] = (st_common & 0xFFFFu) | (((denom & 0x80000000u) ? (0x8000u | (denom & 0x7FFFu)) : denom) << 16);
            else stage[lane * MR_CB + (tcol % MR_CB)] = make_uint2(st_common, denom);
        }
        if (WIN) {
            procmask |= 1u << (tcol % MR_CB);
            if (__ballot(prog) != 0) progmask |= 1u << (tcol % MR_CB);
        }
        if (t1 / MR_CB != tcol / MR_CB || col_of(t1) >= tile.col1) {
            flush_batch(j - (tcol % MR_CB), (tcol % MR_CB) + 1, procmask);
            if (WIN) {
                if (lane == 0) wmask[tcol / MR_CB] = (uint8_t)progmask;
                procmask = progmask = 0;
            }
        }
        tcol = t1;
        t1 = t2;
        t2 = next_col(t2 + 1);
        j = col_of(tcol);
    }
#ifdef MASHGPU_TILE_CLOCKS
    if (a.dbg) {
        if (lane == 0) atomicMax(&a.dbg[3 * (uint64_t)blockIdx.x + 2], (unsigned long long)__builtin_readcyclecounter());
    }
#endif
}

__global__ void table_max_kernel(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t stride,
                                 unsigned long long *out_max)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    if (i < n) {
        uint64_t k = nhash[i];
        if (k > stride) k = stride;
        if (k > 0) v = hashes[i * stride + k - 1];
    }
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned long long o = __shfl_xor(v, d);
        v = o > v ? o : v;
    }
    if ((threadIdx.x & 63) == 0 && v) atomicMax(out_max, v);
}

// u32 prefix image with a padded row stride: entries beyond a row's valid hashes (and the
// padding columns) hold 0xFFFFFFFF, which no real prefix equals (the shift keeps real prefixes
// <= 0xFFFFFFFD), so the compare kernel needs neither index clamps nor in-range masks.
__global__ void make_prefix_kernel(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                                   uint64_t pfx_stride, uint32_t shr, uint32_t *out)
{
    const uint64_t total = n * pfx_stride;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const uint64_t i = e / pfx_stride, j = e - i * pfx_stride;
        uint32_t k = nhash[i];
        if (k > s) k = (uint32_t)s;
        out[e] = j < k ? mr_prefix(hashes[i * s + j], shr) : 0xFFFFFFFFu;
    }
}

hipError_t launch_table_max(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t stride,
                            unsigned long long *out_max, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(table_max_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, hashes, nhash, n,
                       stride, out_max);
    return hipGetLastError();
}

__global__ void row_classes_kernel(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t stride,
                                   uint8_t *out, unsigned long long *last_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k = nhash[i];
    if (k > stride) k = stride;
    uint8_t c = 0;
    unsigned long long last = 0;
    if (k > 0) {
        last = hashes[i * stride + k - 1];
        const uint64_t gap = last / k;                                   // mean spacing of the row's hashes
        c = (uint8_t)(64 - __clzll((long long)(gap | 1ull)));
    }
    out[i] = c;
    last_out[i] = last;
}

hipError_t launch_row_classes(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t stride,
                              uint8_t *out, unsigned long long *last_out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(row_classes_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, hashes, nhash, n,
                       stride, out, last_out);
    return hipGetLastError();
}

uint64_t compare_pfx_stride(uint64_t s)
{
    // every group the pipeline may touch lies inside the row: round s up to 64 and add
    // two groups of the largest KU variant (+1 block)
    return ((s + 63) / 64) * 64 + 64 * 13;             // covers KU <= 4: 64 (KU - 1) + 128 KU <= 832
}

hipError_t launch_make_prefix(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                              uint64_t pfx_stride, uint32_t shr, uint32_t *out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n * pfx_stride + 1023) / 1024;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(make_prefix_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, hashes, nhash, n, s,
                       pfx_stride, shr, out);
    return hipGetLastError();
}

// First index of every row at or above each window boundary (boundary w = w * delta for w < nwin,
// boundary nwin = end of the row): out[i * (nwin + 1) + w].  One thread per (row, boundary).
__global__ void window_offsets_kernel(const uint32_t *pfx, uint64_t pfx_stride, const uint32_t *nhash, uint64_t n, uint32_t s,
                                      uint32_t nwin, uint32_t delta, uint32_t *out)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * (nwin + 1)) return;
    const uint64_t i = t / (nwin + 1);
    const uint32_t w = (uint32_t)(t - i * (nwin + 1));
    uint32_t len = nhash[i];
    if (len > s) len = s;
    uint32_t lo = 0, hi = len;
    if (w < nwin) {
        const uint64_t bound = (uint64_t)w * delta;
        const uint32_t *row = pfx + i * pfx_stride;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint64_t)row[mid] < bound) lo = mid + 1; else hi = mid;
        }
    } else {
        lo = len;
    }
    out[t] = lo;
}

hipError_t launch_window_offsets(const uint32_t *pfx, uint64_t pfx_stride, const uint32_t *nhash, uint64_t n, uint32_t s,
                                 uint32_t nwin, uint32_t delta, uint32_t *out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint64_t total = n * (nwin + 1);
    hipLaunchKernelGGL(window_offsets_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, pfx, pfx_stride,
                       nhash, n, s, nwin, delta, out);
    return hipGetLastError();
}

// rows a window tile may list: 32 while results can be staged as u16 pairs, else 16
uint32_t compare_window_rows(uint32_t s) { return s < 32768u ? MR_WIN_ROWS : MR_WIN_ROWS / 2; }
uint32_t compare_window_entries() { return MR_WIN_ENTRIES; }
uint32_t compare_window_row_entries() { return (1u << MR_WIN_IDX_BITS) - 1u; }

template <int KU, bool WIN = false, bool W0 = false>
static hipError_t launch_merged_k(const CompareArgs &a, uint32_t ntiles, size_t smem, hipStream_t stream)
{
    auto kern = compare_merged_kernel<KU, WIN, W0>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(MR_NT), smem, stream, a);
    return hipGetLastError();
}

hipError_t launch_compare_merged(const CompareArgs &a_in, uint32_t ntiles, hipStream_t stream)
{
    if (ntiles == 0) return hipSuccess;
    CompareArgs a = a_in;
    if (a.nwin > 0) {
        // window mode: fixed tile geometry (up to 32 rows, MR_WIN_ENTRIES entries, MR_WIN_BUCKETS buckets)
        a.stage_pack = a.s < 32768u ? 1u : 0u;
        a.rows_per_tile = compare_window_rows(a.s);
        a.nbuckets = MR_WIN_BUCKETS;
        a.win_ecap = MR_WIN_ENTRIES;
        const size_t wsmem = merged_lds_bytes_e(MR_WIN_ROWS / 2, MR_WIN_ENTRIES, MR_WIN_BUCKETS);
        if (a.win == 0) {
            switch (a.unroll ? (int)a.unroll : 3) {
                case 2: return launch_merged_k<2, true, true>(a, ntiles, wsmem, stream);
                case 4: return launch_merged_k<4, true, true>(a, ntiles, wsmem, stream);
                default: return launch_merged_k<3, true, true>(a, ntiles, wsmem, stream);
            }
        }
        switch (a.unroll ? (int)a.unroll : 3) {
            case 2: return launch_merged_k<2, true>(a, ntiles, wsmem, stream);
            case 4: return launch_merged_k<4, true>(a, ntiles, wsmem, stream);
            default: return launch_merged_k<3, true>(a, ntiles, wsmem, stream);
        }
    }
    a.nbuckets = merged_buckets(a.rows_per_tile, a.s);
    const size_t smem = merged_lds_bytes_nb(a.rows_per_tile, a.s, a.nbuckets);
    // a.unroll (MASHGPU_COMPARE_VARIANT) selects the group size for tuning runs; 0 = default
    switch (a.unroll ? (int)a.unroll : MR_KU_DEFAULT) {
        case 2: return launch_merged_k<2>(a, ntiles, smem, stream);
        case 3: return launch_merged_k<3>(a, ntiles, smem, stream);
        case 4: return launch_merged_k<4>(a, ntiles, smem, stream);
        default: return launch_merged_k<MR_KU_DEFAULT>(a, ntiles, smem, stream);
    }
}

}  // namespace mg
