// compare_direct.hip — gfx950 pairwise comparison, window tiles over a DIRECT-MAPPED key table.
//
// Same contract and the same tile / window / carried-state machinery as compare_merged.hip's window
// mode (the merge loop of compareSketches, CommandDistance.cpp:347-385, in its rank formulation);
// what differs is the table a column element is probed in, because that probe is what the merged
// kernel is bound by: 1 directory gather + 4 entry gathers per element at ~9 LDS cycles each.
//
//   keys   : every DISTINCT value of the tile's rows (inside the launch's value window) is one key:
//            a 4-byte slot {18-bit fingerprint = low bits of the prefix, 14-bit link};
//   bucket : 4 slots = 16 bytes, home bucket = mulhi(prefix - window origin, scale); a key that does
//            not fit goes to the next bucket with a free slot (slots of a bucket fill in order, so
//            "slot 3 taken" = full).  ONE ds_read_b128 answers "is this fingerprint here"; only
//            the lanes that see a full bucket without a match read on;
//   tags   : per key the list of its (row, index-in-window) members, SORTED BY ROW, contiguous,
//            the last one flagged: {last:1, row:5, idx:10}.  The member of row r sits at
//            link + popcount(row mask below r): no scan of a bucket, no register packing.
//
// A fingerprint match is always verified on the 64-bit value of the key's first member before it
// counts (two keys of one bucket may share a fingerprint), so there is no "clean tile" notion.
//
// Build (per tile, ~1 % of its time): entries are histogrammed and scattered by bucket into a
// scratch region in HBM (L2 resident; LDS cannot hold the unsorted entries next to the final
// table), then one thread per bucket sorts its few entries by (prefix, row), splits equal
// prefixes by value, writes the tag lists and claims slots with LDS compare-and-swap.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "compare_internal.h"

namespace mg {

constexpr int DR_NT = 1024;
constexpr int DR_NW = DR_NT / 64;
constexpr int DR_CB = 8;                        // consecutive columns per wave batch
constexpr int DR_EPT = 16;                      // table entries handled per thread while building
constexpr uint32_t DR_ROWS = 32;
constexpr uint32_t DR_ENTRIES = 16000;
constexpr uint32_t DR_NBK = 6912;               // buckets (a key that finds its home bucket full goes on, cyclically)
constexpr uint32_t DR_SENT = DR_NBK;            // one more, always empty: elements outside the window
constexpr uint32_t DR_BUCKETS = DR_SENT + 1;
constexpr uint32_t DR_IDX_BITS = 10;
constexpr uint32_t DR_IDX_MASK = (1u << DR_IDX_BITS) - 1u;
constexpr uint32_t DR_LAST = 0x8000u;
constexpr uint32_t DR_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t DR_LINK_BITS = 14;
constexpr uint32_t DR_LINK_MASK = (1u << DR_LINK_BITS) - 1u;
constexpr uint32_t DR_FP_MASK = 0x3FFFFu;

constexpr size_t DR_OFF_SLOTS = 512;
constexpr size_t DR_OFF_TAGS = DR_OFF_SLOTS + (size_t)DR_BUCKETS * 16;
constexpr size_t DR_OFF_STAGE = DR_OFF_TAGS + (((size_t)DR_ENTRIES + 16) * 2 + 15) / 16 * 16;
constexpr size_t DR_LDS_BYTES = DR_OFF_STAGE + (size_t)DR_NW * DR_ROWS * DR_CB * 4;
static_assert(DR_LDS_BYTES <= 160 * 1024 - 640, "direct tile exceeds LDS");
static_assert(((size_t)DR_NBK + 2) * 2 <= (size_t)DR_NW * DR_ROWS * DR_CB * 4, "bucket offsets must fit the staging area");
static_assert(DR_ENTRIES <= DR_NT * DR_EPT && DR_ENTRIES <= (1u << DR_LINK_BITS), "entry capacity");

struct DirectHdr {
    uint32_t fail;       // a key found no slot (cannot happen within the capacity the host plans for)
    uint32_t scale;
    uint32_t nent;
    uint32_t slot;       // scratch region of this workgroup
    uint32_t row_n[32];
    uint32_t row_base[32];
    uint32_t row_id[32];
};

bool compare_direct_supported(uint32_t s) { return s >= 1 && s < 32768u; }
uint32_t compare_direct_rows() { return DR_ROWS; }
uint32_t compare_direct_entries() { return DR_ENTRIES; }
uint32_t compare_direct_row_entries() { return DR_IDX_MASK; }


__device__ __forceinline__ uint32_t dr_fp(uint32_t x)
{
    const uint32_t f = x & DR_FP_MASK;
    return f == DR_FP_MASK ? DR_FP_MASK - 1u : f;          // 0x3FFFF is what an empty slot shows
}

template <int KU, bool W0>
__global__ __launch_bounds__(DR_NT) void compare_direct_kernel(CompareArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t s = a.s;
    const uint32_t R = a.rows_per_tile;
    DirectHdr *hdr = reinterpret_cast<DirectHdr *>(smem);
    uint32_t *slots = reinterpret_cast<uint32_t *>(smem + DR_OFF_SLOTS);
    const uint4 *slots4 = reinterpret_cast<const uint4 *>(smem + DR_OFF_SLOTS);
    uint16_t *tags = reinterpret_cast<uint16_t *>(smem + DR_OFF_TAGS);
    unsigned char *stage_all = smem + DR_OFF_STAGE;
    uint16_t *boff = reinterpret_cast<uint16_t *>(stage_all);            // build-time view of the staging area: [NBK + 2]
    uint32_t *cnt32 = reinterpret_cast<uint32_t *>(stage_all);
    __shared__ uint32_t s_wsum[DR_NW + 2];
    __shared__ uint32_t s_rowlo[32], s_rowlen[32];

    const MergedTile *tile_p = a.mtiles + blockIdx.x;
    struct { uint32_t col0, col1; } tile = {tile_p->col0, tile_p->col1};
    const int tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);

    if (!W0) {
        // later windows: a tile none of whose columns still has a pair in progress has nothing to do
        const uint8_t *m = a.win_mask + (uint64_t)blockIdx.x * DR_NW * a.win_kmax;
        const uint32_t nbytes = DR_NW * a.win_kmax;
        uint32_t any = 0;
        for (uint32_t b = tid; b < nbytes; b += DR_NT) any |= m[b];
        if (__syncthreads_or((int)any) == 0) return;
    }
    // ------------------------------------------------------------------ rows of the tile
    if (tid < 32) {
        uint32_t n = 0, rid = 0xFFFFFFFFu, lo = 0, hi = 0;
        if ((uint32_t)tid < R) rid = tile_p->rows[tid];
        if (rid != 0xFFFFFFFFu) {
            n = a.row_nhash[rid];
            if (n > s) n = s;
            lo = a.row_win[(uint64_t)rid * (a.nwin + 1) + a.win];
            hi = a.row_win[(uint64_t)rid * (a.nwin + 1) + a.win + 1];
        }
        s_rowlo[tid] = lo;
        s_rowlen[tid] = n;
        hdr->row_n[tid] = hi - lo;                        // entries of this row inside the window
        hdr->row_id[tid] = rid;
    }
    for (uint32_t b = tid; b < DR_BUCKETS * 4; b += DR_NT) slots[b] = DR_EMPTY;
    for (uint32_t b = tid; b < (DR_NBK + 2 + 1) / 2; b += DR_NT) cnt32[b] = 0;
    __syncthreads();
    if (tid == 0) {
        uint32_t e = 0;
        for (uint32_t r = 0; r < 32; r++) {
            hdr->row_base[r] = e;
            e += hdr->row_n[r];
        }
        const uint64_t sc = ((uint64_t)DR_NBK << 32) / (uint64_t)(a.win_hi - a.win_lo);
        hdr->fail = 0;
        hdr->scale = sc > 0xFFFFFFFFULL ? 0xFFFFFFFFu : (uint32_t)sc;
        hdr->nent = e;
        // a scratch region of this workgroup's own (at most one workgroup per CU is resident)
        uint32_t i = blockIdx.x % a.dscr_regions;
        while (atomicCAS(&a.dscr_lock[i], 0u, 1u) != 0u) i = i + 1 < a.dscr_regions ? i + 1 : 0;
        hdr->slot = i;
    }
    __syncthreads();
    const uint32_t scale = hdr->scale, E = hdr->nent, origin = a.win_lo;
    volatile uint32_t *tpfx = a.dscr_pfx + (size_t)hdr->slot * (DR_NT * DR_EPT);
    volatile uint16_t *ttag = a.dscr_tag + (size_t)hdr->slot * (DR_NT * DR_EPT);

    // pass 1: bucket histogram (entries enumerated row-major over the rows' window ranges)
    {
        uint32_t e_pfx[DR_EPT], e_bs[DR_EPT];
        uint16_t e_tag[DR_EPT];
#pragma unroll
        for (int t = 0; t < DR_EPT; t++) {
            const uint32_t e = (uint32_t)tid + (uint32_t)t * DR_NT;
            e_bs[t] = 0xFFFFFFFFu;
            e_pfx[t] = 0;
            e_tag[t] = 0;
            if (e < E) {
                uint32_t r = 0;                            // last row whose base is <= e (row_base is non-decreasing)
#pragma unroll
                for (uint32_t step = 16; step >= 1; step >>= 1) r += hdr->row_base[r + step] <= e ? step : 0u;
                const uint32_t idx = e - hdr->row_base[r];
                const uint32_t x = a.row_pfx[(uint64_t)hdr->row_id[r] * a.row_pfx_stride + s_rowlo[r] + idx];
                uint32_t bk = __umulhi(x - origin, scale);
                if (bk >= DR_NBK) bk = DR_NBK - 1;
                const uint32_t old = atomicAdd(&cnt32[bk >> 1], (bk & 1u) ? 0x10000u : 1u);
                const uint32_t pos = (bk & 1u) ? (old >> 16) : (old & 0xFFFFu);
                e_pfx[t] = x;
                e_tag[t] = (uint16_t)((r << DR_IDX_BITS) | idx);
                e_bs[t] = bk | (pos << 16);
            }
        }
        __syncthreads();
        // exclusive scan of the u16 counters -> boff
        {
            const uint32_t per = (DR_NBK + DR_NT - 1) / DR_NT;
            const uint32_t b0 = tid * per;
            uint32_t sum = 0;
            for (uint32_t b = b0; b < b0 + per && b < DR_NBK; b++) sum += boff[b];
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
            for (uint32_t b = b0; b < b0 + per && b < DR_NBK; b++) {
                const uint32_t c = boff[b];
                boff[b] = (uint16_t)run;
                run += c;
            }
            if (tid == DR_NT - 1) { boff[DR_NBK] = (uint16_t)E; boff[DR_NBK + 1] = (uint16_t)E; }
        }
        __syncthreads();
        // pass 2: scatter to the scratch region, bucket by bucket
#pragma unroll
        for (int t = 0; t < DR_EPT; t++) {
            if (e_bs[t] != 0xFFFFFFFFu) {
                const uint32_t pos = (uint32_t)boff[e_bs[t] & 0xFFFFu] + (e_bs[t] >> 16);
                tpfx[pos] = e_pfx[t];
                ttag[pos] = e_tag[t];
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    // pass 3, one thread per bucket: entries sorted by (prefix, row); equal prefixes split by their
    // 64-bit values; per key the row-sorted tag list and one slot
    for (uint32_t b = tid; b < DR_NBK; b += DR_NT) {
        const uint32_t st = boff[b], c = (uint32_t)boff[b + 1] - st;
        if (c == 0) continue;
        for (uint32_t i = 1; i < c; i++) {                 // insertion sort (buckets hold a few entries)
            const uint32_t px = tpfx[st + i];
            const uint16_t tg = ttag[st + i];
            uint32_t k = i;
            while (k > 0) {
                const uint32_t pp = tpfx[st + k - 1];
                const uint16_t tt = ttag[st + k - 1];
                if (pp < px || (pp == px && tt < tg)) break;   // tag order = (row, idx) order
                tpfx[st + k] = pp;
                ttag[st + k] = tt;
                k--;
            }
            tpfx[st + k] = px;
            ttag[st + k] = tg;
        }
        uint32_t i = 0;
        while (i < c) {
            const uint32_t px = tpfx[st + i];
            uint32_t jn = i + 1;
            while (jn < c && tpfx[st + jn] == px) jn++;
            if (jn - i > 1) {
                // same prefix: one key only if the values are equal -- members with the first member's
                // value stay (in row order), the others move behind them and form the next group
                const uint32_t t0 = ttag[st + i];
                const uint64_t v0 = a.row_hashes[(uint64_t)hdr->row_id[t0 >> DR_IDX_BITS] * a.row_stride + s_rowlo[t0 >> DR_IDX_BITS] + (t0 & DR_IDX_MASK)];
                uint32_t keep = i + 1;
                for (uint32_t m = i + 1; m < jn; m++) {
                    const uint16_t tm = ttag[st + m];
                    const uint64_t vm = a.row_hashes[(uint64_t)hdr->row_id[tm >> DR_IDX_BITS] * a.row_stride + s_rowlo[tm >> DR_IDX_BITS] + (tm & DR_IDX_MASK)];
                    if (vm == v0) {
                        // rotate tm down to position `keep` (keeps both parts in row order)
                        for (uint32_t z = m; z > keep; z--) ttag[st + z] = ttag[st + z - 1];
                        ttag[st + keep] = tm;
                        keep++;
                    }
                }
                jn = keep;
            }
            for (uint32_t m = i; m < jn; m++) tags[st + m] = (uint16_t)((uint32_t)ttag[st + m] | (m + 1 == jn ? DR_LAST : 0u));
            const uint32_t val = (dr_fp(px) << DR_LINK_BITS) | (st + i);
            uint32_t bb = b;
            bool placed = false;
            for (uint32_t tries = 0; tries < DR_NBK && !placed; tries++) {     // 4 * NBK slots >= entries: always ends
                for (uint32_t w = 0; w < 4 && !placed; w++) placed = atomicCAS(&slots[bb * 4 + w], DR_EMPTY, val) == DR_EMPTY;
                bb = bb + 1 == DR_NBK ? 0 : bb + 1;
            }
            if (!placed) hdr->fail = 1;
            i = jn;
        }
    }
    __syncthreads();
    if (tid == 0) atomicExch(&a.dscr_lock[hdr->slot], 0u);          // the scratch region is free again
    if (hdr->fail) __builtin_trap();                                  // beyond the planned capacity: fail loudly
    __syncthreads();                                                  // boff's area becomes the output staging

    // ------------------------------------------------------------------ stream columns
    const uint32_t my_n = lane < 32 ? s_rowlen[lane] : 0;
    const uint32_t my_lo = lane < 32 ? s_rowlo[lane] : 0u;
    const uint32_t my_pend = lane < 32 ? my_lo + hdr->row_n[lane] : 0u;
    const uint32_t my_id = lane < 32 ? hdr->row_id[lane] : 0xFFFFFFFFu;
    const uint32_t my_lim = my_id == 0xFFFFFFFFu ? 0u : (a.triangle ? my_id : 0xFFFFFFFFu);
    uint64_t my_obase = 0;
    if (my_id != 0xFFFFFFFFu) {
        const uint64_t i = my_id;
        my_obase = a.triangle ? i * (i - 1) / 2 - a.out_base : (i - a.row_begin) * a.ncols;
    }
    auto load_group = [&](const uint32_t *src, uint32_t qbase, uint32_t (&dst)[KU]) {
        const uint32_t *gp = src + (uint32_t)__builtin_amdgcn_readfirstlane((int)qbase);
#pragma unroll
        for (int u = 0; u < KU; u++) dst[u] = gp[u * 64 + lane];
    };
    uint32_t *stage_p = reinterpret_cast<uint32_t *>(stage_all) + (size_t)wid * R * DR_CB;   // [R rows][CB], u16 pairs
    auto col_of = [&](uint32_t t) -> uint32_t { return tile.col0 + ((t / DR_CB) * DR_NW + wid) * DR_CB + (t % DR_CB); };
    auto flush_batch = [&](uint32_t jb, uint32_t procmask) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t c0 = (lane & 3u) * 2;
        for (uint32_t r = lane >> 2; r < ((R + 15u) & ~15u); r += 16) {
            const uint32_t lim = (uint32_t)__shfl((int)my_lim, (int)(r & 31u));
            const uint64_t obase = (uint64_t)(uint32_t)__shfl((int)(uint32_t)my_obase, (int)(r & 31u)) |
                                   ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(my_obase >> 32), (int)(r & 31u)) << 32);
            if (r < R && lim != 0) {
                const uint2 w = *reinterpret_cast<const uint2 *>(&stage_p[r * DR_CB + c0]);
                const uint32_t d0 = w.x >> 16, d1 = w.y >> 16;
                uint4 v;
                v.x = w.x & 0xFFFFu;
                v.y = (d0 & 0x8000u) ? (0x80000000u | (d0 & 0x7FFFu)) : d0;
                v.z = w.y & 0xFFFFu;
                v.w = (d1 & 0x8000u) ? (0x80000000u | (d1 & 0x7FFFu)) : d1;
                const uint64_t j0 = (uint64_t)jb + c0;
                const bool ok0 = ((procmask >> c0) & 1u) != 0 && j0 < lim;
                const bool ok1 = ((procmask >> (c0 + 1)) & 1u) != 0 && j0 + 1 < lim;
                uint2 *dst = a.out + (obase + j0);
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                if (ok0 && ok1 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                    u32x4 wv = {v.x, v.y, v.z, v.w};
                    __builtin_nontemporal_store(wv, reinterpret_cast<u32x4 *>(dst));
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
    uint32_t ncol[KU];
    uint32_t nB_next = 0;
    uint32_t nx_lo = 0, nx_hi = 0, n2_lo = 0, n2_hi = 0;
    uint2 st_next = make_uint2(0u, 0u);
    uint8_t *wmask = a.win_mask + ((uint64_t)blockIdx.x * DR_NW + wid) * a.win_kmax;
    uint32_t mk_lane = 0xFFu, mk_base = 0;
    auto mk_load = [&](uint32_t kb) {
        mk_base = kb;
        const uint32_t k = kb + lane;
        mk_lane = k < a.win_kmax ? (uint32_t)wmask[k] : 0u;
    };
    auto next_col = [&](uint32_t t) -> uint32_t {
        if (W0) return t;
        for (;;) {
            const uint32_t k = t / DR_CB;
            if (col_of(k * DR_CB) >= tile.col1) return t;
            if (k - mk_base >= 64u) mk_load(k & ~63u);
            const uint32_t rel = (uint32_t)__builtin_amdgcn_readfirstlane((int)(k - mk_base));
            const uint32_t m = ((uint32_t)__builtin_amdgcn_readlane((int)mk_lane, (int)rel) & 0xFFu) >> (t % DR_CB);
            if (m != 0) return t + (uint32_t)__builtin_ctz(m);
            uint64_t nz = __ballot(mk_lane != 0);
            nz = rel + 1 < 64u ? nz >> (rel + 1) : 0ull;
            t = nz != 0 ? (k + 1 + (uint32_t)__builtin_ctzll(nz)) * DR_CB : (mk_base + 64u) * DR_CB;
        }
    };
    auto bucket_of = [&](uint32_t x) -> uint32_t {
        const uint32_t bk = __umulhi(x - origin, scale);
        return bk < DR_NBK ? bk : DR_SENT;                 // prefixes outside the window (and padding) -> the empty bucket
    };
    if (!W0) mk_load(0);
    uint32_t procmask = 0, progmask = 0;
    uint32_t tcol = next_col(0);
    uint32_t t1 = next_col(tcol + 1), t2 = next_col(t1 + 1);
    uint32_t j = col_of(tcol);
    if (j < tile.col1) {
        win_range(j, nx_lo, nx_hi);
        const uint32_t j2 = col_of(t1);
        win_range(j2 < tile.col1 ? j2 : j, n2_lo, n2_hi);
        if (!W0) st_next = load_state(j);
        load_group(a.col_pfx + (uint64_t)j * a.col_pfx_stride, nx_lo, ncol);
        nB_next = a.col_nhash[j];
    }
    while (j < tile.col1) {
        const uint32_t nB = nB_next < s ? nB_next : s;
        const uint32_t *bsrc = a.col_pfx + (uint64_t)j * a.col_pfx_stride;
        uint32_t cur[KU], nxt[KU];
#pragma unroll
        for (int u = 0; u < KU; u++) cur[u] = ncol[u];
        const uint32_t valid = (uint32_t)__ballot(j < my_lim);
        uint32_t active = valid;
        uint32_t st_call = 0, st_common = 0;                             // lane r <-> row r
        const uint32_t qlo = nx_lo, qhi = nx_hi;                         // this launch's part of the column
        uint32_t fin_denom = 0;
        if (!W0) {
            const bool inprog = (st_next.y & 0x80000000u) != 0;
            st_common = st_next.x;
            st_call = inprog ? (st_next.y & 0x7FFFFFFFu) : 0u;
            fin_denom = st_next.y;
            active &= (uint32_t)__ballot(lane < R && inprog);
        }
        const uint32_t started = active;
        const uint32_t ngroups = active == 0 ? 0 : (qhi - qlo + 64 * KU - 1) / (64 * KU);
        load_group(bsrc, qlo + 64 * KU, nxt);
        {
            const uint32_t jnx = col_of(t1);
            const uint32_t jn = jnx < tile.col1 ? jnx : j;
            nx_lo = n2_lo;
            nx_hi = n2_hi;
            load_group(a.col_pfx + (uint64_t)jn * a.col_pfx_stride, nx_lo, ncol);
            nB_next = a.col_nhash[jn];
            if (!W0) st_next = load_state(jn);
            const uint32_t j2x = col_of(t2);
            win_range(j2x < tile.col1 ? j2x : jn, n2_lo, n2_hi);
        }
        for (uint32_t g = 0; g < ngroups; g++) {
            const uint32_t q0 = qlo + g * 64 * KU;
            const bool col_end = q0 + 64 * KU >= qhi;
            // ---- one probe per element for ALL rows of the tile: one 16-byte bucket ----
            uint32_t x[KU], fpx[KU], bk[KU];
            uint4 q4[KU];
            uint64_t tiem[KU];
            uint64_t anytie = 0, anymore = 0;
#pragma unroll
            for (int u = 0; u < KU; u++) {
                x[u] = cur[u];
                bk[u] = bucket_of(x[u]);
                q4[u] = slots4[bk[u]];
            }
#pragma unroll
            for (int u = 0; u < KU; u++) {
                fpx[u] = dr_fp(x[u]);
                const bool hit = ((q4[u].x >> DR_LINK_BITS) == fpx[u]) | ((q4[u].y >> DR_LINK_BITS) == fpx[u]) |
                                 ((q4[u].z >> DR_LINK_BITS) == fpx[u]) | ((q4[u].w >> DR_LINK_BITS) == fpx[u]);
                tiem[u] = __ballot(hit);
                anymore |= __ballot(!hit && q4[u].w != DR_EMPTY);       // bucket full, not found: the key may sit further on
                anytie |= tiem[u];
            }
            if (anymore != 0) {
#pragma unroll
                for (int u = 0; u < KU; u++) {
                    bool hit = false;
                    if (!((tiem[u] >> lane) & 1ULL) && q4[u].w != DR_EMPTY) {
                        uint32_t bb = bk[u] + 1 == DR_NBK ? 0 : bk[u] + 1;
                        for (uint32_t tries = 1; tries < DR_NBK; tries++) {
                            const uint4 qq = slots4[bb];
                            hit = ((qq.x >> DR_LINK_BITS) == fpx[u]) | ((qq.y >> DR_LINK_BITS) == fpx[u]) |
                                  ((qq.z >> DR_LINK_BITS) == fpx[u]) | ((qq.w >> DR_LINK_BITS) == fpx[u]);
                            if (hit || qq.w == DR_EMPTY) break;
                            bb = bb + 1 == DR_NBK ? 0 : bb + 1;
                        }
                    }
                    const uint64_t t = __ballot(hit);
                    tiem[u] |= t;
                    anytie |= t;
                }
            }
            if (anytie != 0) {
                // ---- exact path: the key of a tied element (verified on 64 bits), its rows, their ranks ----
#pragma unroll
                for (int u = 0; u < KU; u++) {
                    if (tiem[u] == 0) continue;                          // uniform
                    const uint32_t qb = q0 + u * 64;
                    const bool mine = (tiem[u] >> lane) & 1ULL;
                    uint32_t link = 0xFFFFFFFFu, rowmask = 0;
                    if (mine) {
                        const uint64_t b = a.col_hashes[(uint64_t)j * a.col_stride + qb + lane];
                        uint32_t bb = bk[u];
                        for (uint32_t tries = 0; tries < DR_NBK; tries++) {
                            const uint4 qq = slots4[bb];
                            const uint32_t sl[4] = {qq.x, qq.y, qq.z, qq.w};
#pragma unroll
                            for (int w = 0; w < 4; w++) {
                                if (link == 0xFFFFFFFFu && (sl[w] >> DR_LINK_BITS) == fpx[u]) {
                                    const uint32_t L = sl[w] & DR_LINK_MASK;
                                    const uint32_t tg = tags[L];
                                    const uint32_t r0 = (tg >> DR_IDX_BITS) & 31u;
                                    const uint64_t v = a.row_hashes[(uint64_t)hdr->row_id[r0] * a.row_stride + s_rowlo[r0] + (tg & DR_IDX_MASK)];
                                    if (v == b) link = L;
                                }
                            }
                            if (link != 0xFFFFFFFFu || qq.w == DR_EMPTY) break;
                            bb = bb + 1 == DR_NBK ? 0 : bb + 1;
                        }
                        if (link != 0xFFFFFFFFu) {
                            uint32_t e = link, tg;
                            do {
                                tg = tags[e++];
                                rowmask |= 1u << ((tg >> DR_IDX_BITS) & 31u);
                            } while (!(tg & DR_LAST));
                        }
                    }
                    uint32_t any = rowmask;
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) any |= __shfl_xor(any, d);
                    uint32_t rows_any = (uint32_t)__builtin_amdgcn_readfirstlane((int)any) & active;
                    while (rows_any != 0) {
                        const uint32_t r = (uint32_t)__builtin_ctz(rows_any);
                        rows_any &= rows_any - 1;
                        const bool mt = (rowmask >> r) & 1u;
                        uint32_t idx = 0;
                        if (mt) idx = ((uint32_t)tags[link + __popc(rowmask & ((1u << r) - 1u))] & DR_IDX_MASK) + s_rowlo[r];
                        uint32_t c_all = (uint32_t)__builtin_amdgcn_readlane((int)st_call, (int)r);
                        uint32_t common = (uint32_t)__builtin_amdgcn_readlane((int)st_common, (int)r);
                        const uint64_t mm = __ballot(mt);
                        const uint32_t before = c_all + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0));
                        const uint32_t rank = qb + lane + idx - before;
                        common += (uint32_t)__popcll(__ballot(mt && rank < s));
                        c_all += (uint32_t)__popcll(mm);
                        st_call = (lane == r) ? c_all : st_call;
                        st_common = (lane == r) ? common : st_common;
                    }
                }
            }
            if (!col_end) {
#pragma unroll
                for (int u = 0; u < KU; u++) cur[u] = nxt[u];
                load_group(bsrc, q0 + 2 * 64 * KU, nxt);
            }
        }
        bool prog = false;
        if (lane < R) {
            const uint32_t uni = my_n + nB - st_call;
            uint32_t denom = uni < s ? uni : s;
            // decided exactly at the end of the window's part (see compare_merged.hip)
            if (!((started >> lane) & 1u)) denom = fin_denom;
            else if (my_pend + qhi - st_call >= s) denom = s;
            else if (a.win + 1 < a.nwin && qhi < nB && my_pend < my_n) { denom = 0x80000000u | st_call; prog = true; }
            stage_p[lane * DR_CB + (tcol % DR_CB)] = (st_common & 0xFFFFu) | (((denom & 0x80000000u) ? (0x8000u | (denom & 0x7FFFu)) : denom) << 16);
        }
        procmask |= 1u << (tcol % DR_CB);
        if (__ballot(prog) != 0) progmask |= 1u << (tcol % DR_CB);
        if (t1 / DR_CB != tcol / DR_CB || col_of(t1) >= tile.col1) {
            flush_batch(j - (tcol % DR_CB), procmask);
            if (lane == 0) wmask[tcol / DR_CB] = (uint8_t)progmask;
            procmask = progmask = 0;
        }
        tcol = t1;
        t1 = t2;
        t2 = next_col(t2 + 1);
        j = col_of(tcol);
    }
}

template <int KU, bool W0>
static hipError_t launch_direct_k(const CompareArgs &a, uint32_t ntiles, hipStream_t stream)
{
    auto kern = compare_direct_kernel<KU, W0>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DR_LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(DR_NT), DR_LDS_BYTES, stream, a);
    return hipGetLastError();
}

hipError_t launch_compare_direct(const CompareArgs &a_in, uint32_t ntiles, hipStream_t stream)
{
    if (ntiles == 0) return hipSuccess;
    CompareArgs a = a_in;
    a.rows_per_tile = DR_ROWS;
    a.stage_pack = 1;
    const int ku = a.unroll ? (int)a.unroll : 3;
    if (a.win == 0) {
        switch (ku) {
            case 2: return launch_direct_k<2, true>(a, ntiles, stream);
            case 4: return launch_direct_k<4, true>(a, ntiles, stream);
            default: return launch_direct_k<3, true>(a, ntiles, stream);
        }
    }
    switch (ku) {
        case 2: return launch_direct_k<2, false>(a, ntiles, stream);
        case 4: return launch_direct_k<4, false>(a, ntiles, stream);
        default: return launch_direct_k<3, false>(a, ntiles, stream);
    }
}

}  // namespace mg
