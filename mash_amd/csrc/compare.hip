// compare.hip — gfx950 pairwise comparison kernels: the merge loop of compareSketches
// (CommandDistance.cpp:347-385) for whole tiles of the (rows x columns) pair grid.
//
// What the reference computes per pair (A = row sketch, B = column sketch, both
// ascending and distinct):   walk the sorted union, stop after s distinct elements;
//   numer = |A ∩ B ∩ bottom_s(A ∪ B)|,   denom = min(s, |A ∪ B|).
// Equivalent rank formulation used here (no sequential merge): an element b = B[q]
// with p = |{a in A : a < b}| and c = |{matches before b}| has 0-based rank
// q + p - c in the sorted union; it is counted iff it also occurs in A and its rank
// is < s.  Ranks grow with q, so the scan over B stops at the first rank >= s (the
// reference's `denom < sketchSize` exit) — unrelated sketches cost about half a scan.
//
// Tiled kernel (s <= 1024): one 1024-thread workgroup owns R (=16 at s=1000) rows.  The
// rows live in LDS as sorted arrays split into a 32-bit prefix and the low 32 bits, plus
// a 1024-bucket directory (the sorted array doubles as its own hash table: dir[bucket]
// is the lower bound of the bucket; a probe reads a window of 4 prefixes).  Each of the
// 16 waves streams whole columns from HBM/L2 straight into registers (lane l holds
// B[64k + l], fully coalesced 512-B loads), two 64-element blocks at a time, and probes
// every row: 64 lanes = 64 consecutive ranks per step.  Unrelated sketches never tie on
// a prefix, so their whole cost is the fast path: bucket, directory read, window read,
// 4+4 compares, two scalar exit tests.  Ties (true matches) take an exact 64-bit path
// that ranks matches with one ballot + mbcnt.  All integer work: no MFMA; the limiter
// is VALU issue.  Algorithmic traffic: 2*s*8 + 8 B/pair.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "compare_internal.h"

namespace mg {

constexpr int CMP_NB = 1024;                 // directory buckets per row
constexpr int CMP_DIR = CMP_NB + 4;          // entries incl. dir[NB] = n, padded to 8 B
constexpr int CMP_W = 4;                     // probe window: elements read per probe
constexpr uint64_t HMAX = 0xFFFFFFFFFFFFFFFFULL;

// Row image in LDS.  Every value v of the tile's rows is split at a tile-wide bit
// position `shr` (chosen so the largest row value has a 32-bit prefix):
//     hi[p] = v >> shr   (32-bit prefix, non-decreasing in p)   lo[p] = (uint32) v
// Probes compare 32-bit prefixes only (full-rate VALU, 4-byte LDS reads); a probe whose
// prefix TIES with a stored prefix — every true match, and ~1e-7 of the others — is
// resolved exactly in a slow path on the reassembled 64-bit values.
// dir[bucket] = lower bound of the bucket, bucket = mulhi(prefix, scale_row) spreads
// the row's own values evenly over the NB buckets (~1 element per bucket at s = 1000).
struct RowMeta {
    uint32_t n;          // valid entries
    uint32_t xmax;       // prefix of the row's largest value; larger prefixes -> bucket NB
    uint32_t scale;      // floor(NB * 2^32 / (xmax + 1)), clamped to 2^32 - 1
    uint32_t _pad;
};

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ uint32_t prefix_of(uint64_t v, uint32_t shr)
{
    const uint64_t t = v >> shr;
    return (t >> 32) != 0 ? 0xFFFFFFFFu : (uint32_t)t;      // saturate (only for v beyond the tile's range)
}

__device__ __forceinline__ uint32_t row_bucket(uint32_t x, const RowMeta &m)
{
    const uint32_t bk = __umulhi(x, m.scale);
    return x > m.xmax ? (uint32_t)CMP_NB : bk;
}

// LDS bytes per row: (s + 1 + W) x {hi, lo}, then the directory
__host__ __device__ inline size_t row_lds_bytes(uint32_t s)
{
    return (size_t)(s + 1 + CMP_W) * 8 + (size_t)CMP_DIR * 2;
}

constexpr size_t CMP_HDR = 64 * sizeof(RowMeta) + 64 * 8 + 16;     // meta, row maxima, shr

bool compare_tiled_supported(uint32_t s) { return s >= 1 && s <= 1024; }

uint32_t compare_rows_per_tile(uint32_t s)
{
    const size_t budget = 160 * 1024 - CMP_HDR;
    size_t r = budget / row_lds_bytes(s);
    if (r > 16) r = 16;                      // more rows than waves buys nothing at s ~ 1000
    return (uint32_t)r;
}

template <int NT, int KU>
__global__ __launch_bounds__(NT) void compare_tiled_kernel(CompareArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t s = a.s;
    const uint32_t R = a.rows_per_tile;
    const size_t rbytes = row_lds_bytes(s);
    const uint32_t cnt_ent = s + 1 + CMP_W;                              // entries per hi/lo array
    RowMeta *meta = reinterpret_cast<RowMeta *>(smem);                    // [64]
    uint64_t *rowmax = reinterpret_cast<uint64_t *>(smem + 64 * sizeof(RowMeta));   // [64]
    unsigned char *rows = smem + CMP_HDR;

    const CompareTile tile = a.tiles[blockIdx.x];
    const int tid = threadIdx.x;
    const uint32_t lane = lane_id();
    // wave index: uniform, but the compiler cannot know — tell it, so everything derived
    // from the column index (row masks, loop control, row base addresses) stays scalar
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr uint32_t NW = NT / 64;

    // ---- stage R rows into LDS ----
    if (tid < (int)R) {
        const uint64_t i = (uint64_t)tile.row0 + tid;
        uint32_t n = 0;
        if (i < a.row_end) {
            n = a.row_nhash[i];
            if (n > s) n = s;
        }
        meta[tid].n = n;
        rowmax[tid] = n > 0 ? a.row_hashes[i * a.row_stride + n - 1] : 0;
    }
    __syncthreads();
    uint64_t tmax = 1;
    for (uint32_t r = 0; r < R; r++) tmax |= rowmax[r];
    const int tbl = 64 - __clzll((unsigned long long)tmax);               // bit length of the tile's values
    const uint32_t shr = tbl > 32 ? (uint32_t)(tbl - 32) : 0u;
    const uint64_t lomask = (1ULL << shr) - 1ULL;                         // shr <= 32
    for (uint32_t r = 0; r < R; r++) {
        const uint64_t i = (uint64_t)tile.row0 + r;
        uint32_t *hi = reinterpret_cast<uint32_t *>(rows + r * rbytes);
        uint32_t *lo = hi + cnt_ent;
        const uint32_t n = meta[r].n;
        const uint64_t *src = a.row_hashes + i * a.row_stride;
        for (uint32_t p = tid; p < cnt_ent; p += NT) {
            const uint64_t v = (p < n) ? src[p] : HMAX;
            hi[p] = (p < n) ? (uint32_t)(v >> shr) : 0xFFFFFFFFu;
            lo[p] = (uint32_t)v;
        }
        if (tid == 0) {
            const uint32_t xmax = (uint32_t)(rowmax[r] >> shr);
            const uint64_t sc = ((uint64_t)CMP_NB << 32) / ((uint64_t)xmax + 1ULL);
            meta[r].xmax = xmax;
            meta[r].scale = sc > 0xFFFFFFFFULL ? 0xFFFFFFFFu : (uint32_t)sc;
        }
    }
    __syncthreads();
    for (uint32_t r = 0; r < R; r++) {
        const uint32_t *hi = reinterpret_cast<const uint32_t *>(rows + r * rbytes);
        uint16_t *dir = reinterpret_cast<uint16_t *>(rows + r * rbytes + (size_t)cnt_ent * 8);
        const RowMeta m = meta[r];
        const uint32_t n = m.n;
        for (uint32_t p = tid; p <= n; p += NT) {
            // element p opens buckets (bucket(p-1), bucket(p)]; p == n closes the tail
            const uint32_t lob = (p == 0) ? 0u : row_bucket(hi[p - 1], m) + 1u;
            const uint32_t hib = (p == n) ? (uint32_t)CMP_NB : row_bucket(hi[p], m);
            for (uint32_t b = lob; b <= hib && b <= (uint32_t)CMP_NB; b++) dir[b] = (uint16_t)p;
        }
    }
    __syncthreads();

    // row meta of row `lane` stays in this lane's registers: the probe loop fetches it
    // with v_readlane (scalar operands, no LDS round trip in the dependent chain)
    const RowMeta my_meta = meta[lane];
    // ---- stream columns: wave w takes columns col0 + w, col0 + w + NW, ... ----
    // A column is consumed in groups of KU*64 elements (KU probes in flight per lane);
    // per-row running state lives in lane r of two VGPRs so only one group of the
    // column is register-resident, and rows that hit rank >= s drop out of later
    // groups (unrelated pairs never touch the upper half of the column).
    for (uint32_t j = tile.col0 + wid; j < tile.col1; j += NW) {
        uint32_t nB = a.col_nhash[j];
        if (nB > s) nB = s;
        const uint64_t *bsrc = a.col_hashes + (uint64_t)j * a.col_stride;
        uint32_t valid = 0;                                              // rows to compute (bitmask, uniform)
        for (uint32_t r = 0; r < R; r++) {
            const uint64_t i = (uint64_t)tile.row0 + r;
            if (i < a.row_end && (!a.triangle || (uint64_t)j < i)) valid |= 1u << r;
        }
        if (valid == 0) continue;
        uint32_t active = valid, brokem = 0;
        uint32_t st_call = 0, st_common = 0;                             // lane r <-> row r
        const uint32_t ngroups = (nB + 64 * KU - 1) / (64 * KU);
        uint64_t cur[KU], nxt[KU];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const uint32_t q = u * 64 + lane;
            cur[u] = (q < nB) ? bsrc[q] : HMAX;
        }
        for (uint32_t g = 0; g < ngroups && active != 0; g++) {
            const uint32_t q0 = g * 64 * KU;
            if (g + 1 < ngroups) {
#pragma unroll
                for (int u = 0; u < KU; u++) {
                    const uint32_t q = q0 + (KU + u) * 64 + lane;
                    nxt[u] = (q < nB) ? bsrc[q] : HMAX;
                }
            }
            uint32_t x[KU];
            uint64_t inbm[KU];                                           // lanes whose element exists (scalar masks)
#pragma unroll
            for (int u = 0; u < KU; u++) {
                x[u] = prefix_of(cur[u], shr);
                const uint32_t qb = q0 + u * 64;
                inbm[u] = qb + 64 <= nB ? ~0ULL : (qb >= nB ? 0ULL : ((1ULL << (nB - qb)) - 1ULL));
            }
            uint32_t todo = active;
            while (todo != 0) {
                const uint32_t r = (uint32_t)__builtin_ctz(todo);
                todo &= todo - 1;
                const uint32_t *hi = reinterpret_cast<const uint32_t *>(rows + r * rbytes);
                const uint32_t *lo = hi + cnt_ent;
                const uint16_t *dir = reinterpret_cast<const uint16_t *>(rows + r * rbytes + (size_t)cnt_ent * 8);
                RowMeta m;
                m.n = (uint32_t)__builtin_amdgcn_readlane((int)my_meta.n, (int)r);
                m.xmax = (uint32_t)__builtin_amdgcn_readlane((int)my_meta.xmax, (int)r);
                m.scale = (uint32_t)__builtin_amdgcn_readlane((int)my_meta.scale, (int)r);
                const uint32_t nA = m.n;
                // KU independent probes in flight: bucket -> dir -> window of W prefixes
                uint32_t p[KU];
#pragma unroll
                for (int u = 0; u < KU; u++) p[u] = dir[row_bucket(x[u], m)];
                uint32_t h[KU][CMP_W];
#pragma unroll
                for (int u = 0; u < KU; u++)
#pragma unroll
                    for (int w = 0; w < CMP_W; w++) h[u][w] = hi[p[u] + w];
                // tie / long-walk flags live in scalar lane masks (the compare results themselves)
                uint64_t tiem = 0, longm = 0;
#pragma unroll
                for (int u = 0; u < KU; u++) {
                    const uint32_t xu = x[u];
                    uint32_t cnt = 0;
                    uint64_t t = 0;
#pragma unroll
                    for (int w = 0; w < CMP_W; w++) {
                        cnt += h[u][w] < xu ? 1u : 0u;
                        t |= __ballot(h[u][w] == xu);
                    }
                    p[u] += cnt;
                    tiem |= t & inbm[u];
                    longm |= __ballot(cnt == (uint32_t)CMP_W) & inbm[u];
                }
                if (longm != 0) {
                    // some bucket holds more than W smaller prefixes: keep walking (prefix
                    // compares only, sentinel 0xFFFFFFFF / array end terminate the walk)
#pragma unroll
                    for (int u = 0; u < KU; u++) {
                        uint32_t pp = p[u];
                        uint32_t hv = hi[pp];
                        while (hv < x[u] && pp < cnt_ent - 1) { pp++; hv = hi[pp]; }
                        tiem |= __ballot(hv == x[u]) & inbm[u];
                        p[u] = pp;
                    }
                }
                bool broke = false;
                if (tiem == 0) {
                    // fast path: no element of this group occurs in the row and every lower
                    // bound is exact, so match counts stay put; only the exit tests remain
                    const uint32_t c_all = (uint32_t)__builtin_amdgcn_readlane((int)st_call, (int)r);
#pragma unroll
                    for (int u = 0; u < KU; u++) {
                        const uint32_t qb = q0 + u * 64;
                        if (qb >= nB || broke) break;
                        const uint32_t rank0 = qb + (uint32_t)__builtin_amdgcn_readfirstlane((int)p[u]) - c_all;
                        if (rank0 >= s) { broke = true; break; }
                        if (qb + 63 < nB) {
                            const uint32_t rank63 = qb + 63 + (uint32_t)__builtin_amdgcn_readlane((int)p[u], 63) - c_all;
                            if (rank63 + 1u >= s) broke = true;
                        }
                    }
                } else {
                    // exact path on reassembled 64-bit values
                    uint32_t c_all = (uint32_t)__builtin_amdgcn_readlane((int)st_call, (int)r);
                    uint32_t common = (uint32_t)__builtin_amdgcn_readlane((int)st_common, (int)r);
#pragma unroll
                    for (int u = 0; u < KU; u++) {
                        const uint32_t qb = q0 + u * 64;
                        if (qb >= nB || broke) break;                     // uniform
                        const uint64_t b = cur[u];
                        // restart from the bucket's lower bound (p[u] may have skipped ties)
                        uint32_t pp = dir[row_bucket(x[u], m)];
                        uint64_t av = 0;
                        while (pp < nA) {
                            av = ((uint64_t)hi[pp] << shr) | ((uint64_t)lo[pp] & lomask);
                            if (av >= b) break;
                            pp++;
                        }
                        const uint32_t q = qb + lane;
                        const bool mt = ((inbm[u] >> lane) & 1ULL) && (pp < nA) && (av == b);
                        const uint64_t mm = __ballot(mt);
                        const uint32_t before = c_all + __builtin_amdgcn_mbcnt_hi(
                            (uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0));
                        const uint32_t rank = q + pp - before;
                        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)rank) >= s) { broke = true; break; }
                        common += (uint32_t)__popcll(__ballot(mt && rank < s));
                        c_all += (uint32_t)__popcll(mm);
                        if (qb + 63 < nB && (uint32_t)__builtin_amdgcn_readlane((int)rank, 63) + 1u >= s) broke = true;
                    }
                    st_call = (lane == r) ? c_all : st_call;
                    st_common = (lane == r) ? common : st_common;
                }
                if (broke) { active &= ~(1u << r); brokem |= 1u << r; }
            }
#pragma unroll
            for (int u = 0; u < KU; u++) cur[u] = nxt[u];
        }
        if (lane < R && ((valid >> lane) & 1u)) {
            const uint64_t i = (uint64_t)tile.row0 + lane;
            uint32_t denom = s;
            if (!((brokem >> lane) & 1u)) {
                const uint32_t uni = my_meta.n + nB - st_call;
                denom = uni < s ? uni : s;
            }
            uint64_t oidx;
            if (a.triangle) oidx = i * (i - 1) / 2 + j - a.out_base;
            else oidx = (i - a.row_begin) * a.ncols + j;
            a.out[oidx] = make_uint2(st_common, denom);
        }
    }
}

// Generic kernel: one wave per pair, lower bounds by binary search in global memory.
// Workgroup = one row, its waves stride over the columns.
__global__ __launch_bounds__(256) void compare_generic_kernel(CompareArgs a)
{
    const uint64_t i = a.row_begin + blockIdx.x;
    if (i >= a.row_end) return;
    const uint32_t lane = lane_id();
    const uint32_t wid = threadIdx.x >> 6;
    const uint32_t s = a.s;
    uint32_t nA = a.row_nhash[i];
    if (nA > s) nA = s;
    const uint64_t *A = a.row_hashes + i * a.row_stride;
    const uint64_t ncols = a.triangle ? i : a.ncols;
    for (uint64_t j = wid; j < ncols; j += 4) {
        uint32_t nB = a.col_nhash[j];
        if (nB > s) nB = s;
        const uint64_t *B = a.col_hashes + j * a.col_stride;
        uint32_t c_all = 0, common = 0;
        bool broke = false;
        for (uint32_t k0 = 0; k0 < nB; k0 += 64) {
            const uint32_t q = k0 + lane;
            const uint64_t b = (q < nB) ? B[q] : HMAX;
            uint32_t lo = 0, hi = nA;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (A[mid] < b) lo = mid + 1; else hi = mid;
            }
            const uint32_t p = lo;
            const bool match = (q < nB) && (p < nA) && (A[p] == b);
            const uint64_t m = __ballot(match);
            const uint32_t before = c_all + __builtin_amdgcn_mbcnt_hi(
                (uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            const uint32_t rank = q + p - before;
            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)rank) >= s) { broke = true; break; }
            common += (uint32_t)__popcll(__ballot(match && rank < s));
            c_all += (uint32_t)__popcll(m);
        }
        uint32_t denom = s;
        if (!broke) {
            const uint32_t uni = nA + nB - c_all;
            denom = uni < s ? uni : s;
        }
        if (lane == 0) {
            uint64_t oidx;
            if (a.triangle) oidx = i * (i - 1) / 2 + j - a.out_base;
            else oidx = (i - a.row_begin) * a.ncols + j;
            a.out[oidx] = make_uint2(common, denom);
        }
    }
}

template <int NT, int KU>
static hipError_t launch_tiled_k(const CompareArgs &a, uint32_t ntiles, hipStream_t stream)
{
    const size_t smem = CMP_HDR + (size_t)a.rows_per_tile * row_lds_bytes(a.s);
    auto kern = compare_tiled_kernel<NT, KU>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(NT), smem, stream, a);
    return hipGetLastError();
}

// variant = NT*10 + KU (tuning knob MASHGPU_COMPARE_VARIANT); 0 = default
hipError_t launch_compare_tiled(const CompareArgs &a, uint32_t ntiles, hipStream_t stream)
{
    if (ntiles == 0) return hipSuccess;
    switch (a.unroll) {
        case 10242: return launch_tiled_k<1024, 2>(a, ntiles, stream);
        case 10244: return launch_tiled_k<1024, 4>(a, ntiles, stream);
        case 5122: return launch_tiled_k<512, 2>(a, ntiles, stream);
        case 5124: return launch_tiled_k<512, 4>(a, ntiles, stream);
        case 5128: return launch_tiled_k<512, 8>(a, ntiles, stream);
        default: break;
    }
    return launch_tiled_k<1024, 2>(a, ntiles, stream);      // best measured (profiles/r01_compare_sweep.txt)
}

hipError_t launch_compare_generic(const CompareArgs &a, hipStream_t stream)
{
    const uint64_t nrows = a.row_end - a.row_begin;
    if (nrows == 0) return hipSuccess;
    hipLaunchKernelGGL(compare_generic_kernel, dim3((uint32_t)nrows), dim3(256), 0, stream, a);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------
// Distance filter + ordered compaction over a block of {numer, denom} counts.
// The reference rejects a pair when distance > maxDistance (CommandDistance.cpp:409-412);
// distance is a decreasing function of numer for fixed denom, so the host turns the
// threshold into min_numer[denom] with the same libm it finishes with, and the
// device test is a pure integer compare: the surviving set is exactly the
// reference's.  Both output layouts (triangle rows, query-major rect) are
// row-major, so the flat index order IS the reference's output order: segments of
// FL_SEG pairs are counted (pass A), the segment counts are scanned, and pass B
// rewrites every survivor at its exact rank -- no atomics, no sort.
typedef uint32_t fl_u32x2 __attribute__((ext_vector_type(2)));
constexpr int FL_NT = 256;               // 4 waves, each owns FL_PER*64 consecutive pairs
constexpr int FL_PER = 16;
constexpr int FL_SEG = FL_NT * FL_PER;   // 4096 pairs = 32 KiB of counts per workgroup

template <bool WRITE>
__global__ __launch_bounds__(FL_NT) void filter_pass_kernel(FilterArgs a)
{
    __shared__ uint32_t wtot[FL_NT / 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t seg = blockIdx.x;
    const uint64_t base = seg * FL_SEG + (uint64_t)wave * (FL_PER * 64);
    uint2 v[FL_PER];
    unsigned long long m[FL_PER];
#pragma unroll
    for (int it = 0; it < FL_PER; it++) {
        const uint64_t idx = base + (uint64_t)it * 64 + lane;
        v[it] = make_uint2(0, 0xFFFFFFFFu);
        if (idx < a.pairs) {
            const fl_u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const fl_u32x2 *>(a.counts) + idx);
            v[it] = make_uint2(t.x, t.y);
        }
    }
    uint32_t total = 0;
#pragma unroll
    for (int it = 0; it < FL_PER; it++) {
        const bool pass = v[it].y <= a.s && v[it].x >= a.min_numer[v[it].y <= a.s ? v[it].y : 0];
        m[it] = __ballot(pass);
        total += (uint32_t)__popcll(m[it]);
    }
    if (lane == 0) wtot[wave] = total;
    __syncthreads();
    if (!WRITE) {
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int w = 0; w < FL_NT / 64; w++) t += wtot[w];
            a.seg_count[seg] = t;
        }
        return;
    }
    uint64_t pos = a.seg_off[seg];
    for (uint32_t w = 0; w < wave; w++) pos += wtot[w];
    if (pos >= a.win_lo + a.win_n || pos + total <= a.win_lo) return;   // whole wave outside the window
#pragma unroll
    for (int it = 0; it < FL_PER; it++) {
        const unsigned long long mm = m[it];
        if ((mm >> lane) & 1) {
            const uint64_t at = pos + __popcll(mm & ((1ull << lane) - 1));
            if (at >= a.win_lo && at - a.win_lo < a.win_n) {
                const uint64_t idx = base + (uint64_t)it * 64 + lane;
                uint64_t row, col;
                if (a.triangle) {
                    const uint64_t f = a.first_row;
                    const uint64_t g = (f ? f * (f - 1) / 2 : 0) + idx;       // index in the whole triangle
                    row = (uint64_t)((1.0 + sqrt(1.0 + 8.0 * (double)g)) * 0.5);
                    while (row * (row - 1) / 2 > g) row--;
                    while ((row + 1) * row / 2 <= g) row++;
                    col = g - row * (row - 1) / 2;
                } else {
                    row = a.first_row + idx / a.ncols;
                    col = idx % a.ncols;
                }
                a.edges[at - a.win_lo] = make_uint4((uint32_t)row, (uint32_t)col, v[it].x, v[it].y);
            }
        }
        pos += __popcll(mm);
    }
}

// exclusive scan of the segment counts (one workgroup; nseg <= a few 10^5) + grand total
__global__ __launch_bounds__(1024) void filter_scan_kernel(const uint32_t *seg_count, unsigned long long *seg_off,
                                                           uint64_t nseg, unsigned long long *total)
{
    __shared__ unsigned long long part[1024];
    const uint64_t per = (nseg + 1023) / 1024;
    const uint64_t b = threadIdx.x * per, e = b + per < nseg ? b + per : nseg;
    unsigned long long sum = 0;
    for (uint64_t i = b; i < e; i++) sum += seg_count[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long x = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += x;
        __syncthreads();
    }
    unsigned long long run = part[threadIdx.x] - sum;
    for (uint64_t i = b; i < e; i++) { seg_off[i] = run; run += seg_count[i]; }
    if (threadIdx.x == 1023) *total = part[1023];
}

uint64_t filter_segments(uint64_t pairs) { return (pairs + FL_SEG - 1) / FL_SEG; }

hipError_t launch_filter_count(const FilterArgs &a, unsigned long long *total, hipStream_t stream)
{
    const uint64_t nseg = filter_segments(a.pairs);
    if (nseg == 0 || nseg > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(filter_pass_kernel<false>, dim3((uint32_t)nseg), dim3(FL_NT), 0, stream, a);
    hipLaunchKernelGGL(filter_scan_kernel, dim3(1), dim3(1024), 0, stream, a.seg_count, a.seg_off, nseg, total);
    return hipGetLastError();
}

hipError_t launch_filter_write(const FilterArgs &a, hipStream_t stream)
{
    const uint64_t nseg = filter_segments(a.pairs);
    if (nseg == 0 || nseg > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(filter_pass_kernel<true>, dim3((uint32_t)nseg), dim3(FL_NT), 0, stream, a);
    return hipGetLastError();
}

}  // namespace mg
