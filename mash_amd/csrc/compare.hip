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
// Tiled kernel (s <= 1024): one 1024-thread workgroup owns R rows.  The rows live in
// LDS as sorted arrays + a 1024-bucket directory (bucket = value >> shift_row, the
// sorted array doubles as its own hash table: dir[bucket] is the lower bound of the
// bucket, a probe walks ~1-2 elements).  Each of the 16 waves streams whole columns
// from HBM/L2 straight into registers (lane l holds B[64k + l], fully coalesced 512-B
// loads) and probes every row: 64 lanes = 64 consecutive ranks per step, matches are
// ranked with one ballot + mbcnt.  All integer work: no MFMA; the limiters are LDS
// reads and VALU issue.  Algorithmic traffic: 2*s*8 + 8 B per pair.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "compare_internal.h"

namespace mg {

constexpr int CMP_NT = 1024;
constexpr int CMP_NW = CMP_NT / 64;
constexpr int CMP_NB = 1024;                 // directory buckets per row
constexpr int CMP_DIR = CMP_NB + 4;          // entries incl. dir[NB] = n, padded to 8 B
constexpr uint64_t HMAX = 0xFFFFFFFFFFFFFFFFULL;

struct RowMeta { uint32_t n; uint32_t shift; };

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63; }

// LDS bytes per row: (s + 1) values (one sentinel) + directory
__host__ __device__ inline size_t row_lds_bytes(uint32_t s)
{
    return (size_t)(s + 1) * 8 + (size_t)CMP_DIR * 2;
}

bool compare_tiled_supported(uint32_t s) { return s >= 1 && s <= 1024; }

uint32_t compare_rows_per_tile(uint32_t s)
{
    const size_t budget = 160 * 1024 - 64 * sizeof(RowMeta) - 64;
    size_t r = budget / row_lds_bytes(s);
    if (r > 64) r = 64;
    if (r > 16) r = 16;                      // more rows than waves buys nothing at s ~ 1000
    return (uint32_t)r;
}

template <int KITER>
__global__ __launch_bounds__(CMP_NT) void compare_tiled_kernel(CompareArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t s = a.s;
    const uint32_t R = a.rows_per_tile;
    const size_t rbytes = row_lds_bytes(s);
    RowMeta *meta = reinterpret_cast<RowMeta *>(smem);                    // [64]
    unsigned char *rows = smem + 64 * sizeof(RowMeta);

    const CompareTile tile = a.tiles[blockIdx.x];
    const int tid = threadIdx.x;
    const uint32_t lane = lane_id();
    const uint32_t wid = tid >> 6;

    // ---- stage R rows into LDS: values (+sentinels), then the bucket directory ----
    for (uint32_t r = 0; r < R; r++) {
        const uint64_t i = (uint64_t)tile.row0 + r;
        uint64_t *vals = reinterpret_cast<uint64_t *>(rows + r * rbytes);
        uint32_t n = 0;
        if (i < a.row_end) {
            n = a.row_nhash[i];
            if (n > s) n = s;
        }
        const uint64_t *src = a.row_hashes + i * a.row_stride;
        for (uint32_t p = tid; p <= s; p += CMP_NT) vals[p] = (p < n) ? src[p] : HMAX;
        if (tid == 0) {
            uint32_t sh = 0;
            if (n > 0) {
                const uint64_t mx = src[n - 1];
                const int bits = 64 - __clzll((unsigned long long)(mx | 1ULL));
                sh = bits > 10 ? (uint32_t)(bits - 10) : 0u;
            }
            meta[r].n = n;
            meta[r].shift = sh;
        }
    }
    __syncthreads();
    for (uint32_t r = 0; r < R; r++) {
        const uint64_t *vals = reinterpret_cast<const uint64_t *>(rows + r * rbytes);
        uint16_t *dir = reinterpret_cast<uint16_t *>(rows + r * rbytes + (size_t)(s + 1) * 8);
        const uint32_t n = meta[r].n, sh = meta[r].shift;
        for (uint32_t p = tid; p <= n; p += CMP_NT) {
            // element p opens buckets (bucket(p-1), bucket(p)]; p == n closes the tail
            const uint32_t lo = (p == 0) ? 0u : (uint32_t)(vals[p - 1] >> sh) + 1u;
            const uint32_t hi = (p == n) ? (uint32_t)CMP_NB : (uint32_t)(vals[p] >> sh);
            for (uint32_t b = lo; b <= hi && b <= (uint32_t)CMP_NB; b++) dir[b] = (uint16_t)p;
        }
    }
    __syncthreads();

    // ---- stream columns: wave w takes columns col0 + w, col0 + w + 16, ... ----
    for (uint32_t j = tile.col0 + wid; j < tile.col1; j += CMP_NW) {
        uint32_t nB = a.col_nhash[j];
        if (nB > s) nB = s;
        const uint64_t *bsrc = a.col_hashes + (uint64_t)j * a.col_stride;
        uint64_t bv[KITER];
#pragma unroll
        for (int k = 0; k < KITER; k++) {
            const uint32_t q = k * 64 + lane;
            bv[k] = (q < nB) ? bsrc[q] : HMAX;
        }
        for (uint32_t r = 0; r < R; r++) {
            const uint64_t i = (uint64_t)tile.row0 + r;
            if (i >= a.row_end) break;
            if (a.triangle && (uint64_t)j >= i) continue;
            const uint64_t *vals = reinterpret_cast<const uint64_t *>(rows + r * rbytes);
            const uint16_t *dir = reinterpret_cast<const uint16_t *>(rows + r * rbytes + (size_t)(s + 1) * 8);
            const uint32_t nA = meta[r].n, sh = meta[r].shift;
            uint32_t c_all = 0, common = 0;
            bool broke = false;
#pragma unroll
            for (int k = 0; k < KITER; k++) {
                if ((uint32_t)(k * 64) >= nB) break;                      // uniform
                const uint32_t q = k * 64 + lane;
                const uint64_t b = bv[k];
                const uint64_t bk64 = b >> sh;
                const uint32_t bk = bk64 > (uint64_t)CMP_NB ? (uint32_t)CMP_NB : (uint32_t)bk64;
                uint32_t p = dir[bk];
                uint64_t av = vals[p];
                while (av < b) { p++; av = vals[p]; }                     // sentinel-terminated
                const bool match = (av == b) && (q < nB) && (p < nA);
                const uint64_t m = __ballot(match);
                const uint32_t before = c_all + __builtin_amdgcn_mbcnt_hi(
                    (uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                const uint32_t rank = q + p - before;
                if ((uint32_t)__builtin_amdgcn_readfirstlane((int)rank) >= s) { broke = true; break; }
                const uint64_t mi = __ballot(match && rank < s);
                common += (uint32_t)__popcll(mi);
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

template <int KITER>
static hipError_t launch_tiled_k(const CompareArgs &a, uint32_t ntiles, hipStream_t stream)
{
    const size_t smem = 64 * sizeof(RowMeta) + (size_t)a.rows_per_tile * row_lds_bytes(a.s);
    auto kern = compare_tiled_kernel<KITER>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(CMP_NT), smem, stream, a);
    return hipGetLastError();
}

hipError_t launch_compare_tiled(const CompareArgs &a, uint32_t ntiles, hipStream_t stream)
{
    if (ntiles == 0) return hipSuccess;
    const uint32_t kiter = (a.s + 63) / 64;
    if (kiter <= 1) return launch_tiled_k<1>(a, ntiles, stream);
    if (kiter <= 2) return launch_tiled_k<2>(a, ntiles, stream);
    if (kiter <= 4) return launch_tiled_k<4>(a, ntiles, stream);
    if (kiter <= 8) return launch_tiled_k<8>(a, ntiles, stream);
    return launch_tiled_k<16>(a, ntiles, stream);
}

hipError_t launch_compare_generic(const CompareArgs &a, hipStream_t stream)
{
    const uint64_t nrows = a.row_end - a.row_begin;
    if (nrows == 0) return hipSuccess;
    hipLaunchKernelGGL(compare_generic_kernel, dim3((uint32_t)nrows), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace mg
