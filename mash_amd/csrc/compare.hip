// compare.hip — the GENERIC comparison kernel (one wavefront per pair, binary search in global memory, any
// sketch size: the independent cross-check of the two engines and the fallback where neither applies) and the
// distance filter + ordered compaction of the counts-only thresholded calls (filter_pass_kernel).
//
// What the reference computes per pair (A = row sketch, B = column sketch, both ascending and distinct,
// CommandDistance.cpp:347-385):   walk the sorted union, stop after s distinct elements;
//   numer = |A ∩ B ∩ bottom_s(A ∪ B)|,   denom = min(s, |A ∪ B|).
// Rank formulation used by the generic kernel (no sequential merge): an element b = B[q] with
// p = |{a in A : a < b}| and c = |{matches before b}| has 0-based rank q + p - c in the sorted union; it is
// counted iff it also occurs in A and its rank is < s.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "compare_internal.h"

namespace mg {

constexpr uint64_t HMAX = 0xFFFFFFFFFFFFFFFFULL;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63; }

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
