// compare_pairs.hip — gfx950 pairwise comparison, one wavefront per pair by merge path.
//
// The engine for pairs that really share most of their hashes (same contract as the other
// compare kernels: the merge loop of compareSketches, CommandDistance.cpp:347-385).  The tile
// kernel (compare_merged.hip) answers "nothing in common" with one probe per column element for
// sixteen rows at once, but pays ~1000 instructions for a pair that matches everywhere; here a
// pair costs ~500 whatever it shares:
//
//   * the row A sits in LDS for the whole workgroup, every wave stages its column B next to it;
//   * the merged sequence of A and B (ties: A first) is cut into 64 equal segments by
//     merge-path partition (one binary search per lane), each lane walks its segment and counts
//     union elements (a B equal to the A just taken is the same element) and matches;
//   * a wave prefix sum of the union counts places every segment in the union; matches whose
//     union index is below min(s, |A u B|) are the reference's `common`, that minimum its `denom`.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "compare_internal.h"

namespace mg {

constexpr uint64_t PR_NONE = 0xFFFFFFFFFFFFFFFFULL;      // never a hash (MG_HASH_PAD)

bool compare_pairs_supported(uint32_t s) { return s >= 1 && (size_t)5 * (s + 1) * 8 <= 160 * 1024 - 256; }

__global__ __launch_bounds__(256) void compare_pairs_kernel(CompareArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t s = a.s;
    const uint32_t sp = s + 1;
    uint64_t *A = reinterpret_cast<uint64_t *>(smem);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint64_t *B = A + (size_t)sp * (1 + wid);
    const uint64_t i = a.row_begin + blockIdx.x;
    if (i >= a.row_end) return;
    uint32_t nA = a.row_nhash[i];
    if (nA > s) nA = s;
    {
        const uint64_t *row = a.row_hashes + i * a.row_stride;
        for (uint32_t t = tid; t < nA; t += 256) A[t] = row[t];
    }
    __syncthreads();
    const uint64_t ncols = a.triangle ? i : a.ncols;
    for (uint64_t j = wid; j < ncols; j += 4) {
        uint32_t nB = a.col_nhash[j];
        if (nB > s) nB = s;
        {
            const uint64_t *col = a.col_hashes + j * a.col_stride;
            __builtin_amdgcn_wave_barrier();                         // previous column fully consumed
            for (uint32_t t = lane; t < nB; t += 64) B[t] = col[t];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const uint32_t M = nA + nB;
        const uint32_t T = (M + 63) / 64;
        const uint32_t d0 = lane * T < M ? lane * T : M;
        const uint32_t d1 = d0 + T < M ? d0 + T : M;
        // merge-path partition of diagonal d0: ia = number of A elements among the first d0 merged
        uint32_t lo = d0 > nB ? d0 - nB : 0, hi = d0 < nA ? d0 : nA;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (A[mid] <= B[d0 - 1 - mid]) lo = mid + 1; else hi = mid;
        }
        const uint32_t ia0 = lo, ib0 = d0 - lo;
        auto walk = [&](uint32_t limit_union, uint32_t start_union, bool counting, uint32_t &u_out, uint32_t &c_out,
                        bool &dupfirst_out) {
            uint32_t ia = ia0, ib = ib0, u = 0, c = 0;
            uint64_t lastA = ia0 > 0 ? A[ia0 - 1] : PR_NONE;
            uint64_t ca = ia < nA ? A[ia] : PR_NONE, cb = ib < nB ? B[ib] : PR_NONE;
            bool dupfirst = false;
            for (uint32_t d = d0; d < d1; d++) {
                const bool take_a = ib >= nB || (ia < nA && ca <= cb);
                if (take_a) {
                    lastA = ca;
                    ia++;
                    ca = ia < nA ? A[ia] : PR_NONE;
                    u++;
                } else {
                    const bool dup = cb == lastA;
                    ib++;
                    if (dup) {
                        if (!counting || start_union + u <= limit_union) c++;   // union index start+u-1 < limit
                        if (d == d0) dupfirst = true;
                    } else {
                        u++;
                    }
                    cb = ib < nB ? B[ib] : PR_NONE;
                }
            }
            u_out = u; c_out = c; dupfirst_out = dupfirst;
        };
        uint32_t u = 0, c = 0;
        bool dupfirst = false;
        walk(0, 0, false, u, c, dupfirst);
        // exclusive prefix of the union counts
        uint32_t inc = u;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t t = __shfl_up(inc, dd);
            if (lane >= (uint32_t)dd) inc += t;
        }
        const uint32_t total_u = (uint32_t)__shfl((int)inc, 63);
        const uint32_t U = inc - u;
        const uint32_t denom = total_u < s ? total_u : s;
        uint32_t cnt;
        if (U + u <= denom) {
            cnt = c;                                                 // segment entirely inside the union prefix
        } else if (U >= denom) {
            cnt = (dupfirst && U == denom) ? 1u : 0u;                // only a duplicate of the previous lane's last A
        } else {
            uint32_t u2, c2;
            bool df2;
            walk(denom, U, true, u2, c2, df2);                       // the segment that crosses the cut
            cnt = c2;
        }
#pragma unroll
        for (int dd = 32; dd > 0; dd >>= 1) cnt += __shfl_xor(cnt, dd);
        if (lane == 0) {
            uint64_t oidx;
            if (a.triangle) oidx = i * (i - 1) / 2 + j - a.out_base;
            else oidx = (i - a.row_begin) * a.ncols + j;
            a.out[oidx] = make_uint2(cnt, denom);
        }
    }
}

hipError_t launch_compare_pairs(const CompareArgs &a, hipStream_t stream)
{
    const uint64_t nrows = a.row_end - a.row_begin;
    if (nrows == 0) return hipSuccess;
    const size_t smem = (size_t)5 * (a.s + 1) * 8;
    auto kern = compare_pairs_kernel;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((uint32_t)nrows), dim3(256), smem, stream, a);
    return hipGetLastError();
}

}  // namespace mg
