// finish.hip — the tail of compareSketches on the device (CommandDistance.cpp:387-424, pValue :427-448):
// {numer, denom} -> Mash distance, p-value, both filters (-d, -v), optional ordered compaction.
//
// distance: the reference computes -log(2j / (1 + j)) / k with the host libm; the device cannot
// reproduce glibc's log bit for bit, so distances come from a TABLE the host builds with that very
// libm: one row of denom + 1 doubles per denominator that occurs (lut_start[denom] -> first entry;
// denom = s is always there, others are tabulated on demand after a flag pass).  A denominator
// that is not tabulated (budget exceeded) yields NaN and the host patches the pair -- never a
// device log.
// p-value: mg::p_value of pvalue.h -- the exact double-double binomial tail, only + - * / fma
// frexp ldexp, identical code and identical bits on host and device.
// filters: distance via the integer table min_numer[denom] (exact, see filter_pass_kernel in
// compare.hip), then p-value > max_p rejects (:419-422); a pair rejected by the distance filter
// gets no p-value, as in the reference (:409-412 returns early).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "finish_internal.h"
#include "pvalue.h"

namespace mg {

constexpr int FN_NT = 256;
constexpr int FN_PER = 16;
constexpr int FN_SEG = FN_NT * FN_PER;       // pairs per workgroup of the mark / write passes

__device__ __forceinline__ void pair_rc(const FinishArgs &a, uint64_t idx, uint64_t &row, uint64_t &col)
{
    if (a.list_rc) {                                       // a list of pairs (the sparse engine's candidates), not a block of the matrix
        const uint2 rc = a.list_rc[idx];
        row = rc.x;
        col = rc.y;
        return;
    }
    if (a.triangle) {
        const uint64_t f = a.first_row;
        const uint64_t g = (f ? f * (f - 1) / 2 : 0) + idx;               // index in the whole triangle
        row = (uint64_t)((1.0 + sqrt(1.0 + 8.0 * (double)g)) * 0.5);
        while (row * (row - 1) / 2 > g) row--;
        while ((row + 1) * row / 2 <= g) row++;
        col = g - row * (row - 1) / 2;
    } else {
        row = a.first_row + idx / a.ncols;
        col = idx % a.ncols;
    }
}

__device__ __forceinline__ double lut_distance(const FinishArgs &a, uint32_t numer, uint32_t denom)
{
    if (numer == denom) return 0.0;                                       // CommandDistance.cpp:389-392
    if (numer == 0) return 1.0;                                           // :393-396
    const uint32_t st = denom <= a.s ? a.lut_start[denom] : 0xFFFFFFFFu;
    if (st == 0xFFFFFFFFu) return __builtin_nan("");                      // not tabulated: the host patches it
    return a.lut[(uint64_t)st + numer];
}

__device__ __forceinline__ bool passes_distance(const FinishArgs &a, uint32_t numer, uint32_t denom)
{
    if (!a.min_numer) return true;
    return denom <= a.s && numer >= a.min_numer[denom];
}

// every pair -> mg_pair (numer, denom, distance, p_value, pass), grid-stride
__global__ __launch_bounds__(256) void finish_pairs_kernel(FinishArgs a)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < a.pairs; idx += stride) {
        const uint2 c = a.counts[idx];
        FinishPair o;
        o.numer = c.x;
        o.denom = c.y;
        o.distance = lut_distance(a, c.x, c.y);
        o.p_value = 0.0;
        o.pass = 0;
        for (int b = 0; b < 7; b++) o._pad[b] = 0;                        // records are compared and hashed as bytes
        if (passes_distance(a, c.x, c.y)) {
            uint64_t row, col;
            pair_rc(a, idx, row, col);
            o.p_value = p_value(c.x, a.len_row[row], a.len_col[col], a.kmer_space, c.y);
            o.pass = (a.max_p < 0.0 || !(o.p_value > a.max_p)) ? 1u : 0u;
        }
        a.pairs_out[idx] = o;
    }
}

// pass A of the edge list: survivors of both filters as one bit per pair (ballots), their number
// per segment, and the denominators they carry (for the distance table of pass B)
__global__ __launch_bounds__(FN_NT) void finish_mark_kernel(FinishArgs a)
{
    __shared__ uint32_t wtot[FN_NT / 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t seg = blockIdx.x;
    const uint64_t base = seg * FN_SEG + (uint64_t)wave * (FN_PER * 64);
    uint32_t total = 0;
    for (int it = 0; it < FN_PER; it++) {
        const uint64_t idx = base + (uint64_t)it * 64 + lane;
        bool pass = false;
        if (idx < a.pairs) {
            const uint2 c = a.counts[idx];
            pass = passes_distance(a, c.x, c.y);
            if (pass && a.max_p >= 0.0 && a.max_p < 1.0) {
                uint64_t row, col;
                pair_rc(a, idx, row, col);
                pass = !(p_value(c.x, a.len_row[row], a.len_col[col], a.kmer_space, c.y) > a.max_p);
            }
            if (pass && c.y <= a.s) a.denom_seen[c.y] = 1u;
        }
        const unsigned long long m = __ballot(pass);
        if (lane == 0) a.masks[(seg * (FN_NT / 64) + wave) * FN_PER + it] = m;
        total += (uint32_t)__popcll(m);
    }
    if (lane == 0) wtot[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < FN_NT / 64; w++) t += wtot[w];
        a.seg_count[seg] = t;
    }
}

__global__ __launch_bounds__(1024) void finish_scan_kernel(const uint32_t *seg_count, unsigned long long *seg_off,
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

// pass B: survivors (bits of pass A) with rank in [win_lo, win_lo + win_n) are written, in
// reference order, as full records
__global__ __launch_bounds__(FN_NT) void finish_write_kernel(FinishArgs a)
{
    __shared__ uint32_t wtot[FN_NT / 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t seg = blockIdx.x;
    const uint64_t base = seg * FN_SEG + (uint64_t)wave * (FN_PER * 64);
    const unsigned long long *mk = a.masks + (seg * (FN_NT / 64) + wave) * FN_PER;
    uint32_t total = 0;
    for (int it = 0; it < FN_PER; it++) total += (uint32_t)__popcll(mk[it]);
    if (lane == 0) wtot[wave] = total;
    __syncthreads();
    uint64_t pos = a.seg_off[seg];
    for (uint32_t w = 0; w < wave; w++) pos += wtot[w];
    if (pos >= a.win_lo + a.win_n || pos + total <= a.win_lo) return;      // whole wave outside the window
    for (int it = 0; it < FN_PER; it++) {
        const unsigned long long mm = mk[it];
        if ((mm >> lane) & 1) {
            const uint64_t at = pos + __popcll(mm & ((1ull << lane) - 1));
            if (at >= a.win_lo && at - a.win_lo < a.win_n) {
                const uint64_t idx = base + (uint64_t)it * 64 + lane;
                const uint2 c = a.counts[idx];
                uint64_t row, col;
                pair_rc(a, idx, row, col);
                FinishEdge e;
                e.row = (uint32_t)row;
                e.col = (uint32_t)col;
                e.numer = c.x;
                e.denom = c.y;
                e.distance = lut_distance(a, c.x, c.y);
                e.p_value = p_value(c.x, a.len_row[row], a.len_col[col], a.kmer_space, c.y);
                a.edges[at - a.win_lo] = e;
            }
        }
        pos += __popcll(mm);
    }
}

// denominators present in a block of counts (for the distance table of finish_pairs_kernel)
__global__ __launch_bounds__(256) void denom_flags_kernel(const uint2 *counts, uint64_t pairs, uint32_t s, uint32_t *seen)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < pairs; idx += stride) {
        const uint32_t d = counts[idx].y;
        if (d != s && d <= s) seen[d] = 1u;                                // (row s is always tabulated)
    }
}

uint64_t finish_segments(uint64_t pairs) { return (pairs + FN_SEG - 1) / FN_SEG; }
uint64_t finish_mask_words(uint64_t pairs) { return finish_segments(pairs) * (FN_NT / 64) * FN_PER; }

hipError_t launch_finish_pairs(const FinishArgs &a, hipStream_t stream)
{
    if (a.pairs == 0) return hipSuccess;
    uint64_t blocks = (a.pairs + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(finish_pairs_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_finish_mark(const FinishArgs &a, unsigned long long *total, hipStream_t stream)
{
    const uint64_t nseg = finish_segments(a.pairs);
    if (nseg == 0 || nseg > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(finish_mark_kernel, dim3((uint32_t)nseg), dim3(FN_NT), 0, stream, a);
    hipLaunchKernelGGL(finish_scan_kernel, dim3(1), dim3(1024), 0, stream, a.seg_count, a.seg_off, nseg, total);
    return hipGetLastError();
}

hipError_t launch_finish_write(const FinishArgs &a, hipStream_t stream)
{
    const uint64_t nseg = finish_segments(a.pairs);
    if (nseg == 0 || nseg > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(finish_write_kernel, dim3((uint32_t)nseg), dim3(FN_NT), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_denom_flags(const uint2 *counts, uint64_t pairs, uint32_t s, uint32_t *seen, hipStream_t stream)
{
    if (pairs == 0) return hipSuccess;
    uint64_t blocks = (pairs + 1023) / 1024;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(denom_flags_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, counts, pairs, s, seen);
    return hipGetLastError();
}

}  // namespace mg
