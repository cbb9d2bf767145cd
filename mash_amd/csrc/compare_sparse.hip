// compare_sparse.hip — gfx950 pairwise comparison through an INVERTED INDEX of the sketch table.
//
// Same contract as the tile kernels (the merge loop of compareSketches, CommandDistance.cpp:347-385:
// {common, denom} per pair, reference output order), different cost: the tile engine
// (compare_merged.hip) pays for every pair, although two sketches that share no hash always give
// {0, min(s, |A| + |B|)} -- and in a collection of n sketches almost every pair is such a pair.
// Here the work is
//
//   fill      every output slot gets {0, min(s, |A| + |B|)}: one streaming write of 8 B per pair,
//             the compulsory HBM traffic of the whole job;
//   discover  the pairs that share at least one hash, from an index built once per table: all
//             (value, row) entries sorted by value (rows ascending inside a value), so the rows
//             sharing a value are one contiguous run; a workgroup per row ORs the runs of the
//             row's values into a bitmap of columns in LDS and appends the set bits to a
//             candidate list;
//   merge     one lane per candidate pair walks the reference's merge loop literally -- on
//             32-bit RANKS instead of 64-bit values: every entry of the table is replaced by
//             the dense rank of its value among the table's distinct values (a by-product of the
//             sort), which preserves order and equality inside the table, so the loop takes the
//             same branches and the counts are the reference's.
//
// Cost: 8 B written per pair + O(shared hashes) instead of O(n^2 * s / rows per tile).
// Rect (mash dist): the queries are located in the reference table's index by binary search; a
// query value between two table values gets the odd code between their even codes (2 * rank),
// which again preserves every comparison the merge makes.
//
// The sort of the index is rocPRIM's stable radix sort (value keys, entry ids as payload); every
// other step is a kernel below.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "compare_internal.h"

namespace mg {

// ------------------------------------------------------------------------------------------------
// index build
//
// Layout (round 4).  The code of an entry is 2 x the sorted position at which the GROUP of its value starts (the
// start position orders and identifies the values exactly as their dense rank does, and it needs no second scan,
// no gather).  Per table: keys_sorted[E], sorted_rows[E], gend[E] (defined at group starts: one past the group's
// last position), the code image (n x rs, padded with 0xFFFFFFFF) and the position image (n x rs: the entry's OWN
// sorted position, so [code / 2, own position) is the run of rows below it holding the same value).
// The sort's payload is the entry's index in the images (row * rs + position in the row): the pass that writes
// the images back needs neither a search for the row nor the compact entry ids.

// keys[e] = value, idx[e] = image index; row r owns the compact range [off[r], off[r + 1]) -- its first min(nhash, s) hashes
__global__ __launch_bounds__(256) void sp_fill_entries_kernel(const uint64_t *hashes, uint64_t stride, const uint32_t *off,
                                                              uint32_t rs, uint64_t *keys, uint32_t *idx)
{
    const uint32_t row = blockIdx.x;
    const uint32_t b = off[row], cnt = off[row + 1] - b;
    const uint64_t *src = hashes + (uint64_t)row * stride;
    const uint32_t ib = row * rs;
    for (uint32_t p = threadIdx.x; p < cnt; p += 256) {
        keys[b + p] = src[p];
        idx[b + p] = ib + p;
    }
}

// head[pos] = pos where a new value starts, else 0 (an inclusive max-scan then gives every position the start of
// its group); inside a value the image indices must ascend (the sort is stable and they ascend with the row),
// and the values themselves must ascend, else *bad is set and the index is not used
__global__ __launch_bounds__(256) void sp_heads_kernel(const uint64_t *keys, const uint32_t *idx, uint32_t E, uint32_t *head,
                                                       uint32_t *bad)
{
    const uint32_t pos = blockIdx.x * 256u + threadIdx.x;
    if (pos >= E) return;
    uint32_t h = pos;
    if (pos > 0) {
        const uint64_t k = keys[pos], kp = keys[pos - 1];
        if (k == kp) {
            h = 0;
            if (idx[pos] <= idx[pos - 1]) *bad = 1;
        } else if (k < kp) {
            *bad = 1;                                        // (not in order: a sort on leading bits whose ties were not all repaired)
        }
    }
    head[pos] = h;
}

// per sorted position: the row, and the entry's code and own position back into the images; the last position of a
// group leaves the group's end at the group's start.  Code = 2 x group start + (1 if another row holds the value too):
// the low bit travels with the value (every holder of a shared value carries it), so codes still compare as values do.
// Statistics go to one of SP_STAT_SLOTS slots per workgroup -- 390 000 workgroups adding to ONE address made this
// kernel 13.6 ms on C3 (5 ms on a table without repeated values, whose workgroups skipped two of the three atomics).
constexpr uint32_t SP_STAT_SLOTS = 1024;
struct SpStatSlot { unsigned long long inc; uint32_t max_group, groups; };

__global__ __launch_bounds__(256) void sp_index_scatter_kernel(const uint32_t *idx, const uint32_t *gs_of, uint32_t E,
                                                               uint32_t rs, uint32_t *sorted_rows, uint32_t *gend, uint32_t *code_img,
                                                               uint32_t *pos_img, SpStatSlot *stat)
{
    const uint32_t pos = blockIdx.x * 256u + threadIdx.x;
    unsigned long long inc = 0;
    uint32_t glen = 0, heads = 0;
    if (pos < E) {
        const uint32_t i = idx[pos];
        const uint32_t gs = gs_of[pos];
        const bool last = pos + 1u == E || gs_of[pos + 1u] != gs;
        const bool shared = !(last && gs == pos);
        sorted_rows[pos] = i / rs;
        code_img[i] = (gs << 1) | (shared ? 1u : 0u);
        pos_img[i] = pos;
        inc = pos - gs;
        heads = gs == pos ? 1u : 0u;
        if (last) {
            gend[gs] = pos + 1u;
            glen = pos + 1u - gs;
        }
    }
    // block sums (one atomic per workgroup and statistic, spread over the slots)
    for (int d = 32; d > 0; d >>= 1) {
        inc += __shfl_xor(inc, d);
        heads += __shfl_xor(heads, d);
        const uint32_t o = __shfl_xor(glen, d);
        glen = o > glen ? o : glen;
    }
    __shared__ unsigned long long s_inc[4];
    __shared__ uint32_t s_len[4], s_heads[4];
    if ((threadIdx.x & 63) == 0) { s_inc[threadIdx.x >> 6] = inc; s_len[threadIdx.x >> 6] = glen; s_heads[threadIdx.x >> 6] = heads; }
    __syncthreads();
    if (threadIdx.x == 0) {
        SpStatSlot *sl = stat + (blockIdx.x & (SP_STAT_SLOTS - 1u));
        const unsigned long long t = s_inc[0] + s_inc[1] + s_inc[2] + s_inc[3];
        uint32_t m = s_len[0];
        for (int w = 1; w < 4; w++) m = s_len[w] > m ? s_len[w] : m;
        if (t) atomicAdd(&sl->inc, t);
        if (m > 1) atomicMax(&sl->max_group, m);
        atomicAdd(&sl->groups, s_heads[0] + s_heads[1] + s_heads[2] + s_heads[3]);
    }
}

__global__ __launch_bounds__(256) void sp_stat_reduce_kernel(const SpStatSlot *stat, unsigned long long *incidences, uint32_t *max_group,
                                                             uint32_t *groups)
{
    unsigned long long inc = 0;
    uint32_t m = 0, g = 0;
    for (uint32_t i = threadIdx.x; i < SP_STAT_SLOTS; i += 256u) {
        inc += stat[i].inc;
        m = stat[i].max_group > m ? stat[i].max_group : m;
        g += stat[i].groups;
    }
    if (inc) atomicAdd(incidences, inc);
    if (m) atomicMax(max_group, m);
    if (g) atomicAdd(groups, g);
}

size_t sparse_stat_scratch_bytes() { return sizeof(SpStatSlot) * SP_STAT_SLOTS; }

// the entries of every row that enter the index: its first min(nhash, cap) hashes
__global__ __launch_bounds__(256) void sp_entry_counts_kernel(const uint32_t *nhash, uint32_t n, uint32_t cap, uint32_t *cnt)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) cnt[i] = nhash[i] < cap ? nhash[i] : cap;
}

hipError_t launch_sparse_row_counts(const uint32_t *nhash, uint32_t n, uint32_t cap, uint32_t *cnt, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sp_entry_counts_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, nhash, n, cap, cnt);
    return hipGetLastError();
}

// off[a] = entries of the rows in front of row a of the INDEX's order (row a there = table row inv[a]; inv == nullptr: the
// table's own order), off[n] = all of them.  The host knows these numbers, but an upload from pageable memory holds it
// until the stream has caught up -- with the offsets made here the build's kernels are queued behind the clustered copy
// while that is still running.  Two kernels: sums of blocks of 1024 rows; every block then adds up the sums in front of it
// (n / 1024 numbers) and scans its own rows.
__global__ __launch_bounds__(1024) void sp_offsets_sums_kernel(const uint32_t *cnt, const uint32_t *inv, uint32_t n, uint32_t *part)
{
    __shared__ uint32_t s_part[16];
    const uint32_t tid = threadIdx.x, a = blockIdx.x * 1024u + tid;
    uint32_t c = a < n ? cnt[inv ? inv[a] : a] : 0u;
#pragma unroll
    for (uint32_t d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
    if ((tid & 63u) == 0) s_part[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0;
        for (uint32_t k = 0; k < 16u; k++) t += s_part[k];
        part[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void sp_offsets_scan_kernel(const uint32_t *cnt, const uint32_t *inv, uint32_t n, const uint32_t *part, uint32_t *off)
{
    __shared__ uint32_t s_part[16], s_base[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6, a = blockIdx.x * 1024u + tid;
    uint32_t before = 0;
    for (uint32_t k = tid; k < blockIdx.x; k += 1024u) before += part[k];
#pragma unroll
    for (uint32_t d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d);
    if (lane == 0) s_base[wid] = before;
    const uint32_t c = a < n ? cnt[inv ? inv[a] : a] : 0u;
    uint32_t incl = c;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
    if (lane == 63u) s_part[wid] = incl;
    __syncthreads();
    uint32_t run = incl - c;
    for (uint32_t k = 0; k < 16u; k++) run += s_base[k] + (k < wid ? s_part[k] : 0u);
    if (a < n) off[a] = run;
    if (a + 1u == n) off[n] = run + c;
}

size_t sparse_offsets_temp_bytes(uint32_t n) { return ((size_t)n / 1024u + 1u) * 4u; }

hipError_t launch_sparse_offsets(const uint32_t *cnt, const uint32_t *inv, uint32_t n, void *temp, uint32_t *off, hipStream_t stream)
{
    if (n == 0) return hipMemsetAsync(off, 0, 4, stream);
    const uint32_t nb = (n + 1023u) / 1024u;
    hipLaunchKernelGGL(sp_offsets_sums_kernel, dim3(nb), dim3(1024), 0, stream, cnt, inv, n, static_cast<uint32_t *>(temp));
    hipLaunchKernelGGL(sp_offsets_scan_kernel, dim3(nb), dim3(1024), 0, stream, cnt, inv, n, static_cast<const uint32_t *>(temp), off);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void sp_fill_u32_kernel(uint32_t *p, uint64_t count, uint32_t v)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < count; i += stride) p[i] = v;
}

// ---- identical rows: collections hold many copies of one sketch (isolates of an outbreak, re-submitted
// genomes).  Copies form a CLASS represented by its first row: only representatives enter the index,
// discovery marks classes and expands them to rows when it lists candidates, and a pair inside a
// class is {n, n} without a merge.

// order-independent 64-bit digest of a row's first cnt values (sum of mixed (value, position) words)
__global__ __launch_bounds__(256) void sp_row_digest_kernel(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt,
                                                            unsigned long long *digest)
{
    const uint32_t row = blockIdx.x;
    const uint32_t n = cnt[row];
    const uint64_t *src = hashes + (uint64_t)row * stride;
    unsigned long long h = 0;
    for (uint32_t p = threadIdx.x; p < n; p += 256) {
        unsigned long long z = src[p] + 0x9E3779B97F4A7C15ull * (unsigned long long)(p + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        h += z ^ (z >> 31);
    }
    for (int d = 32; d > 0; d >>= 1) h += __shfl_xor(h, d);
    __shared__ unsigned long long s_h[4];
    if ((threadIdx.x & 63) == 0) s_h[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) digest[row] = s_h[0] + s_h[1] + s_h[2] + s_h[3];
}

// pairs[k] = {row, candidate representative}: equal[k] = 1 iff the two rows hold the same cnt values
__global__ __launch_bounds__(256) void sp_row_equal_kernel(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt,
                                                           const uint2 *pairs, uint32_t npairs, uint32_t *equal)
{
    const uint32_t k = blockIdx.x;
    if (k >= npairs) return;
    const uint2 pr = pairs[k];
    const uint32_t n = cnt[pr.x];
    int diff = cnt[pr.y] != n;
    const uint64_t *x = hashes + (uint64_t)pr.x * stride, *y = hashes + (uint64_t)pr.y * stride;
    for (uint32_t p = threadIdx.x; p < n && !diff; p += 256) diff |= x[p] != y[p];
    diff = __syncthreads_or(diff);
    if (threadIdx.x == 0) equal[k] = diff ? 0u : 1u;
}

hipError_t launch_sparse_row_digest(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt, uint32_t n,
                                    unsigned long long *digest, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sp_row_digest_kernel, dim3(n), dim3(256), 0, stream, hashes, stride, cnt, digest);
    return hipGetLastError();
}

hipError_t launch_sparse_row_equal(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt, const uint2 *pairs, uint32_t npairs,
                                   uint32_t *equal, hipStream_t stream)
{
    if (npairs == 0) return hipSuccess;
    hipLaunchKernelGGL(sp_row_equal_kernel, dim3(npairs), dim3(256), 0, stream, hashes, stride, cnt, pairs, npairs, equal);
    return hipGetLastError();
}

// ---- copies, found on the device: rows sorted by digest (stable, so equal digests keep ascending rows); a row
// whose digest and length equal its predecessor's is a suspect -- the host only looks at the sorted lists when
// there are suspects at all
__global__ __launch_bounds__(256) void sp_dup_flags_kernel(const unsigned long long *dig_sorted, const uint32_t *rows_sorted,
                                                           const uint32_t *cnt, uint32_t n, uint32_t *flags, uint32_t *nflag)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    uint32_t f = 0;
    if (k > 0 && dig_sorted[k] == dig_sorted[k - 1]) {
        const uint32_t c = cnt[rows_sorted[k]];
        f = (c != 0 && c == cnt[rows_sorted[k - 1]]) ? 1u : 0u;
    }
    flags[k] = f;
    if (f) atomicAdd(nflag, 1u);
}

size_t sparse_dup_temp_bytes(uint32_t n)
{
    size_t b = 0;
    rocprim::radix_sort_pairs(nullptr, b, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                              rocprim::counting_iterator<uint32_t>(0u), (uint32_t *)nullptr, (size_t)n, 0u, 64u, (hipStream_t) nullptr);
    return b;
}

// dig[n] -> dig_sorted[n], rows_sorted[n], flags[n], *nflag (zeroed here)
hipError_t launch_sparse_dup_suspects(const unsigned long long *dig, const uint32_t *cnt, uint32_t n, void *temp, size_t temp_bytes,
                                      unsigned long long *dig_sorted, uint32_t *rows_sorted, uint32_t *flags, uint32_t *nflag,
                                      hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(nflag, 0, 4, stream);
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(temp, temp_bytes, dig, dig_sorted, rocprim::counting_iterator<uint32_t>(0u), rows_sorted, (size_t)n, 0u, 64u,
                                  stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sp_dup_flags_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, dig_sorted, rows_sorted, cnt, n, flags, nflag);
    return hipGetLastError();
}

// ---- visiting order of the rows.  Key of a row: the start of the run of its first value that another row holds too
// (rows of one clade then sit next to each other and read the same runs), 0xFFFFFFFF for a row that shares nothing;
// a copy reads what its representative reads; inside a key larger rows first.  The 64-bit sort key is
// {key, ~row}; sorted ascending, the low words give the order.
__global__ __launch_bounds__(256) void sp_row_key_kernel(const uint32_t *off, const uint32_t *code_img, const uint32_t *gend,
                                                         const uint32_t *rep, uint32_t n, uint32_t rs, unsigned long long *key64)
{
    // one wave per row: 64 entries at a time, the first lane whose value another row holds too decides
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (row >= n) return;
    const uint32_t er = rep ? rep[row] : row;
    const uint32_t cnt = off[er + 1] - off[er];
    uint32_t k = 0xFFFFFFFFu;
    for (uint32_t base = 0; base < cnt; base += 64u) {
        const uint32_t p = base + lane;
        const uint32_t code = p < cnt ? code_img[(uint64_t)er * rs + p] : 0u;     // (low bit: another row holds the value too)
        const uint64_t m = __ballot((code & 1u) != 0);
        if (m != 0) {
            k = (uint32_t)__builtin_amdgcn_readlane((int)code, __builtin_ctzll(m)) >> 1;
            break;
        }
    }
    if (lane == 0) key64[row] = ((unsigned long long)k << 32) | (unsigned long long)(0xFFFFFFFFu - row);
}

__global__ __launch_bounds__(256) void sp_order_from_keys_kernel(const unsigned long long *key64_sorted, uint32_t n, uint32_t *order)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) order[i] = 0xFFFFFFFFu - (uint32_t)(key64_sorted[i] & 0xFFFFFFFFull);
}

size_t sparse_order_temp_bytes(uint32_t n)
{
    size_t b = 0;
    rocprim::radix_sort_keys(nullptr, b, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (size_t)n, 0u, 64u,
                             (hipStream_t) nullptr);
    return b;
}

// key_a / key_b: scratch of n u64 each; order[n] out
hipError_t launch_sparse_row_order(const uint32_t *off, const uint32_t *code_img, const uint32_t *gend, const uint32_t *rep, uint32_t n,
                                   uint32_t rs, void *temp, size_t temp_bytes, unsigned long long *key_a, unsigned long long *key_b,
                                   uint32_t *order, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sp_row_key_kernel, dim3((n + 3u) / 4u), dim3(256), 0, stream, off, code_img, gend, rep, n, rs, key_a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_keys(temp, temp_bytes, (const unsigned long long *)key_a, key_b, (size_t)n, 0u, 64u, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sp_order_from_keys_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, (const unsigned long long *)key_b, n, order);
    return hipGetLastError();
}

// the rows of [rb, re) in the table's visiting order (order-preserving selection)
struct sp_in_range {
    uint32_t rb, re;
    __device__ bool operator()(const uint32_t &r) const { return r >= rb && r < re; }
};

size_t sparse_order_slice_temp_bytes(uint32_t n)
{
    size_t b = 0;
    rocprim::select(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, sp_in_range{0u, 0u},
                    (hipStream_t) nullptr);
    return b;
}

hipError_t launch_sparse_order_slice(const uint32_t *order, uint32_t n, uint32_t rb, uint32_t re, void *temp, size_t temp_bytes,
                                     uint32_t *out, uint32_t *count_out, hipStream_t stream)
{
    return rocprim::select(temp, temp_bytes, order, out, count_out, (size_t)n, sp_in_range{rb, re}, stream);
}

// row stride of a code image: s rounded up to a chunk of four, plus one chunk the loop may load behind the row
uint32_t sparse_img_stride(uint32_t s) { return ((s + 3u) & ~3u) + 4u; }

struct sp_max_u32 {
    __device__ uint32_t operator()(const uint32_t &a, const uint32_t &b) const { return a > b ? a : b; }
};

size_t sparse_sort_temp_bytes(uint32_t E, uint32_t end_bit, uint32_t begin_bit)
{
    size_t bytes = 0;
    rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr,
                              (uint32_t *)nullptr, (size_t)E, 0u, end_bit, (hipStream_t) nullptr);
    size_t b1 = 0;                                           // (the sort on the leading bits only)
    rocprim::radix_sort_pairs(nullptr, b1, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const uint32_t *)nullptr,
                              (uint32_t *)nullptr, (size_t)E, begin_bit, end_bit, (hipStream_t) nullptr);
    bytes = bytes > b1 ? bytes : b1;
    size_t b2 = 0;
    rocprim::inclusive_scan(nullptr, b2, (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)E, sp_max_u32(), (hipStream_t) nullptr);
    return bytes > b2 ? bytes : b2;
}

// ---- fewer sort passes.  The values are hashes: 10^8 of them in a range of 2^54 need their top ~40 bits to be told apart,
// so the radix sort runs over the bits [begin_bit, end_bit) only (5 passes of 8 bits instead of 7) and the few segments in
// which two DIFFERENT values share all of those bits (expected: E^2 / 2^(bits + 1), a few thousand) are put in order
// afterwards: sp_tie_find_kernel lists the breaks between two such values, sp_tie_segments_kernel (a wave per break; the
// one at a segment's first break does the work) finds the segments' ends, sp_tie_repair_kernel (a wave per segment) rewrites
// each by its full values, smallest value first, entries of one value in the order the sort left them (stable: rows ascend).  A table that defeats the assumption
// (too many breaks, a segment too long to walk, more than 64 values in one) raises a flag and the caller builds the index
// again with every bit sorted.
constexpr uint32_t SP_TIE_WALK = 1024, SP_TIE_CAP = 1u << 16, SP_TIE_VALUES = 64;      // (walk: chunks of 64 entries)

__global__ __launch_bounds__(256) void sp_tie_find_kernel(const uint64_t *keys, uint32_t E, uint32_t bb, uint32_t *breaks, uint32_t *nbreak,
                                                          uint32_t *overflow)
{
    const uint32_t pos = blockIdx.x * 256u + threadIdx.x;
    if (pos == 0 || pos >= E) return;
    const uint64_t k = keys[pos], kp = keys[pos - 1];
    if ((kp >> bb) != (k >> bb) || k == kp) return;        // not a break between two values of one prefix
    const uint32_t slot = atomicAdd(nbreak, 1u);
    if (slot >= SP_TIE_CAP) { *overflow = 1; return; }
    breaks[slot] = pos;
}

// one wave per break: the wave at a segment's FIRST break finds the segment's ends and lists it (nothing is rewritten
// while other waves still walk)
__global__ __launch_bounds__(64) void sp_tie_segments_kernel(const uint64_t *keys, uint32_t E, uint32_t bb, const uint32_t *breaks,
                                                             const uint32_t *nbreak, uint2 *segs, uint32_t *nseg, uint32_t *overflow)
{
    const uint32_t m = *nbreak < SP_TIE_CAP ? *nbreak : SP_TIE_CAP;
    if (blockIdx.x >= m) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t pos = breaks[blockIdx.x];
    const uint64_t kp = keys[pos - 1];
    const uint64_t prefix = kp >> bb;
    // left: over the entries that hold kp, to the segment's start; another value on the way = an earlier break, whose wave does the work
    uint32_t s0 = 0;
    {
        uint32_t hi = pos - 1;                              // entries [.., hi) are still to look at
        uint32_t chunks = 0;
        for (;;) {
            if (hi == 0) { s0 = 0; break; }
            const bool in = lane < hi;
            const uint64_t k = in ? keys[hi - 1u - lane] : 0;
            const bool other_prefix = in && (k >> bb) != prefix;
            const bool other_value = in && !other_prefix && k != kp;
            const uint64_t stop = __ballot(other_prefix || other_value);
            if (stop) {
                const uint32_t first = (uint32_t)__builtin_ctzll(stop);
                if ((__ballot(other_value) >> first) & 1ull) return;
                s0 = hi - first;
                break;
            }
            if (hi <= 64u) { s0 = 0; break; }
            hi -= 64u;
            if (++chunks > SP_TIE_WALK) { if (lane == 0) *overflow = 1; return; }
        }
    }
    uint32_t s1 = E;
    {
        uint32_t lo = pos;
        uint32_t chunks = 0;
        for (;;) {
            if (lo >= E) { s1 = E; break; }
            const uint32_t q = lo + lane;
            const bool out = q < E && (keys[q] >> bb) != prefix;
            const uint64_t stop = __ballot(out);
            if (stop) { s1 = lo + (uint32_t)__builtin_ctzll(stop); break; }
            lo += 64u;
            if (++chunks > SP_TIE_WALK) { if (lane == 0) *overflow = 1; return; }
        }
    }
    if (lane == 0) segs[atomicAdd(nseg, 1u)] = make_uint2(s0, s1);      // (at most one per break)
}

// one wave per segment: rewritten by its full values
__global__ __launch_bounds__(64) void sp_tie_repair_kernel(uint64_t *keys, uint32_t *idx, const uint2 *segs, const uint32_t *nseg, uint64_t *tk,
                                                           uint32_t *ti, uint32_t *overflow)
{
    if (blockIdx.x >= *nseg) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t s0 = segs[blockIdx.x].x, s1 = segs[blockIdx.x].y;
    // selection by value: smallest first, the entries of one value in the order the sort left them (stable)
    uint32_t out = s0, rounds = 0;
    bool have = false;
    uint64_t last = 0;
    while (out < s1) {
        uint64_t best = ~0ull;
        for (uint32_t q = s0 + lane; q < s1; q += 64u) {
            const uint64_t k = keys[q];
            if ((!have || k > last) && k < best) best = k;
        }
        for (uint32_t d = 32; d; d >>= 1) {
            const uint64_t o = __shfl_xor(best, d, 64);
            best = o < best ? o : best;
        }
        if (++rounds > SP_TIE_VALUES) { if (lane == 0) *overflow = 1; return; }      // (nothing written back: the caller sorts every bit)
        for (uint32_t base = s0; base < s1; base += 64u) {
            const uint32_t q = base + lane;
            const bool hit = q < s1 && keys[q] == best;
            const uint64_t bal = __ballot(hit);
            if (hit) {
                const uint32_t w = out + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                tk[w] = best;
                ti[w] = idx[q];
            }
            out += (uint32_t)__popcll(bal);
        }
        last = best;
        have = true;
    }
    __threadfence();
    for (uint32_t q = s0 + lane; q < s1; q += 64u) {
        keys[q] = __hip_atomic_load(&tk[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        idx[q] = __hip_atomic_load(&ti[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// bits the sort looks at: enough that two different values rarely agree in all of them.  forced_bits (nullable, a test
// knob): that many bits whatever the table's size -- few bits, many ties; all_bits: every bit
uint32_t sparse_sort_begin_bit(uint32_t E, uint32_t end_bit, const char *forced_bits, bool all_bits)
{
    // (end_bit 64: on 1155 entries and bits [48, 64) rocprim's path for small inputs returned a sequence that was not in
    // order -- seen, not pursued; values that reach the top bit are rare and the gain is one pass, so every bit is sorted)
    if (all_bits || end_bit >= 64u) return 0;
    const char *forced = forced_bits;
    if (!forced && E < (1u << 22)) return 0;                         // (small tables: nothing to gain)
    uint32_t lg = 0;
    while ((1ull << lg) < (uint64_t)E) lg++;
    const uint32_t want = 2u * lg > 29u ? 2u * lg - 13u : 16u;      // expected ties E^2 / 2^(bits + 1) = 2^12
    uint32_t passes = (want + 7u) / 8u;                              // whole passes of 8 bits; one less if that costs two bits at most
    if (passes > 1u && want - (passes - 1u) * 8u <= 2u) passes--;
    uint32_t bits = passes * 8u;
    if (forced) bits = (uint32_t)atoi(forced);
    if (bits < 8u) bits = 8u;
    if (bits + 8u > end_bit) return 0;                               // (less than a pass to gain)
    return end_bit - bits;
}

size_t sparse_tie_scratch_bytes() { return (size_t)SP_TIE_CAP * (sizeof(uint32_t) + sizeof(uint2)) + 16; }

// All device buffers are the caller's.  keys_a / idx_a: scratch of E entries each (input of the sort),
// keys_sorted / idx_sorted: its output; head: scratch of E u32 (may be idx_a), gs_of: scratch of E u32.  On return
// (stream order) the index arrays are complete; *bad != 0 means the order inside a value was not by row (never
// seen: the sort is stable) and the index must not be used.  *incidences, *max_group, *groups, *bad: zeroed by the caller.
hipError_t sparse_build_index(const uint64_t *hashes, uint64_t stride, const uint32_t *off, uint32_t n, uint32_t E,
                              uint32_t rs, uint32_t end_bit, void *temp, size_t temp_bytes, uint64_t *keys_a,
                              uint32_t *idx_a, uint64_t *keys_sorted, uint32_t *idx_sorted, uint32_t *head, uint32_t *gs_of,
                              uint32_t *sorted_rows, uint32_t *gend, uint32_t *code_img, uint32_t *pos_img, void *stat_scratch,
                              uint32_t begin_bit, void *tie_scratch, unsigned long long *incidences, uint32_t *max_group, uint32_t *groups,
                              uint32_t *bad, uint32_t *tie_overflow, hipStream_t stream)
{
    if (n == 0 || E == 0) return hipSuccess;
    hipLaunchKernelGGL(sp_fill_entries_kernel, dim3(n), dim3(256), 0, stream, hashes, stride, off, rs, keys_a, idx_a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint64_t *)keys_a, keys_sorted, (const uint32_t *)idx_a, idx_sorted,
                                  (size_t)E, begin_bit, end_bit, stream);
    if (e != hipSuccess) return e;
    const uint32_t blocks = (E + 255u) / 256u;
    if (begin_bit > 0) {
        // values that agree in every sorted bit: found, and put in order by their full values (keys_a / idx_a, the sort's input, are scratch now)
        uint2 *segs = static_cast<uint2 *>(tie_scratch);
        uint32_t *breaks = reinterpret_cast<uint32_t *>(segs + SP_TIE_CAP);
        uint32_t *nbreak = breaks + SP_TIE_CAP, *nseg = nbreak + 1;
        e = hipMemsetAsync(nbreak, 0, 8, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(sp_tie_find_kernel, dim3(blocks), dim3(256), 0, stream, (const uint64_t *)keys_sorted, E, begin_bit, breaks, nbreak, tie_overflow);
        hipLaunchKernelGGL(sp_tie_segments_kernel, dim3(SP_TIE_CAP), dim3(64), 0, stream, (const uint64_t *)keys_sorted, E, begin_bit,
                           (const uint32_t *)breaks, (const uint32_t *)nbreak, segs, nseg, tie_overflow);
        hipLaunchKernelGGL(sp_tie_repair_kernel, dim3(SP_TIE_CAP), dim3(64), 0, stream, keys_sorted, idx_sorted, (const uint2 *)segs,
                           (const uint32_t *)nseg, keys_a, idx_a, tie_overflow);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(sp_heads_kernel, dim3(blocks), dim3(256), 0, stream, keys_sorted, idx_sorted, E, head, bad);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::inclusive_scan(temp, temp_bytes, (const uint32_t *)head, gs_of, (size_t)E, sp_max_u32(), stream);
    if (e != hipSuccess) return e;
    // padding of the code image: larger than every code, so chunked loads past a row's end are harmless
    {
        const uint64_t total = (uint64_t)n * rs;
        uint64_t fb = (total + 1023) / 1024;
        if (fb > 8192) fb = 8192;
        hipLaunchKernelGGL(sp_fill_u32_kernel, dim3((uint32_t)fb), dim3(256), 0, stream, code_img, total, 0xFFFFFFFFu);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    e = hipMemsetAsync(stat_scratch, 0, sparse_stat_scratch_bytes(), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sp_index_scatter_kernel, dim3(blocks), dim3(256), 0, stream, (const uint32_t *)idx_sorted, (const uint32_t *)gs_of, E, rs,
                       sorted_rows, gend, code_img, pos_img, static_cast<SpStatSlot *>(stat_scratch));
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sp_stat_reduce_kernel, dim3(1), dim3(256), 0, stream, static_cast<const SpStatSlot *>(stat_scratch), incidences, max_group,
                       groups);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// rect: locate the queries' values in the reference table's index

// One thread per query entry: code in the reference table's code space and the run of rows holding the value.
__global__ __launch_bounds__(256) void sp_locate_kernel(const uint64_t *qhashes, uint64_t qstride, const uint32_t *qoff,
                                                        uint32_t q_begin, uint32_t nq, const uint64_t *keys_sorted,
                                                        const uint32_t *gend, uint32_t E, uint32_t rs, uint32_t *qlo_img,
                                                        uint32_t *qhi_img, uint32_t *qcode_img)
{
    const uint32_t q = blockIdx.x;                       // query index relative to q_begin
    if (q >= nq) return;
    const uint32_t cnt = qoff[q + 1] - qoff[q];
    const uint64_t *src = qhashes + (uint64_t)(q_begin + q) * qstride;
    for (uint32_t p = threadIdx.x; p < cnt; p += 256) {
        const uint64_t v = src[p];
        uint32_t lo = 0, hi = E;                          // lower bound: first position with key >= v
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (keys_sorted[mid] < v) lo = mid + 1; else hi = mid;
        }
        // Codes of the query image: table codes are 2 * (start position of the value's group) (+ 1 for values several
        // rows hold), and the merge sets the low bit of every table code it loads in rect mode (an unsigned code
        // cannot sit below the first group otherwise): a value found -- its lower bound IS its group's start -- gets
        // 2 lo + 1, equal to the table's; a value between two groups gets 2 lo (lo = E: above every key), which is
        // above the previous group's 2 gs + 1 (gs < lo) and below the next group's 2 lo + 1.
        const bool found = lo < E && keys_sorted[lo] == v;
        const uint64_t at = (uint64_t)q * rs + p;
        qcode_img[at] = found ? (lo << 1) + 1u : (lo << 1);
        qlo_img[at] = lo;
        qhi_img[at] = found ? gend[lo] : lo;
    }
}

// ------------------------------------------------------------------------------------------------
// discovery

// rows of the class of representative c that count as partners of `row`: all of them (rect), those
// below `row` (triangle); a representative without copies is its own single row
__device__ __forceinline__ uint32_t sp_class_span(const SparseArgs &a, uint32_t c, uint32_t row, uint32_t &first)
{
    const uint32_t k = a.cls_of[c];
    if (k == 0xFFFFFFFFu) { first = 0xFFFFFFFFu; return 1u; }
    const uint32_t lo = a.cls_off[k], hi = a.cls_off[k + 1];
    first = lo;
    if (!a.triangle) return hi - lo;
    uint32_t l = lo, h = hi;                               // class rows ascend: how many lie below `row`
    while (l < h) {
        const uint32_t mid = (l + h) >> 1;
        if (a.cls_rows[mid] < row) l = mid + 1; else h = mid;
    }
    return l - lo;
}

// DEDUP: the column table holds copies (a.rep / a.cls_* are set).  Bits of the bitmap then stand for
// CLASSES (bit = the representative's row); a copy takes its representative's index entries -- and,
// being a later row, every class of their values whose representative lies below it -- and classes
// are expanded to rows when the candidates are listed.  Pairs INSIDE a class never become
// candidates: they are {n, n}, written by sp_class_pairs_kernel.
template <bool COUNT_ONLY, bool DEDUP>
__global__ __launch_bounds__(256) void sp_discover_kernel(SparseArgs a)
{
    extern __shared__ uint32_t bm[];
    __shared__ uint32_t s_w[4];
    __shared__ unsigned long long s_base;
    __shared__ unsigned long long s_inc[4];
    // rows in the order of a.order (rows that share runs of the index next to each other: the runs then stay
    // in L2), else largest rows first
    const uint32_t row = a.order ? a.order[blockIdx.x] : a.row_end - 1u - blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t ncols = a.triangle ? row : a.ncols;
    const uint32_t W = (ncols + 31u) >> 5;
    const uint32_t er = (DEDUP && a.triangle) ? a.rep[row] : row;      // the row whose index entries speak for this one
    const bool copy = DEDUP && a.triangle && er != row;
    const uint32_t cnt = a.off[er + 1] - a.off[er];
    if (cnt == 0 || ncols == 0) return;                  // uniform
    if (!copy) {
        // a row none of whose values is held by a row below it (most rows of a collection of unrelated genomes) has
        // nothing to mark: neither the bitmap (n / 8 bytes of LDS to clear and to scan) nor the rest is needed
        int any = 0;
        for (uint32_t p = tid; p < cnt; p += 256u) {
            const uint64_t at = (uint64_t)er * a.rs_row + p;
            any |= (a.lo_img[at] >> a.lo_shift) != a.hi_img[at];
        }
        if (__syncthreads_or(any) == 0) return;          // uniform
    }
    for (uint32_t w = tid; w < W; w += 256) bm[w] = 0;
    __syncthreads();
    unsigned long long inc = 0;
    constexpr uint32_t SHORT = 6;
    for (uint32_t base = wid * 64u; base < cnt; base += 256u) {
        const uint32_t p = base + lane;
        uint2 lh = make_uint2(0u, 0u);
        if (p < cnt) {
            const uint64_t at = (uint64_t)er * a.rs_row + p;
            lh = make_uint2(a.lo_img[at] >> a.lo_shift, a.hi_img[at]);
            // a copy: the whole run of the value (its classes ascend; those at or above `row` are skipped below)
            if (copy) lh.y = a.gend[lh.x];
        }
        const uint32_t len = lh.y - lh.x;
        inc += len;
        if (len != 0 && len <= SHORT) {
            for (uint32_t t = 0; t < len; t++) {
                const uint32_t r = a.sorted_rows[lh.x + t];
                if (copy && r >= row) break;
                if (copy && r == er) continue;           // its own class: sp_class_pairs_kernel
                if (!((bm[r >> 5] >> (r & 31u)) & 1u)) atomicOr(&bm[r >> 5], 1u << (r & 31u));
            }
        }
        // Long runs, four at a time: the first 64 rows of each are requested before any is consumed (a run is
        // a dependent global load away; one run at a time left the wave waiting on each), the rest of a run
        // longer than 64 follows in strides.
        // (test before set: the rows of a clade name each other in every one of their runs, and consecutive
        //  rows share a word -- reads of one word broadcast, atomics on it queue up)
        auto mark = [&](uint32_t r) {
            if (copy && (r >= row || r == er)) return;               // (its own class: sp_class_pairs_kernel)
            if (!((bm[r >> 5] >> (r & 31u)) & 1u)) atomicOr(&bm[r >> 5], 1u << (r & 31u));
        };
        uint64_t longm = __ballot(len > SHORT);
        while (longm != 0) {
            uint32_t lo4[4], hi4[4], r4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                lo4[k] = hi4[k] = 0;
                if (longm != 0) {                                    // uniform
                    const int l = __builtin_ctzll(longm);
                    longm &= longm - 1;
                    lo4[k] = (uint32_t)__builtin_amdgcn_readlane((int)lh.x, l);
                    hi4[k] = (uint32_t)__builtin_amdgcn_readlane((int)lh.y, l);
                }
                r4[k] = lo4[k] + lane < hi4[k] ? a.sorted_rows[lo4[k] + lane] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (r4[k] != 0xFFFFFFFFu) mark(r4[k]);
                for (uint32_t q = lo4[k] + 64u + lane; q < hi4[k]; q += 64u) mark(a.sorted_rows[q]);
            }
        }
    }
    __syncthreads();
    // set bits of the bitmap -> candidates, in column order; thread t owns a contiguous run of words
    const uint32_t per = (W + 255u) / 256u;
    const uint32_t w0 = tid * per, w1 = w0 + per < W ? w0 + per : W;
    uint32_t mine = 0;
    for (uint32_t w = w0; w < w1; w++) {
        if (!DEDUP) {
            mine += (uint32_t)__popc(bm[w]);
        } else {
            uint32_t bits = bm[w];
            while (bits != 0) {
                const uint32_t c = (w << 5) + (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1;
                uint32_t first;
                mine += sp_class_span(a, c, row, first);
            }
        }
    }
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += t;
    }
    for (int d = 32; d > 0; d >>= 1) inc += __shfl_xor(inc, d);
    if (lane == 63) s_w[wid] = incl;
    if (lane == 0) s_inc[wid] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (uint32_t w = 0; w < wid; w++) woff += s_w[w];
    const uint32_t total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    if (tid == 0) {
        s_base = atomicAdd(&a.counters[0], (unsigned long long)total);
        atomicAdd(&a.counters[1], s_inc[0] + s_inc[1] + s_inc[2] + s_inc[3]);
    }
    if (COUNT_ONLY) return;
    __syncthreads();
    const unsigned long long gbase = s_base;
    if (gbase + total > a.cand_cap) {                    // uniform; the host grows the list and repeats
        if (tid == 0) a.counters[2] = 1;
        return;
    }
    if (tid == 0) {                                      // the row's segment of the list (slot = blockIdx.x)
        a.seg_base[blockIdx.x] = gbase;
        a.seg_cnt[blockIdx.x] = total;
    }
    unsigned long long o = gbase + woff + incl - mine;
    for (uint32_t w = w0; w < w1; w++) {
        uint32_t bits = bm[w];
        while (bits != 0) {
            const uint32_t c = (w << 5) + (uint32_t)__builtin_ctz(bits);
            bits &= bits - 1;
            if (!DEDUP) {
                a.cand[o++] = make_uint2(row, c);
            } else {
                uint32_t first;
                const uint32_t k = sp_class_span(a, c, row, first);
                if (first == 0xFFFFFFFFu) a.cand[o++] = make_uint2(row, c);
                else for (uint32_t t = 0; t < k; t++) a.cand[o++] = make_uint2(row, a.cls_rows[first + t]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// merge: one lane per candidate pair, the reference's loop on 32-bit codes

__device__ __forceinline__ uint32_t sp_pick(const uint4 &c, uint32_t k)
{
    const uint32_t lo = (k & 1u) ? c.y : c.x;
    const uint32_t hi = (k & 1u) ? c.w : c.z;
    return (k & 2u) ? hi : lo;
}

// RECT: column codes are the table's 2 * rank, the query image is in the shifted space (see
// sp_locate_kernel), so the column side adds one to whatever it loads (the padding 0xFFFFFFFF is
// never compared: the loop ends at the row's length).
template <bool RECT>
__global__ __launch_bounds__(256) void sp_merge_kernel(SparseArgs a)
{
    unsigned long long K = a.counters[0];
    if (K > a.cand_cap) K = a.cand_cap;
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    const uint32_t s = a.s;
    for (uint64_t c = (uint64_t)blockIdx.x * 256u + threadIdx.x; c < K; c += stride) {
        const uint2 pr = a.cand[c];
        // copies are compared through their representatives; two rows of one class are {n, n}
        const uint32_t i = (a.rep && !RECT) ? a.rep[pr.x] : pr.x, j = a.rep ? a.rep[pr.y] : pr.y;
        const uint32_t nA = a.off[i + 1] - a.off[i];
        if (!RECT && i == j) { a.res[c] = make_uint2(nA, nA); continue; }
        const uint32_t nB = a.col_cnt_off[j + 1] - a.col_cnt_off[j];
        const uint4 *A4 = reinterpret_cast<const uint4 *>(a.row_img + (uint64_t)i * a.rs_row);
        const uint4 *B4 = reinterpret_cast<const uint4 *>(a.col_img + (uint64_t)j * a.rs_col);
        uint4 ca = A4[0], cb = B4[0];
        uint32_t av = ca.x, bv = RECT ? (cb.x | 1u) : cb.x;
        uint32_t ia = 0, ib = 0, common = 0, denom = 0;
        while (denom < s && ia < nA && ib < nB) {          // CommandDistance.cpp:347-365
            const bool adva = av <= bv, advb = bv <= av;
            common += (adva && advb) ? 1u : 0u;
            denom++;
            if (adva) {
                ia++;
                if ((ia & 3u) == 0) ca = A4[ia >> 2];
                av = sp_pick(ca, ia & 3u);
            }
            if (advb) {
                ib++;
                if ((ib & 3u) == 0) cb = B4[ib >> 2];
                bv = sp_pick(cb, ib & 3u);
                if (RECT) bv |= 1u;
            }
        }
        if (denom < s) {                                   // :367-385
            denom += (nA - ia) + (nB - ib);
            if (denom > s) denom = s;
        }
        a.res[c] = make_uint2(common, denom);
    }
}

// Work items of the row-blocked merge: SPM_NT candidates of one row.  chunks[slot] = items of the
// row in discover slot `slot`; an inclusive scan turns it into the item -> row map.
constexpr uint32_t SPM_NT = 128;
constexpr uint32_t SPM_RING = 32;                         // column codes a lane keeps in LDS
constexpr uint32_t SPM_AWIN = 2048;                       // codes of the row staged in LDS at a time

__global__ __launch_bounds__(256) void sp_chunks_kernel(const uint32_t *seg_cnt, uint32_t nrows, uint32_t *chunks)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r < nrows) chunks[r] = (seg_cnt[r] + SPM_NT - 1u) / SPM_NT;
}

// Row-blocked merge: a workgroup = one row and up to SPM_NT of its candidates, one lane per
// candidate.  The row's codes sit in LDS for everyone; every lane streams its column's codes through
// a private ring of SPM_RING codes in LDS (code e of lane l at word (e mod 32) * 64 + l: reads are
// conflict free), refilled eight codes at a
// time on a fixed cadence -- every 8 merge steps a lane lands the chunk it requested a round ago
// and requests the next if its ring has room -- so the loop body has no divergent loads and a
// request has a whole round to arrive.  A lane advances at most 8 codes per round and holds at
// least 16 after landing, so it never runs dry; a chunk is only requested when the 8 codes it
// overwrites are consumed.  The loop itself is the reference's (CommandDistance.cpp:347-385).
// (two instantiations of the same loop: this one keeps the WHOLE row in LDS -- rows of up to SPM_AWIN codes, every
//  sketch size in common use -- and has no window logic in its rounds; sp_merge_rows_win_kernel below stages the
//  row a window at a time.  Measured at s = 1000: 4.3 ms against 6.9 ms for the windowed form of the same pass.)
template <bool RECT>
__global__ __launch_bounds__(SPM_NT) void sp_merge_rows_kernel(SparseArgs a)
{
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t nrows = a.row_end - a.row_begin;
    const uint32_t item = blockIdx.x;
    if (item >= a.chunk_inc[nrows - 1]) return;
    uint32_t lo = 0, hi = nrows - 1;                     // first slot whose inclusive count exceeds item
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.chunk_inc[mid] > item) hi = mid; else lo = mid + 1;
    }
    const uint32_t slot = lo;
    const uint32_t chunk = item - (slot ? a.chunk_inc[slot - 1] : 0u);
    const uint32_t row = a.order ? a.order[slot] : a.row_end - 1u - slot;      // as sp_discover_kernel maps its workgroups
    const uint32_t cnt = a.seg_cnt[slot];
    const uint64_t base = a.seg_base[slot] + (uint64_t)chunk * SPM_NT;
    const uint32_t left = cnt - chunk * SPM_NT;
    const bool have = tid < left;
    const uint32_t s = a.s;
    // copies are compared through their representatives; two rows of one class are {n, n}
    const uint32_t arow = (a.rep && !RECT) ? a.rep[row] : row;
    const uint32_t nA = a.off[arow + 1] - a.off[arow];
    // the row's codes (and one chunk of its padding: A[nA] is read by a lane that has just finished)
    uint32_t *A = lds;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.row_img + (uint64_t)arow * a.rs_row);
        const uint32_t nvec = (nA >> 2) + 1u;
        for (uint32_t v = tid; v < nvec; v += SPM_NT) reinterpret_cast<uint4 *>(A)[v] = src[v];
    }
    uint32_t *myring = lds + a.rs_row + (tid >> 6) * (SPM_RING * 64u) + lane;   // code e of this lane: myring[(e & 31) * 64]
    uint32_t j = have ? a.cand[base + tid].y : 0u;
    if (a.rep) j = a.rep[j];
    const bool same = !RECT && a.rep != nullptr && j == arow;
    const uint32_t nB = have ? a.col_cnt_off[j + 1] - a.col_cnt_off[j] : 0u;
    const uint4 *B4 = reinterpret_cast<const uint4 *>(a.col_img + (uint64_t)j * a.rs_col);
    auto land = [&](uint32_t e, const uint4 &v) {         // codes e .. e + 3 (e a multiple of 4)
        uint32_t *q = myring + (e & (SPM_RING - 1u)) * 64u;
        q[0] = v.x; q[64] = v.y; q[128] = v.z; q[192] = v.w;
    };
    // codes [0, 16) land now, [16, 24) are the first pending chunk
    {
        const uint4 x0 = B4[0], x1 = B4[1], x2 = B4[2], x3 = B4[3];
        land(0, x0); land(4, x1); land(8, x2); land(12, x3);
    }
    uint4 p0 = B4[4], p1 = B4[5];
    uint32_t loaded = 16;
    bool pend = true;
    __syncthreads();                                     // A staged
    uint32_t ia = 0, ib = 0, denom = 0;                  // (common = ia + ib - denom: a match advances both sides for one union element)
    bool active = have && !same && s > 0 && nA > 0 && nB > 0;
    while (__ballot(active) != 0) {
        // A lane with at least 8 codes left on both sides and 8 union elements to go cannot reach any of
        // the loop's three bounds within 8 steps: when that holds for every lane still merging, the round
        // runs without the per-step tests (two compares, two advances, two LDS reads per step).
        uint32_t room = 0;
        if (active) {
            const uint32_t ra = nA - ia, rb = nB - ib, rd = s - denom;
            room = ra < rb ? ra : rb;
            room = room < rd ? room : rd;
        }
        if (__ballot(active && room < 8u) == 0) {
            if (active) {
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const uint32_t av = A[ia];
                    uint32_t bv = myring[(ib & (SPM_RING - 1u)) * 64u];
                    if (RECT) bv |= 1u;
                    ia += av <= bv ? 1u : 0u;
                    ib += bv <= av ? 1u : 0u;
                }
                denom += 8;
                active = denom < s && ia < nA && ib < nB;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const uint32_t av = A[ia];
                uint32_t bv = myring[(ib & (SPM_RING - 1u)) * 64u];
                if (RECT) bv |= 1u;
                const bool adva = active && av <= bv, advb = active && bv <= av;
                denom += active ? 1u : 0u;
                ia += adva ? 1u : 0u;
                ib += advb ? 1u : 0u;
                active = active && denom < s && ia < nA && ib < nB;
            }
        }
        if (pend) {
            land(loaded, p0);
            land(loaded + 4u, p1);
            loaded += 8;
        }
        pend = active && loaded + 8u - ib <= SPM_RING;
        if (pend) {
            p0 = B4[loaded >> 2];
            p1 = B4[(loaded >> 2) + 1u];
        }
    }
    if (have) {
        uint32_t common = ia + ib - denom;
        if (same) {
            common = denom = nA;
        } else if (denom < s) {                            // :367-385
            denom += (nA - ia) + (nB - ib);
            if (denom > s) denom = s;
        }
        a.res[base + tid] = make_uint2(common, denom);
    }
}

template <bool RECT>
__global__ __launch_bounds__(SPM_NT) void sp_merge_rows_win_kernel(SparseArgs a)
{
    extern __shared__ __align__(16) uint32_t lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t nrows = a.row_end - a.row_begin;
    const uint32_t item = blockIdx.x;
    if (item >= a.chunk_inc[nrows - 1]) return;
    uint32_t lo = 0, hi = nrows - 1;                     // first slot whose inclusive count exceeds item
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.chunk_inc[mid] > item) hi = mid; else lo = mid + 1;
    }
    const uint32_t slot = lo;
    const uint32_t chunk = item - (slot ? a.chunk_inc[slot - 1] : 0u);
    const uint32_t row = a.order ? a.order[slot] : a.row_end - 1u - slot;      // as sp_discover_kernel maps its workgroups
    const uint32_t cnt = a.seg_cnt[slot];
    const uint64_t base = a.seg_base[slot] + (uint64_t)chunk * SPM_NT;
    const uint32_t left = cnt - chunk * SPM_NT;
    const bool have = tid < left;
    const uint32_t s = a.s;
    // copies are compared through their representatives; two rows of one class are {n, n}
    const uint32_t arow = (a.rep && !RECT) ? a.rep[row] : row;
    const uint32_t nA = a.off[arow + 1] - a.off[arow];
    // The row's codes sit in LDS a WINDOW of SPM_AWIN at a time (a whole row of s <= 2048; a row of
    // s = 10 000 in five windows: 40 KB of LDS per workgroup would leave four waves per CU).  A lane that
    // reaches the window's end waits there; the next window is staged when none is left running.
    uint32_t *A = lds;
    const uint32_t awords = a.rs_row < SPM_AWIN + 8u ? a.rs_row : SPM_AWIN + 8u;
    uint32_t *myring = lds + awords + (tid >> 6) * (SPM_RING * 64u) + lane;      // code e of this lane: myring[(e & 31) * 64]
    uint32_t j = have ? a.cand[base + tid].y : 0u;
    if (a.rep) j = a.rep[j];
    const bool same = !RECT && a.rep != nullptr && j == arow;
    const uint32_t nB = have ? a.col_cnt_off[j + 1] - a.col_cnt_off[j] : 0u;
    const uint4 *B4 = reinterpret_cast<const uint4 *>(a.col_img + (uint64_t)j * a.rs_col);
    auto land = [&](uint32_t e, const uint4 &v) {         // codes e .. e + 3 (e a multiple of 4)
        uint32_t *q = myring + (e & (SPM_RING - 1u)) * 64u;
        q[0] = v.x; q[64] = v.y; q[128] = v.z; q[192] = v.w;
    };
    // codes [0, 16) land now, [16, 24) are the first pending chunk
    {
        const uint4 x0 = B4[0], x1 = B4[1], x2 = B4[2], x3 = B4[3];
        land(0, x0); land(4, x1); land(8, x2); land(12, x3);
    }
    uint4 p0 = B4[4], p1 = B4[5];
    uint32_t loaded = 16;
    bool pend = true;
    uint32_t ia = 0, ib = 0, denom = 0;                  // (common = ia + ib - denom: a match advances both sides for one union element)
    bool active = have && !same && s > 0 && nA > 0 && nB > 0;
    const uint4 *Asrc = reinterpret_cast<const uint4 *>(a.row_img + (uint64_t)arow * a.rs_row);
    for (uint32_t a0 = 0;; a0 += SPM_AWIN) {
        const uint32_t aend = a0 + SPM_AWIN < nA ? a0 + SPM_AWIN : nA;       // this window: codes [a0, aend) (+ one chunk: A[aend] is read, not used)
        {
            const uint32_t nvec = ((aend - a0) >> 2) + 1u;
            for (uint32_t v = tid; v < nvec; v += SPM_NT) reinterpret_cast<uint4 *>(A)[v] = Asrc[(a0 >> 2) + v];
        }
        __syncthreads();
        bool running = active && ia < aend;
        while (__ballot(running) != 0) {
            // A lane with at least 8 codes left on both sides (of the window, of its column) and 8 union
            // elements to go cannot reach any bound within 8 steps: when that holds for every lane still
            // running, the round runs without the per-step tests (two compares, two advances, two LDS reads).
            uint32_t room = 0;
            if (running) {
                const uint32_t ra = aend - ia, rb = nB - ib, rd = s - denom;
                room = ra < rb ? ra : rb;
                room = room < rd ? room : rd;
            }
            if (__ballot(running && room < 8u) == 0) {
                if (running) {
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        const uint32_t av = A[ia - a0];
                        uint32_t bv = myring[(ib & (SPM_RING - 1u)) * 64u];
                        if (RECT) bv |= 1u;
                        ia += av <= bv ? 1u : 0u;
                        ib += bv <= av ? 1u : 0u;
                    }
                    denom += 8;
                }
            } else {
                bool go = running;
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const uint32_t av = A[go ? ia - a0 : 0u];
                    uint32_t bv = myring[(ib & (SPM_RING - 1u)) * 64u];
                    if (RECT) bv |= 1u;
                    const bool adva = go && av <= bv, advb = go && bv <= av;
                    denom += go ? 1u : 0u;
                    ia += adva ? 1u : 0u;
                    ib += advb ? 1u : 0u;
                    go = go && denom < s && ia < aend && ib < nB;
                }
            }
            if (running) active = denom < s && ia < nA && ib < nB;
            running = active && ia < aend;
            if (pend) {
                land(loaded, p0);
                land(loaded + 4u, p1);
                loaded += 8;
            }
            pend = active && loaded + 8u - ib <= SPM_RING;
            if (pend) {
                p0 = B4[loaded >> 2];
                p1 = B4[(loaded >> 2) + 1u];
            }
        }
        if (aend >= nA) break;                             // uniform: the row is through
        if (__syncthreads_or(active ? 1 : 0) == 0) break;  // nobody waits for the next window (also: this window's reads are done)
    }
    if (have) {
        uint32_t common = ia + ib - denom;
        if (same) {
            common = denom = nA;
        } else if (denom < s) {                            // :367-385
            denom += (nA - ia) + (nB - ib);
            if (denom > s) denom = s;
        }
        a.res[base + tid] = make_uint2(common, denom);
    }
}

// The same merge with SEVERAL rows per work item.  A collection's rows have tens of candidates each (C3: 0 ... 99,
// 50 on average), so an item of one row fills half its lanes.  Here the candidates of all rows form one line of
// UNITS in visiting order -- a row with candidates takes max(candidates, SPM_PACK_MIN) units, so that at most
// SPM_PACK_ROWS rows meet in the 128 units of an item -- and item t takes units [128 t, 128 t + 128): every lane
// finds its row among the item's (at most one row boundary lies between two units 32 apart, so the rows of the
// units 0, 32, 64, 96 and 127 are all there are), all of them are staged in LDS, and the loop is the one above
// with a per-lane row base.  Lanes on a row's padding units idle (C3: 9 % against 43 %).
__global__ __launch_bounds__(256) void sp_pack_costs_kernel(const uint32_t *seg_cnt, uint32_t nrows, uint32_t pack_min, uint32_t *chunks)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r < nrows) {
        const uint32_t c = seg_cnt[r];
        chunks[r] = c ? (c < pack_min ? pack_min : c) : 0u;
    }
}

// SPM_PACK_MIN: units a row with candidates takes at least (32: up to 5 rows per item, 43: 4, 64: 3 -- fewer rows staged
// = more workgroups per CU, more padding units = more idle lanes)
template <bool RECT, uint32_t SPM_PACK_MIN>
__global__ __launch_bounds__(SPM_NT) void sp_merge_pack_kernel(SparseArgs a)
{
    constexpr uint32_t SPM_PACK_ROWS = (SPM_NT + SPM_PACK_MIN - 1u) / SPM_PACK_MIN + 1u;
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ uint32_t pslot[SPM_PACK_ROWS];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t nrows = a.row_end - a.row_begin;
    const uint32_t total = a.chunk_inc[nrows - 1];         // units in all (inclusive scan of the rows' costs)
    const uint32_t u0 = blockIdx.x * SPM_NT;
    if (u0 >= total) return;
    if (tid < SPM_PACK_ROWS) {                             // the rows of units 0, 32, 64, 96, 127 of this item
        uint32_t u = u0 + (tid == SPM_PACK_ROWS - 1u ? SPM_NT - 1u : tid * SPM_PACK_MIN);
        if (u >= total) u = total - 1u;
        uint32_t lo = 0, hi = nrows - 1;                   // first slot whose inclusive cost exceeds u
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.chunk_inc[mid] > u) hi = mid; else lo = mid + 1;
        }
        pslot[tid] = lo;
    }
    __syncthreads();
    const uint32_t s = a.s;
    // stage the item's rows (copies are compared through their representatives)
    {
        uint32_t nd = 0;
        for (uint32_t j = 0; j < SPM_PACK_ROWS; j++) {
            if (j && pslot[j] == pslot[j - 1]) continue;
            const uint32_t sl = pslot[j];
            const uint32_t row = a.order ? a.order[sl] : a.row_end - 1u - sl;
            const uint32_t ar = (a.rep && !RECT) ? a.rep[row] : row;
            const uint32_t n = a.off[ar + 1] - a.off[ar];
            const uint4 *src = reinterpret_cast<const uint4 *>(a.row_img + (uint64_t)ar * a.rs_row);
            uint4 *dst = reinterpret_cast<uint4 *>(lds + nd * a.rs_row);
            const uint32_t nvec = (n >> 2) + 1u;           // (and one chunk of the padding: A[nA] is read by a lane that has just finished)
            for (uint32_t v = tid; v < nvec; v += SPM_NT) dst[v] = src[v];
            nd++;
        }
    }
    // this lane's unit -> row, candidate
    const uint32_t u = u0 + tid;
    const uint32_t sA = pslot[tid / SPM_PACK_MIN], sB = pslot[tid / SPM_PACK_MIN + 1u];
    uint32_t slot = sA;
    if (sB != sA && u >= a.chunk_inc[sB - 1u]) slot = sB;  // (sB > sA: the units before sB's first end with the slot before it)
    const uint32_t q = u - (slot ? a.chunk_inc[slot - 1u] : 0u);
    const uint32_t cnt = a.seg_cnt[slot];
    const bool have = u < total && q < cnt;
    uint32_t ridx = 0;
#pragma unroll
    for (uint32_t j = 1; j < SPM_PACK_ROWS; j++) ridx += (pslot[j] != pslot[j - 1] && pslot[j] <= slot) ? 1u : 0u;
    const uint32_t row = a.order ? a.order[slot] : a.row_end - 1u - slot;
    const uint32_t arow = (a.rep && !RECT) ? a.rep[row] : row;
    const uint32_t nA = a.off[arow + 1] - a.off[arow];
    const uint32_t *A = lds + ridx * a.rs_row;
    const uint64_t at = a.seg_base[slot] + q;              // this lane's candidate (and result slot)
    uint32_t *myring = lds + SPM_PACK_ROWS * a.rs_row + (tid >> 6) * (SPM_RING * 64u) + lane;   // code e of this lane: myring[(e & 31) * 64]
    uint32_t j = have ? a.cand[at].y : 0u;
    if (a.rep) j = a.rep[j];
    const bool same = !RECT && a.rep != nullptr && j == arow;
    const uint32_t nB = have ? a.col_cnt_off[j + 1] - a.col_cnt_off[j] : 0u;
    const uint4 *B4 = reinterpret_cast<const uint4 *>(a.col_img + (uint64_t)j * a.rs_col);
    auto land = [&](uint32_t e, const uint4 &v) {         // codes e .. e + 3 (e a multiple of 4)
        uint32_t *p = myring + (e & (SPM_RING - 1u)) * 64u;
        p[0] = v.x; p[64] = v.y; p[128] = v.z; p[192] = v.w;
    };
    {
        const uint4 x0 = B4[0], x1 = B4[1], x2 = B4[2], x3 = B4[3];
        land(0, x0); land(4, x1); land(8, x2); land(12, x3);
    }
    uint4 p0 = B4[4], p1 = B4[5];
    uint32_t loaded = 16;
    bool pend = true;
    __syncthreads();                                     // rows staged
    uint32_t ia = 0, ib = 0, denom = 0;                  // (common = ia + ib - denom)
    bool active = have && !same && s > 0 && nA > 0 && nB > 0;
    while (__ballot(active) != 0) {
        uint32_t room = 0;
        if (active) {
            const uint32_t ra = nA - ia, rb = nB - ib, rd = s - denom;
            room = ra < rb ? ra : rb;
            room = room < rd ? room : rd;
        }
        if (__ballot(active && room < 8u) == 0) {
            if (active) {
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const uint32_t av = A[ia];
                    uint32_t bv = myring[(ib & (SPM_RING - 1u)) * 64u];
                    if (RECT) bv |= 1u;
                    ia += av <= bv ? 1u : 0u;
                    ib += bv <= av ? 1u : 0u;
                }
                denom += 8;
                active = denom < s && ia < nA && ib < nB;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const uint32_t av = A[ia];
                uint32_t bv = myring[(ib & (SPM_RING - 1u)) * 64u];
                if (RECT) bv |= 1u;
                const bool adva = active && av <= bv, advb = active && bv <= av;
                denom += active ? 1u : 0u;
                ia += adva ? 1u : 0u;
                ib += advb ? 1u : 0u;
                active = active && denom < s && ia < nA && ib < nB;
            }
        }
        if (pend) {
            land(loaded, p0);
            land(loaded + 4u, p1);
            loaded += 8;
        }
        pend = active && loaded + 8u - ib <= SPM_RING;
        if (pend) {
            p0 = B4[loaded >> 2];
            p1 = B4[(loaded >> 2) + 1u];
        }
    }
    if (have) {
        uint32_t common = ia + ib - denom;
        if (same) {
            common = denom = nA;
        } else if (denom < s) {                            // :367-385
            denom += (nA - ia) + (nB - ib);
            if (denom > s) denom = s;
        }
        a.res[at] = make_uint2(common, denom);
    }
}

// chunks / chunk_inc / temp as for launch_sparse_merge_rows; false in *used: the job is not one for this kernel (rows
// longer than the LDS window, too much LDS) and the caller takes launch_sparse_merge_rows
hipError_t launch_sparse_merge_pack(const SparseArgs &a, uint64_t expect, uint32_t *chunks, void *temp, size_t temp_bytes, bool *used,
                                    hipStream_t stream)
{
    *used = false;
    const uint32_t nrows = a.row_end - a.row_begin;
    if (expect == 0 || nrows == 0) return hipSuccess;
    uint32_t pack_min = 32;
    if (const char *ev = getenv("MASHGPU_SPARSE_PACK_MIN")) pack_min = atoi(ev) >= 64 ? 64u : atoi(ev) >= 43 ? 43u : 32u;
    const uint32_t pack_rows = (SPM_NT + pack_min - 1u) / pack_min + 1u;
    const size_t smem = ((size_t)pack_rows * a.rs_row + (SPM_NT / 64u) * SPM_RING * 64u) * 4;
    if (a.rs_row > SPM_AWIN + 8u || smem > 160 * 1024 - 256) return hipSuccess;
    const uint64_t items = (expect + (uint64_t)pack_min * nrows) / SPM_NT + 1;
    if (items >= (1ull << 31) || expect + (uint64_t)pack_min * nrows >= (1ull << 32)) return hipSuccess;
    hipLaunchKernelGGL(sp_pack_costs_kernel, dim3((nrows + 255u) / 256u), dim3(256), 0, stream, a.seg_cnt, nrows, pack_min, chunks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::inclusive_scan(temp, temp_bytes, (const uint32_t *)chunks, a.chunk_inc, (size_t)nrows, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e2 != hipSuccess) return e2;
        hipLaunchKernelGGL(kern, dim3((uint32_t)items), dim3(SPM_NT), smem, stream, a);
        return hipGetLastError();
    };
    *used = true;
    if (pack_min == 32) return a.triangle ? go(sp_merge_pack_kernel<false, 32>) : go(sp_merge_pack_kernel<true, 32>);
    if (pack_min == 43) return a.triangle ? go(sp_merge_pack_kernel<false, 43>) : go(sp_merge_pack_kernel<true, 43>);
    return a.triangle ? go(sp_merge_pack_kernel<false, 64>) : go(sp_merge_pack_kernel<true, 64>);
}


// candidates in REFERENCE order (rows ascending; a row's segment is already in column order): the rows'
// counts (by row), then, after an exclusive scan, every segment copied to its row's place
__global__ __launch_bounds__(256) void sp_row_counts_kernel(SparseArgs a, uint32_t *cnt_by_row)
{
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    const uint32_t nrows = a.row_end - a.row_begin;
    if (slot >= nrows) return;
    const uint32_t row = a.order ? a.order[slot] : a.row_end - 1u - slot;
    uint32_t c = a.seg_cnt[slot];
    if (a.dn_grp_of) {                                     // (triangle) a grouped row: its partners inside the group follow its candidates
        const uint32_t g = a.dn_grp_of[row];
        if (g != 0xFFFFFFFFu) c += row - a.dn_groups[g].g0;
    }
    cnt_by_row[row - a.row_begin] = c;
}

__global__ __launch_bounds__(256) void sp_gather_rows_kernel(SparseArgs a, const uint32_t *row_base, uint32_t row_add, uint2 *rc_out, uint2 *counts_out)
{
    const uint32_t slot = blockIdx.x;
    const uint32_t row = a.order ? a.order[slot] : a.row_end - 1u - slot;
    const uint32_t cnt = a.seg_cnt[slot];
    const unsigned long long src = a.seg_base[slot];
    const uint32_t dst = row_base[row - a.row_begin];
    for (uint32_t t = threadIdx.x; t < cnt; t += 256) {
        const uint2 pr = a.cand[src + t];
        rc_out[dst + t] = make_uint2(pr.x + row_add, pr.y);
        counts_out[dst + t] = a.res[src + t];
    }
}

hipError_t launch_sparse_gather_rows(const SparseArgs &a, uint32_t *cnt_by_row, uint32_t *row_base, void *temp, size_t temp_bytes, uint32_t row_add,
                                     uint2 *rc_out, uint2 *counts_out, hipStream_t stream)
{
    const uint32_t nrows = a.row_end - a.row_begin;
    if (nrows == 0) return hipSuccess;
    hipLaunchKernelGGL(sp_row_counts_kernel, dim3((nrows + 255u) / 256u), dim3(256), 0, stream, a, cnt_by_row);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::exclusive_scan(temp, temp_bytes, (const uint32_t *)cnt_by_row, row_base, 0u, (size_t)nrows, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sp_gather_rows_kernel, dim3(nrows), dim3(256), 0, stream, a, (const uint32_t *)row_base, row_add, rc_out, counts_out);
    return hipGetLastError();
}

// ---- a list of pairs -> the entries that are worth a record: {row, col, numer, denom} of every pair with numer >= 1, order kept.
// Blocks of 1024 entries: counted, the counts scanned (rocPRIM), written at their ranks.
constexpr uint32_t SP_EDGE_BLOCK = 1024;

__global__ __launch_bounds__(256) void sp_edges_count_kernel(const uint2 *counts, uint64_t K, uint32_t *blk_cnt)
{
    const uint64_t base = (uint64_t)blockIdx.x * SP_EDGE_BLOCK;
    uint32_t c = 0;
    for (uint32_t k = threadIdx.x; k < SP_EDGE_BLOCK; k += 256u)
        if (base + k < K && counts[base + k].x != 0) c++;
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
    __shared__ uint32_t s_c[4];
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

__global__ __launch_bounds__(256) void sp_edges_write_kernel(const uint2 *rc, const uint2 *counts, uint64_t K, const uint32_t *blk_off, uint32_t nblk,
                                                             uint4 *edges, unsigned long long *total)
{
    const uint64_t base = (uint64_t)blockIdx.x * SP_EDGE_BLOCK;
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    __shared__ uint32_t s_w[4];
    uint32_t run = blk_off[blockIdx.x];
    for (uint32_t k0 = 0; k0 < SP_EDGE_BLOCK; k0 += 256u) {               // (entry order = k0 + thread: kept)
        const uint64_t i = base + k0 + threadIdx.x;
        uint2 c = make_uint2(0u, 0u);
        if (i < K) c = counts[i];
        const bool keep = i < K && c.x != 0;
        const uint64_t bal = __ballot(keep);
        if (lane == 0) s_w[wid] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < wid; w++) before += s_w[w];
        const uint32_t all = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (keep) {
            const uint2 p = rc[i];
            edges[run + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = make_uint4(p.x, p.y, c.x, c.y);
        }
        run += all;
        __syncthreads();
    }
    if (blockIdx.x == nblk - 1u && threadIdx.x == 0) *total = run;
}

size_t sparse_edges_temp_bytes(uint64_t K)
{
    size_t b = 0;
    rocprim::exclusive_scan(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (size_t)((K + SP_EDGE_BLOCK - 1) / SP_EDGE_BLOCK), rocprim::plus<uint32_t>(),
                            (hipStream_t) nullptr);
    return b;
}

uint64_t sparse_edges_blocks(uint64_t K) { return (K + SP_EDGE_BLOCK - 1) / SP_EDGE_BLOCK; }

// rc / counts: K list entries; blk_cnt / blk_off: scratch of sparse_edges_blocks(K) u32; edges: room for K; *total (device): how many were written
hipError_t launch_sparse_list_edges(const uint2 *rc, const uint2 *counts, uint64_t K, uint32_t *blk_cnt, uint32_t *blk_off, void *temp, size_t temp_bytes,
                                    uint4 *edges, unsigned long long *total, hipStream_t stream)
{
    if (K == 0) return hipMemsetAsync(total, 0, 8, stream);
    const uint32_t nblk = (uint32_t)sparse_edges_blocks(K);
    hipLaunchKernelGGL(sp_edges_count_kernel, dim3(nblk), dim3(256), 0, stream, counts, K, blk_cnt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::exclusive_scan(temp, temp_bytes, (const uint32_t *)blk_cnt, blk_off, 0u, (size_t)nblk, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sp_edges_write_kernel, dim3(nblk), dim3(256), 0, stream, rc, counts, K, (const uint32_t *)blk_off, nblk, edges, total);
    return hipGetLastError();
}

size_t sparse_gather_temp_bytes(uint32_t nrows)
{
    size_t b = 0;
    rocprim::exclusive_scan(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (size_t)nrows, rocprim::plus<uint32_t>(), (hipStream_t) nullptr);
    return b;
}

// results of the candidates -> their output slots (after the fill)
__global__ __launch_bounds__(256) void sp_scatter_kernel(SparseArgs a)
{
    unsigned long long K = a.counters[0];
    if (K > a.cand_cap) K = a.cand_cap;
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t c = (uint64_t)blockIdx.x * 256u + threadIdx.x; c < K; c += stride) {
        const uint2 pr = a.cand[c];
        uint32_t i = pr.x, j = pr.y;
        if (a.inv) {                                       // (triangle on a permuted index: back to the table's rows)
            i = a.inv[pr.x];
            j = a.inv[pr.y];
            if (i < j) { const uint32_t t = i; i = j; j = t; }
        }
        uint64_t oidx;
        if (a.triangle) oidx = (uint64_t)i * (i - 1u) / 2u + j - a.out_base;
        else oidx = (uint64_t)(i - a.row_begin) * a.ncols + j;
        a.out[oidx] = a.res[c];
    }
}

// pairs of two copies of one sketch (triangle): {n, n}, n = the sketch's hash count.  One workgroup per
// member of a class of two rows and more: member i of cls_rows writes its pairs with the members of
// its class below it (first[i] = where its class starts in cls_rows).
__global__ __launch_bounds__(256) void sp_class_pairs_kernel(uint2 *out, const uint32_t *cls_rows, const uint32_t *cls_first,
                                                             const uint32_t *off, const uint32_t *rep, uint32_t members,
                                                             uint32_t row_begin, uint32_t row_end, uint64_t out_base, const uint32_t *inv)
{
    const uint32_t i = blockIdx.x;
    if (i >= members) return;
    const uint32_t row = cls_rows[i];
    if (row < row_begin || row >= row_end) return;       // uniform
    const uint32_t r = rep[row], n = off[r + 1] - off[r];
    const uint2 v = make_uint2(n, n);
    if (!inv) {
        const uint64_t base = (uint64_t)row * (row - 1u) / 2u - out_base;
        for (uint32_t u = cls_first[i] + threadIdx.x; u < i; u += 256) out[base + cls_rows[u]] = v;
    } else {                                               // (permuted index: back to the table's rows)
        const uint32_t oi = inv[row];
        for (uint32_t u = cls_first[i] + threadIdx.x; u < i; u += 256) {
            const uint32_t oj = inv[cls_rows[u]];
            const uint32_t hi = oi > oj ? oi : oj, lo = oi > oj ? oj : oi;
            out[(uint64_t)hi * (hi - 1u) / 2u - out_base + lo] = v;
        }
    }
}

hipError_t launch_sparse_class_pairs(uint2 *out, const uint32_t *cls_rows, const uint32_t *cls_first, const uint32_t *off,
                                     const uint32_t *rep, uint32_t members, uint32_t row_begin, uint32_t row_end, uint64_t out_base,
                                     const uint32_t *inv, hipStream_t stream)
{
    if (members == 0) return hipSuccess;
    hipLaunchKernelGGL(sp_class_pairs_kernel, dim3(members), dim3(256), 0, stream, out, cls_rows, cls_first, off, rep, members, row_begin,
                       row_end, out_base, inv);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// fill: {0, denom} for every pair

typedef uint32_t sp_u32x4 __attribute__((ext_vector_type(4)));

// every wave writes 4 KB of consecutive addresses per round (four 16-byte nontemporal stores 1 KB apart; 6.7 ms for the 40 GB of
// C3 against 7.1 ms with four streams a grid apart, profiles/r03_sparse_tuning.json); {0, s} for a collection, {c, c} for a
// table that is nothing but copies of one sketch
__global__ __launch_bounds__(256) void sp_fill_value_kernel(uint2 *out, uint64_t pairs, uint32_t numer, uint32_t denom)
{
    const uint64_t head = ((reinterpret_cast<uintptr_t>(out) & 8u) != 0 && pairs > 0) ? 1u : 0u;
    const uint64_t nvec = (pairs - head) >> 1;
    sp_u32x4 *body = reinterpret_cast<sp_u32x4 *>(out + head);
    const sp_u32x4 v = {numer, denom, numer, denom};
    const uint64_t lane = threadIdx.x & 63u;
    const uint64_t gw = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6), tw = (uint64_t)gridDim.x * 4u;
    for (uint64_t base = gw * 256u; base < nvec; base += tw * 256u) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t i = base + (uint64_t)u * 64u + lane;
            if (i < nvec) __builtin_nontemporal_store(v, body + i);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (head) out[0] = make_uint2(numer, denom);
        if (((pairs - head) & 1u) != 0) out[pairs - 1] = make_uint2(numer, denom);
    }
}

// The same fill in CHUNKS of 64 KB that a wave takes from a counter, with `naps` pauses of 64 cycles behind every 4 KB it
// writes: launched with few workgroups and some naps it runs beside kernels that live on round trips to memory (the index
// build: a fill at full speed keeps the memory's queues full of writes and the build's loads wait behind them), and a second
// launch -- whole device, no naps -- takes what is left from the same counter when the other work has ended.
constexpr uint32_t SP_FILL_CHUNK = 4096;                   // 16-byte vectors per chunk

__global__ __launch_bounds__(256) void sp_fill_chunks_kernel(uint2 *out, uint64_t pairs, uint32_t numer, uint32_t denom, uint32_t naps, uint32_t *ctr,
                                                             uint32_t nchunks)
{
    const uint64_t head = ((reinterpret_cast<uintptr_t>(out) & 8u) != 0 && pairs > 0) ? 1u : 0u;
    const uint64_t nvec = (pairs - head) >> 1;
    sp_u32x4 *body = reinterpret_cast<sp_u32x4 *>(out + head);
    const sp_u32x4 v = {numer, denom, numer, denom};
    const uint32_t lane = threadIdx.x & 63u;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (head) out[0] = make_uint2(numer, denom);
        if (((pairs - head) & 1u) != 0) out[pairs - 1] = make_uint2(numer, denom);
    }
    for (;;) {
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(ctr, 1u);
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        if (c >= nchunks) break;
        const uint64_t first = (uint64_t)c * SP_FILL_CHUNK;
        for (uint32_t r = 0; r < SP_FILL_CHUNK; r += 256u) {
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) {
                const uint64_t i = first + r + u * 64u + lane;
                if (i < nvec) __builtin_nontemporal_store(v, body + i);
            }
            for (uint32_t z = 0; z < naps; z++) __builtin_amdgcn_s_sleep(1);
        }
    }
}

uint64_t sparse_fill_chunks(uint64_t pairs) { return (pairs / 2 + SP_FILL_CHUNK - 1) / SP_FILL_CHUNK; }

hipError_t launch_sparse_fill_chunks(uint2 *out, uint64_t pairs, uint32_t numer, uint32_t denom, uint32_t blocks, uint32_t naps, uint32_t *ctr,
                                     hipStream_t stream, uint32_t threads)
{
    if (pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(sp_fill_chunks_kernel, dim3(blocks ? blocks : 256u), dim3(threads), 0, stream, out, pairs, numer, denom, naps, ctr,
                       (uint32_t)sparse_fill_chunks(pairs));
    return hipGetLastError();
}

hipError_t launch_sparse_fill_value(uint2 *out, uint64_t pairs, uint32_t numer, uint32_t denom, uint32_t blocks_per_cu, uint32_t cus,
                                    hipStream_t stream)
{
    if (pairs == 0) return hipSuccess;
    uint64_t blocks = (pairs / 2 + 1023) / 1024;
    const uint64_t most = (uint64_t)(cus ? cus : 256) * (blocks_per_cu ? blocks_per_cu : 16);
    if (blocks > most) blocks = most;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(sp_fill_value_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, out, pairs, numer, denom);
    return hipGetLastError();
}

// pairs of two SHORT sketches (|A| + |B| < s): denom = |A| + |B|.  One workgroup per short row of
// the row side, threads over the short rows of the column side (both lists ascending).
__global__ __launch_bounds__(256) void sp_fill_short_kernel(uint2 *out, const uint32_t *short_rows, const uint32_t *short_rcnt,
                                                            uint32_t nshort_rows, const uint32_t *short_cols,
                                                            const uint32_t *short_ccnt, uint32_t nshort_cols, uint32_t row_begin,
                                                            uint32_t ncols, uint32_t triangle, uint64_t out_base, uint32_t s,
                                                            const uint32_t *inv)
{
    const uint32_t k = blockIdx.x;
    if (k >= nshort_rows) return;
    const uint32_t i = short_rows[k], ni = short_rcnt[k];
    for (uint32_t t = threadIdx.x; t < nshort_cols; t += 256) {
        const uint32_t j = short_cols[t];
        if (triangle && j >= i) break;                      // ascending
        const uint32_t d = ni + short_ccnt[t];
        if (d < s) {
            uint32_t oi = i, oj = j;
            if (inv) {                                      // (triangle on a permuted index: back to the table's rows)
                oi = inv[i];
                oj = inv[j];
                if (oi < oj) { const uint32_t x = oi; oi = oj; oj = x; }
            }
            const uint64_t oidx = triangle ? (uint64_t)oi * (oi - 1u) / 2u + oj - out_base : (uint64_t)(oi - row_begin) * ncols + oj;
            out[oidx] = make_uint2(0u, d);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers

hipError_t launch_sparse_locate(const uint64_t *qhashes, uint64_t qstride, const uint32_t *qoff, uint32_t q_begin, uint32_t nq,
                                const uint64_t *keys_sorted, const uint32_t *gend, uint32_t E, uint32_t rs, uint32_t *qlo_img,
                                uint32_t *qhi_img, uint32_t *qcode_img, hipStream_t stream)
{
    if (nq == 0) return hipSuccess;
    {
        const uint64_t total = (uint64_t)nq * rs;
        uint64_t fb = (total + 1023) / 1024;
        if (fb > 8192) fb = 8192;
        hipLaunchKernelGGL(sp_fill_u32_kernel, dim3((uint32_t)fb), dim3(256), 0, stream, qcode_img, total, 0xFFFFFFFFu);
    }
    hipLaunchKernelGGL(sp_locate_kernel, dim3(nq), dim3(256), 0, stream, qhashes, qstride, qoff, q_begin, nq, keys_sorted, gend, E, rs,
                       qlo_img, qhi_img, qcode_img);
    return hipGetLastError();
}

size_t sparse_discover_lds(uint32_t ncols_max) { return (((size_t)ncols_max + 31) / 32) * 4 + 16; }
bool sparse_discover_supported(uint32_t ncols_max) { return sparse_discover_lds(ncols_max) <= 160 * 1024 - 256; }

hipError_t launch_sparse_discover(const SparseArgs &a, bool count_only, hipStream_t stream)
{
    const uint32_t nrows = a.row_end - a.row_begin;
    if (nrows == 0) return hipSuccess;
    const size_t smem = sparse_discover_lds(a.triangle ? a.row_end : a.ncols);
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(nrows), dim3(256), smem, stream, a);
        return hipGetLastError();
    };
    const bool dedup = a.rep != nullptr;
    if (count_only) return dedup ? go(sp_discover_kernel<true, true>) : go(sp_discover_kernel<true, false>);
    return dedup ? go(sp_discover_kernel<false, true>) : go(sp_discover_kernel<false, false>);
}

hipError_t launch_sparse_merge(const SparseArgs &a, uint64_t expect, uint32_t cus, hipStream_t stream)
{
    if (expect == 0) return hipSuccess;
    uint64_t blocks = (expect + 255) / 256;
    const uint64_t most = (uint64_t)(cus ? cus : 256) * 32;      // grid-stride beyond ~8 workgroups per CU x 4 rounds
    if (blocks > most) blocks = most;
    if (a.triangle) hipLaunchKernelGGL(sp_merge_kernel<false>, dim3((uint32_t)blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(sp_merge_kernel<true>, dim3((uint32_t)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

size_t sparse_merge_rows_lds(uint32_t rs_row)
{
    const size_t awords = rs_row < SPM_AWIN + 8u ? rs_row : SPM_AWIN + 8u;
    return (awords + (SPM_NT / 64u) * SPM_RING * 64u) * 4;
}
bool sparse_merge_rows_supported(uint32_t rs_row) { return sparse_merge_rows_lds(rs_row) <= 160 * 1024 - 256; }
size_t sparse_scan_temp_bytes(uint32_t nrows)
{
    size_t b = 0;
    rocprim::inclusive_scan(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)nrows, rocprim::plus<uint32_t>(),
                            (hipStream_t) nullptr);
    return b;
}

// Row-blocked merge of the candidates the discover launch listed: work items from the rows' segment
// sizes (scan), then one workgroup per item.  `expect`: the candidate count of the counting pass.
hipError_t launch_sparse_merge_rows(const SparseArgs &a, uint64_t expect, uint32_t *chunks, void *temp, size_t temp_bytes,
                                    hipStream_t stream)
{
    const uint32_t nrows = a.row_end - a.row_begin;
    if (expect == 0 || nrows == 0) return hipSuccess;
    hipLaunchKernelGGL(sp_chunks_kernel, dim3((nrows + 255u) / 256u), dim3(256), 0, stream, a.seg_cnt, nrows, chunks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::inclusive_scan(temp, temp_bytes, (const uint32_t *)chunks, a.chunk_inc, (size_t)nrows, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    const uint64_t items = expect / SPM_NT + nrows;              // upper bound: one partial item per row
    if (items >= (1ull << 31)) return hipErrorInvalidValue;
    const size_t smem = sparse_merge_rows_lds(a.rs_row);
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e2 != hipSuccess) return e2;
        hipLaunchKernelGGL(kern, dim3((uint32_t)items), dim3(SPM_NT), smem, stream, a);
        return hipGetLastError();
    };
    // rows that fit the LDS window whole take the kernel without window logic
    const bool whole = a.rs_row <= SPM_AWIN + 8u && !getenv("MASHGPU_SPARSE_MERGE_WINDOWS");
    if (a.triangle) return whole ? go(sp_merge_rows_kernel<false>) : go(sp_merge_rows_win_kernel<false>);
    return whole ? go(sp_merge_rows_kernel<true>) : go(sp_merge_rows_win_kernel<true>);
}

hipError_t launch_sparse_scatter(const SparseArgs &a, uint64_t expect, uint32_t cus, hipStream_t stream)
{
    if (expect == 0) return hipSuccess;
    uint64_t blocks = (expect + 255) / 256;
    const uint64_t most = (uint64_t)(cus ? cus : 256) * 16;
    if (blocks > most) blocks = most;
    hipLaunchKernelGGL(sp_scatter_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_sparse_fill_short(uint2 *out, const uint32_t *short_rows, const uint32_t *short_rcnt, uint32_t nshort_rows,
                                    const uint32_t *short_cols, const uint32_t *short_ccnt, uint32_t nshort_cols,
                                    uint32_t row_begin, uint32_t ncols, uint32_t triangle, uint64_t out_base, uint32_t s,
                                    const uint32_t *inv, hipStream_t stream)
{
    if (nshort_rows == 0 || nshort_cols == 0) return hipSuccess;
    hipLaunchKernelGGL(sp_fill_short_kernel, dim3(nshort_rows), dim3(256), 0, stream, out, short_rows, short_rcnt, nshort_rows,
                       short_cols, short_ccnt, nshort_cols, row_begin, ncols, triangle, out_base, s, inv);
    return hipGetLastError();
}

}  // namespace mg
