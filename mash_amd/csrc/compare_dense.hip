// compare_dense.hip — gfx950: the pairs INSIDE a group of near-identical sketches, as bit-mask arithmetic.
//
// The inverted-index engine (compare_sparse.hip) pays per shared hash (discovery reads every run) and per candidate
// (a merge of ~s steps).  Inside a clade -- thousands of isolates of one species, rows that are near-copies of each
// other -- every pair is a candidate sharing ~s hashes, and both costs explode (one clade of 32 768 rows: 4.8e11
// shared hashes to read, 5.4e8 merges).  But such a group has a tiny UNIVERSE: the values held by at least two of
// its rows (about s of them for near-copies).  A row is then a bit mask over the universe plus its EXTRAS -- values no
// other row of the group holds, which can never be common and only count as union elements -- recorded by their gap
// (the number of universe values below them).  The loop of compareSketches (CommandDistance.cpp:347-385) counts the
// common values among the first s values of the union; in universe order:
//
//     common = popcount(A & B) over the universe positions e with f(e) < s,
//     f(e)   = union bits before e + extras of either row with gap <= e,        denom = min(s, |A u B|).
//
// A word of 64 universe positions is taken whole while the count at its end stays <= s (a popcount, two table
// reads); the word in which s is reached is resolved by bisection over its bit positions.  ~17 word steps per pair at
// s = 1000 instead of ~1100 merge steps.  The arithmetic is pinned against the reference's loop by a numpy model
// (tests/test_dense_model.py) and the kernels against the oracle (tests/test_gpu_parity.py).
//
// Groups are runs of CONSECUTIVE rows whose neighbours are related (collections are listed in taxonomic order); any
// set of rows would be correct, so the relatedness test is a sample.  Layout per group, in blocks of 128 rows:
// mask word w of the block's rows side by side (w * 128 + lane, u64), then the cumulative extra counts
// cx[w] = extras with gap < 64 w (w = 0 .. W, u16, the same way); extras as u16 gaps per row.
// host_compare.cpp::table_sparse_index builds this next to the inverted index, whose runs it then clips so that discovery
// sees only the partners OUTSIDE a row's group; run_compare_sparse launches dn_pairs_kernel after the fill.
// (The encode and pairs kernels also compile for tools/hipemu -- MG_HIP_EMU: work-items as fibers on the CPU,
//  tests/test_dense_emu.py; what needs rocPRIM or wave intrinsics of the hardware is left out of that build.)
#ifdef MG_HIP_EMU
#include "hipemu.h"
#else
#include <hip/hip_runtime.h>
#define MG_DYN_SHARED(T, name)                                   \
    extern __shared__ __align__(16) unsigned char name##_raw[]; \
    T *name = reinterpret_cast<T *>(name##_raw)
#endif
#include <stdint.h>

#include <cstdlib>

#ifndef MG_HIP_EMU
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#endif

#include "compare_internal.h"

namespace mg {

#ifndef MG_HIP_EMU
// ------------------------------------------------------------------------------------------------
// which neighbours are related: link[i] = 1 if at least half of the first 64 values of row i occur among the first
// 256 of row i - 1 (the smallest hashes of a sketch are as good a sample as any)
__global__ __launch_bounds__(256) void dn_neighbor_kernel(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt, uint32_t n,
                                                          uint8_t *link)
{
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (row >= n) return;
    uint32_t hit = 0;
    uint32_t ci = 0, cj = 0;
    if (row > 0) {
        ci = cnt[row];
        cj = cnt[row - 1];
        const uint32_t take = ci < 64u ? ci : 64u, span = cj < 256u ? cj : 256u;
        if (lane < take && span > 0) {
            const uint64_t v = hashes[(uint64_t)row * stride + lane];
            const uint64_t *q = hashes + (uint64_t)(row - 1) * stride;
            uint32_t lo = 0, hi = span;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (q[mid] < v) lo = mid + 1; else hi = mid;
            }
            hit = (lo < span && q[lo] == v) ? 1u : 0u;
        }
    }
    const uint32_t hits = (uint32_t)__popcll(__ballot(hit != 0));
    const uint32_t take = ci < 64u ? ci : 64u;
    if (lane == 0) link[row] = (row > 0 && take >= 16u && cj >= 16u && 2u * hits >= take) ? 1 : 0;
}

hipError_t launch_dense_neighbors(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt, uint32_t n, uint8_t *link,
                                  hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(dn_neighbor_kernel, dim3((n + 3u) / 4u), dim3(256), 0, stream, hashes, stride, cnt, n, link);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// universe of every group: the values held by at least two of its rows.  In the sorted index the rows of a value
// ascend, a group is a row interval, so a group's holders of a value sit side by side: the first of them is the
// value's LEADER if a second one follows.  Every leader leaves {group, value, its own sorted position} in a list
// (one atomic per wave for the slots); sorted by (group, value) that list is every group's universe, ascending --
// values as the starts of their runs in the sorted index, which is what a row's code says -- and beside every
// universe value the position of its leader: the first holder INSIDE the group, i.e. the end of the run of partners
// OUTSIDE the group for every other holder in the group (the clipped run, see dn_encode_kernel).
// (the leaders are appended to one of DN_SUBLISTS lists, by workgroup: a million waves adding to ONE counter took 8 ms on
//  C3, where nearly every wave holds a leader; dn_sublists_scan / dn_sublists_copy then make the lists one)
constexpr uint32_t DN_SUBLISTS = 1024;

__global__ __launch_bounds__(256) void dn_leaders_kernel(const uint32_t *sorted_rows, const uint32_t *gs_of, const uint32_t *gend,
                                                         const uint32_t *grp_of, const DenseGroup *groups, uint32_t E,
                                                         unsigned long long *key, uint32_t *val, uint32_t cap_sub, uint32_t *cnt)
{
    const uint32_t pos = blockIdx.x * 256u + threadIdx.x;
    bool lead = false;
    uint32_t g = 0xFFFFFFFFu, gs = 0;
    if (pos < E) {
        g = grp_of[sorted_rows[pos]];
        if (g != 0xFFFFFFFFu) {
            const uint32_t g0 = groups[g].g0, g1 = groups[g].g1;
            gs = gs_of[pos];
            const bool first = pos == gs || sorted_rows[pos - 1] < g0;
            const bool more = pos + 1u < gend[gs] && sorted_rows[pos + 1] < g1;
            lead = first && more;
        }
    }
    const uint64_t bal = __ballot(lead);
    if (bal == 0) return;
    const uint32_t lane = threadIdx.x & 63u, sub = blockIdx.x & (DN_SUBLISTS - 1u);
    uint32_t base = 0;
    if (lane == (uint32_t)__builtin_ctzll(bal)) base = atomicAdd(&cnt[sub], (uint32_t)__popcll(bal));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(bal));
    if (lead) {
        const uint32_t k = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (k < cap_sub) {
            const uint64_t slot = (uint64_t)sub * cap_sub + k;
            key[slot] = ((unsigned long long)g << 32) | gs;
            val[slot] = pos;
        }
    }
}

// one workgroup: where every list goes in the joined list; total[0] = leaders kept, total[1] = the longest list asked for
__global__ __launch_bounds__(DN_SUBLISTS) void dn_sublists_scan_kernel(const uint32_t *cnt, uint32_t cap_sub, uint32_t *off, uint32_t *total)
{
    __shared__ uint32_t s_w[DN_SUBLISTS / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wid = t >> 6;
    const uint32_t want = cnt[t], c = want < cap_sub ? want : cap_sub;
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t x = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += x;
    }
    if (lane == 63u) s_w[wid] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < wid; k++) base += s_w[k];
    off[t] = base + incl - c;
    if (t == DN_SUBLISTS - 1u) total[0] = base + incl;
    atomicMax(&total[1], want);
}

__global__ __launch_bounds__(256) void dn_sublists_copy_kernel(const unsigned long long *key, const uint32_t *val, const uint32_t *cnt,
                                                               const uint32_t *off, uint32_t cap_sub, unsigned long long *key_out,
                                                               uint32_t *val_out)
{
    const uint32_t sub = blockIdx.x;
    const uint32_t c = cnt[sub] < cap_sub ? cnt[sub] : cap_sub, o = off[sub];
    for (uint32_t k = threadIdx.x; k < c; k += 256u) {
        key_out[o + k] = key[(uint64_t)sub * cap_sub + k];
        val_out[o + k] = val[(uint64_t)sub * cap_sub + k];
    }
}

// after the sort by (group, value): the universes' values and leader positions side by side, and where every group's
// universe starts and ends
__global__ __launch_bounds__(256) void dn_universe_split_kernel(const unsigned long long *key_sorted, uint32_t m, uint32_t *ulist,
                                                                uint32_t *ustart, uint32_t *uend)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const unsigned long long k = key_sorted[i];
    const uint32_t g = (uint32_t)(k >> 32);
    ulist[i] = (uint32_t)k;
    if (i == 0 || (uint32_t)(key_sorted[i - 1] >> 32) != g) ustart[g] = i;
    if (i + 1u == m || (uint32_t)(key_sorted[i + 1] >> 32) != g) uend[g] = i + 1u;
}

size_t dense_universe_temp_bytes(uint32_t cap)
{
    size_t b = 0;
    rocprim::radix_sort_pairs(nullptr, b, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (const uint32_t *)nullptr,
                              (uint32_t *)nullptr, (size_t)cap, 0u, 64u, (hipStream_t) nullptr);
    return b;
}

uint32_t dense_sublists() { return DN_SUBLISTS; }

// step 1: key / val: scratch of DN_SUBLISTS * cap_sub entries; key_out / val_out: the leaders joined (room for as many);
// cnt / off: scratch of DN_SUBLISTS u32; total (device, 2 u32): [0] leaders kept, [1] the longest list ASKED for -- beyond
// cap_sub leaders were dropped and the caller repeats with more room
hipError_t dense_find_leaders(const uint32_t *sorted_rows, const uint32_t *gs_of, const uint32_t *gend, const uint32_t *grp_of,
                              const DenseGroup *groups, uint32_t E, unsigned long long *key, uint32_t *val, uint32_t cap_sub,
                              unsigned long long *key_out, uint32_t *val_out, uint32_t *cnt, uint32_t *off, uint32_t *total,
                              hipStream_t stream)
{
    if (E == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(cnt, 0, DN_SUBLISTS * 4, stream);
    if (e == hipSuccess) e = hipMemsetAsync(total, 0, 8, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dn_leaders_kernel, dim3((E + 255u) / 256u), dim3(256), 0, stream, sorted_rows, gs_of, gend, grp_of, groups, E, key, val, cap_sub,
                       cnt);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return dense_join_leaders(key, val, cap_sub, key_out, val_out, cnt, off, total, stream);
}

// the lists of leaders made one (also behind the index build by tiles, whose bucket sorts find the leaders themselves:
// index_build.h, IxLeaders); total (device, 2 u32, zeroed by the caller): [0] leaders kept, [1] the longest list asked for
hipError_t dense_join_leaders(const unsigned long long *key, const uint32_t *val, uint32_t cap_sub, unsigned long long *key_out, uint32_t *val_out,
                              const uint32_t *cnt, uint32_t *off, uint32_t *total, hipStream_t stream)
{
    hipLaunchKernelGGL(dn_sublists_scan_kernel, dim3(1), dim3(DN_SUBLISTS), 0, stream, cnt, cap_sub, off, total);
    hipLaunchKernelGGL(dn_sublists_copy_kernel, dim3(DN_SUBLISTS), dim3(256), 0, stream, key, val, cnt, (const uint32_t *)off, cap_sub, key_out, val_out);
    return hipGetLastError();
}

// step 2 (the host knows m = *nlead <= cap): key_sorted: scratch of m u64; ulist / upos: out, the universes' values (run starts)
// and leader positions grouped by group, ascending inside a group; ustart / uend [ngroups]: zeroed by the caller
hipError_t dense_sort_universes(const unsigned long long *key, const uint32_t *val, uint32_t m, void *temp, size_t temp_bytes,
                                unsigned long long *key_sorted, uint32_t *ulist, uint32_t *upos, uint32_t *ustart, uint32_t *uend,
                                uint32_t group_bits, hipStream_t stream)
{
    if (m == 0) return hipSuccess;
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, key, key_sorted, val, upos, (size_t)m, 0u, 32u + group_bits, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dn_universe_split_kernel, dim3((m + 255u) / 256u), dim3(256), 0, stream, (const unsigned long long *)key_sorted, m, ulist,
                       ustart, uend);
    return hipGetLastError();
}

#endif  // !MG_HIP_EMU

// ------------------------------------------------------------------------------------------------
// a grouped row -> its mask words, cumulative extra counts, extras -- and its CLIPPED runs.  One workgroup per row, the
// group's universe staged in LDS, 256 entries at a time, each located in the universe by bisection.
//  * a value of the universe sets its bit; the end of its run of partners becomes the position of the value's leader --
//    the first holder inside the group: of the rows below this one that hold the value, discovery is to see those
//    OUTSIDE the group only (the pairs inside are this file's), and rows ascend inside a run;
//  * any other value is held by no other row of the group (its run is untouched: whoever holds it lies outside) and is
//    an EXTRA, recorded by its gap -- the number of universe values below it: in order in `ext`, counted per word in
//    cx, and per word as four bit PLANES over the gap's offset in the word (in the block: bit o of plane j = bit j of the number
//    of extras at offset o), which is what the resolve step of dn_pairs_kernel counts with; a word with sixteen extras
//    in one gap is flagged (bit 15 of the word's cx entry) and resolved from the list instead.  (Rounds 4-5a kept three
//    unary masks -- up to three extras per gap.  Loosely related clusters have dozens of extras per word: every word of
//    C3 was flagged, and the list walk was 3 000 instructions per pair.)
constexpr uint32_t DN_CX_MASK = 0x7FFFu, DN_CX_FLAG = 0x8000u;           // cx: a count (<= s <= 16384) | flag

template <uint32_t DN_EK>                                  // entries a work-item takes at a time
__global__ __launch_bounds__(256) void dn_encode_kernel(const uint32_t *off, const uint32_t *code_img, uint32_t *pos_img, uint32_t rs,
                                                        const uint32_t *grp_of, const DenseGroup *groups, const uint32_t *ulist,
                                                        const uint32_t *upos, unsigned long long *gdata, uint32_t wstride, uint16_t *ext,
                                                        uint32_t xs, uint32_t n, uint32_t ul_in_lds)
{
    MG_DYN_SHARED(uint32_t, lds);          // [2 W] mask halves, [W + 1] extras per word, [16 W] extras per gap (bytes), [W] overflow flags, [8 W] plane halves, [8] scratch, [u] the universe
    // (the hardware deals workgroups to the eight XCDs round-robin; a row's words go to one lane of its block's lines, eight
    //  bytes a kilobyte apart: of 64 workgroups that follow each other every XCD takes eight CONSECUTIVE rows, so that its L2
    //  holds 64 contiguous bytes of every line instead of every eighth lane)
    const uint32_t bid = blockIdx.x;
    const uint32_t row = (bid & ~63u) | ((bid & 7u) << 3) | ((bid >> 3) & 7u);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    if (row >= n) return;
    const uint32_t g = grp_of[row];
    if (g == 0xFFFFFFFFu) return;                        // uniform
    const DenseGroup G = groups[g];
    const uint32_t W = G.W, u = G.u;
    uint32_t *mask32 = lds, *hist = lds + 2u * W, *gap8 = hist + W + 1u, *ovf = gap8 + 16u * W, *pl32 = ovf + W, *scr = pl32 + 8u * W;
    uint32_t *ull = lds + 28u * wstride + 9u;
    for (uint32_t i = tid; i < 28u * W + 1u; i += 256u) lds[i] = 0;
    const uint32_t *ulg = ulist + G.ustart, *up = upos + G.ustart;
    if (ul_in_lds)
        for (uint32_t i = tid; i < u; i += 256u) ull[i] = ulg[i];
    const uint32_t *ul = ul_in_lds ? ull : ulg;
    __syncthreads();
    const uint32_t cnt = off[row + 1] - off[row];
    uint16_t *xrow = ext + (uint64_t)(G.xrow0 + (row - G.g0)) * xs;
    uint32_t nx = 0;                                     // extras written so far (uniform)
    if constexpr (DN_EK == 1) {
        // sketches of a thousand values: an entry per work-item, located in the universe by bisection (a round of 256 consecutive
        // entries reads and writes whole lines; with eight entries per work-item half a workgroup would idle)
        for (uint32_t base = 0; base < cnt; base += 256u) {
            const uint32_t p = base + tid;
            uint32_t idx = 0;
            bool extra = false;
            if (p < cnt) {
                const uint64_t img = (uint64_t)row * rs + p;
                const uint32_t gs = code_img[img] >> 1;
                uint32_t lo = 0, hi = u;                      // lower bound of gs in the universe
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (ul[mid] < gs) lo = mid + 1; else hi = mid;
                }
                idx = lo;
                if (lo < u && ul[lo] == gs) {
                    atomicOr(&mask32[idx >> 5], 1u << (idx & 31u));
                    pos_img[img] = up[idx];                  // (the leader itself: its own position again)
                } else {
                    extra = true;
                }
            }
            // extras keep their order (entries ascend, so gaps ascend): block-wide exclusive scan of the flags
            const uint64_t bal = __ballot(extra);
            const uint32_t before = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) scr[wid] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t woff = 0;
            for (uint32_t k = 0; k < wid; k++) woff += scr[k];
            const uint32_t total = scr[0] + scr[1] + scr[2] + scr[3];
            if (extra) {
                xrow[nx + woff + before] = (uint16_t)idx;
                atomicAdd(&hist[idx >> 6], 1u);
                // ... and per gap: a byte each (a word with 256 extras and more is flagged whatever its bytes say, see below)
                atomicAdd(&gap8[idx >> 2], 1u << (8u * (idx & 3u)));
            }
            nx += total;
            __syncthreads();                                 // scr is reused
        }
    } else {
        // A work-item takes DN_EK consecutive entries: the first is located in the universe by bisection, the others by walking
        // on from there -- the row ascends and so does the universe, and a near-copy holds almost every value of it, so the
        // next entry is a step or two away (a bisection per entry was fourteen dependent LDS reads at s = 10 000: 13.6 of C5's 95 ms).
        for (uint32_t base = 0; base < cnt; base += 256u * DN_EK) {
            const uint32_t p0 = base + tid * DN_EK;
            const uint32_t nv = p0 < cnt ? (cnt - p0 < DN_EK ? cnt - p0 : DN_EK) : 0u;
            uint32_t xidx[DN_EK];
            uint32_t nxt = 0;                                // this work-item's extras, in order: xidx[0 .. nxt)
            if (nv) {
                const uint64_t img0 = (uint64_t)row * rs + p0;
                // (16-byte loads: rs and p0 are multiples of 4, and the image has room behind every row's last entry)
                const uint4 c0 = *reinterpret_cast<const uint4 *>(code_img + img0), c1 = *reinterpret_cast<const uint4 *>(code_img + img0 + 4u);
                const uint32_t code[DN_EK] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                // (the entries' clipped run ends leave as two 16-byte stores: eight 4-byte stores per work-item, 32 bytes apart
                //  from the next lane's, were a sector request each)
                uint4 *pp = reinterpret_cast<uint4 *>(pos_img + img0);
                const uint4 q0 = pp[0], q1 = pp[1];
                uint32_t pv[DN_EK] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                uint32_t pos = 0;
                {
                    const uint32_t gs0 = code[0] >> 1;
                    uint32_t lo = 0, hi = u;                  // lower bound of the first entry in the universe
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (ul[mid] < gs0) lo = mid + 1; else hi = mid;
                    }
                    pos = lo;
                }
    #pragma unroll
                for (uint32_t e = 0; e < DN_EK; e++) {
                    if (e < nv) {
                        const uint32_t gs = code[e] >> 1;
                        {   // lower bound of gs from `pos` on: galloping (a step or two for a near-copy, logarithmic for a
                            // short row in a wide universe), then bisection of the last stride
                            uint32_t lo = pos, hi = pos, step = 1;           // everything before lo is below gs
                            while (hi < u && ul[hi] < gs) {
                                lo = hi + 1u;
                                hi += step;
                                step <<= 1;
                            }
                            hi = hi < u ? hi : u;
                            while (lo < hi) {
                                const uint32_t mid = (lo + hi) >> 1;
                                if (ul[mid] < gs) lo = mid + 1u; else hi = mid;
                            }
                            pos = lo;
                        }
                        if (pos < u && ul[pos] == gs) {
                            atomicOr(&mask32[pos >> 5], 1u << (pos & 31u));
                            pv[e] = up[pos];                 // (the leader itself: its own position again)
                        } else {
    #pragma unroll
                            for (uint32_t j = 0; j < DN_EK; j++)
                                if (j == nxt) xidx[j] = pos;  // (static indices: the array stays in registers)
                            nxt++;
                        }
                    }
                }
                // (behind the row's last entry the position image holds nothing anybody reads: written back as it was read)
                pp[0] = make_uint4(pv[0], pv[1], pv[2], pv[3]);
                pp[1] = make_uint4(pv[4], pv[5], pv[6], pv[7]);
            }
            // extras keep their order (entries ascend, so gaps ascend): block-wide exclusive scan of the work-items' counts
            uint32_t incl = nxt;
    #pragma unroll
            for (uint32_t d = 1; d < 64u; d <<= 1) {
                const uint32_t y = __shfl_up(incl, d);
                if (lane >= d) incl += y;
            }
            if (lane == 63u) scr[wid] = incl;
            __syncthreads();
            uint32_t woff = 0;
            for (uint32_t k = 0; k < wid; k++) woff += scr[k];
            const uint32_t total = scr[0] + scr[1] + scr[2] + scr[3];
            const uint32_t at0 = nx + woff + incl - nxt;
    #pragma unroll
            for (uint32_t j = 0; j < DN_EK; j++) {
                if (j < nxt) {
                    const uint32_t idx = xidx[j];
                    xrow[at0 + j] = (uint16_t)idx;
                    atomicAdd(&hist[idx >> 6], 1u);
                    // ... and per gap: a byte each (a word with 256 extras and more is flagged whatever its bytes say, see below)
                    atomicAdd(&gap8[idx >> 2], 1u << (8u * (idx & 3u)));
                }
            }
            nx += total;
            __syncthreads();                                 // scr is reused
        }
    }
    __syncthreads();
    const uint32_t jb = (row - G.g0) >> 7, bl = (row - G.g0) & 127u;
    const uint64_t bw = dense_block_words(W);
    unsigned long long *blk = gdata + G.data_off + (uint64_t)jb * bw;
    uint16_t *cxp = reinterpret_cast<uint16_t *>(blk + 128ull * W), *totp = cxp + (W + 1u) * 128u;
    unsigned long long *xpl = blk + dense_block_planes(W);                                // [W][4][128]
    for (uint32_t i = tid; i < 16u * W; i += 256u) {      // the gaps' counts as four bit planes: a work-item takes four gaps
        const uint32_t four = gap8[i], w = i >> 4, q = i & 15u;
        uint32_t bits[4] = {0u, 0u, 0u, 0u}, over = 0;
#pragma unroll
        for (uint32_t e = 0; e < 4u; e++) {
            const uint32_t c = (four >> (8u * e)) & 255u;
            over |= c >= 16u ? 1u : 0u;
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++) bits[j] |= ((c >> j) & 1u) << ((4u * q + e) & 31u);
        }
        if (over || (q == 0 && hist[w] >= 256u)) ovf[w] = 1u;
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++)
            if (bits[j]) atomicOr(&pl32[2u * (4u * w + j) + (q >> 3)], bits[j]);
    }
    __syncthreads();
    for (uint32_t i = tid; i < 4u * W; i += 256u) xpl[i * 128u + bl] = (unsigned long long)pl32[2u * i] | ((unsigned long long)pl32[2u * i + 1u] << 32);
    if (tid == 0) {                                       // cumulative counts: cx[w] = extras with gap < 64 w (w = 0 .. W),
        uint32_t run = 0, all = 0;                        // entry w + 1 carries the flag of word w; tot[w] = ALL the row's values
        for (uint32_t w = 0; w <= W; w++) {               // before that boundary (universe values held + extras)
            cxp[w * 128u + bl] = (uint16_t)(run | ((w > 0 && ovf[w - 1u]) ? DN_CX_FLAG : 0u));
            totp[w * 128u + bl] = (uint16_t)all;
            if (w < W) {
                run += hist[w];
                all += hist[w] + (uint32_t)__popc(mask32[2u * w]) + (uint32_t)__popc(mask32[2u * w + 1u]);
            }
        }
    }
    for (uint32_t w = tid; w < W; w += 256u)
        blk[w * 128u + bl] = (unsigned long long)mask32[2u * w] | ((unsigned long long)mask32[2u * w + 1u] << 32);
}

hipError_t launch_dense_encode(const uint32_t *off, const uint32_t *code_img, uint32_t *pos_img, uint32_t rs, const uint32_t *grp_of,
                               const DenseGroup *groups, const uint32_t *ulist, const uint32_t *upos, unsigned long long *gdata,
                               uint16_t *ext, uint32_t xs, uint32_t n, uint32_t wmax, hipStream_t stream, int ul_mode)
{
    if (n == 0) return hipSuccess;
    // The universe in LDS while that leaves a CU several workgroups: a universe of 15 000 values (C5: 86 KB with the masks) made it
    // ONE workgroup of four waves per CU, each staging 60 KB for one row -- read from the L2 instead (the row ascends, so do its
    // probes) the kernel keeps five workgroups per CU: C5 per table 92.3 -> 88.1 ms; C3's 1 500 values stay staged (15.0 against
    // 14.8 ms unstaged).  ul_mode 0 / 1 (MASHGPU_DENSE_UL_LDS): never / whenever it fits at all.
    const size_t fixed = ((size_t)28 * wmax + 9) * 4, with_ul = fixed + (size_t)wmax * 64 * 4;
    const bool ul_in_lds = ul_mode == 0 ? false : ul_mode == 1 ? with_ul <= 150 * 1024 : with_ul <= 40 * 1024;
    const size_t smem = ul_in_lds ? with_ul : fixed;
    // (measured: eight entries per work-item 13.6 -> 11.4 ms at s = 10 000, 0.72 -> 0.86 ms at s = 1 000)
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((n + 63u) & ~63u), dim3(256), smem, stream, off, code_img, pos_img, rs, grp_of, groups, ulist, upos, gdata, wmax, ext, xs,
                           n, ul_in_lds ? 1u : 0u);
        return hipGetLastError();
    };
    return rs >= 4096u ? go(dn_encode_kernel<8>) : go(dn_encode_kernel<1>);
}

// ------------------------------------------------------------------------------------------------
// the pairs of a tile: DN_ROWS rows x the 128 columns of one block of the group, lane = column (so that a row's
// results leave as one contiguous store per wave).  The column block's words are staged in LDS as they lie in
// memory (word w of lane l at w * 128 + l: conflict free), the rows' words beside them (read by all lanes at once).

// Where a pair of index rows lands in the output: the index may have been built on a PERMUTED table (rows that belong
// together next to each other, see dense_cluster_rows); inv maps an index row back to the table's row, and the pair of table
// rows i > j stands at i (i - 1) / 2 - out_base + j.

// popcount(x) + acc as ONE instruction (the compiler adds three terms with v_add3 behind two counts into zero)
__device__ __forceinline__ uint32_t dn_count_add(uint32_t x, uint32_t acc)
{
#ifdef MG_HIP_EMU
    return (uint32_t)__popc(x) + acc;
#else
    uint32_t d;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
    return d;
#endif
}

// the low y bits of x (y = 0 .. 31)
__device__ __forceinline__ uint32_t dn_low_bits(uint32_t x, uint32_t y)
{
#ifdef MG_HIP_EMU
    return x & ((1u << y) - 1u);
#else
    return __builtin_amdgcn_ubfe(x, 0u, y);
#endif
}

// One half (32 offsets) of the word in which the union reaches its s-th element: the number of offsets x in 0 .. 30 with
// f(x) < s, bit by bit from the top (f never falls) -- the smallest offset at which s is reached, or 31.
// f(x) = base + the union's bits below x + the extras of either row at offsets <= x; u1 is the union's half shifted up by
// one, so that both are "the low x + 1 bits".  NP: the planes above the first that hold anything (0, 1 or 3).
template <uint32_t NP>
__device__ __forceinline__ uint32_t dn_descend(uint32_t base, uint32_t s, uint32_t u1, uint32_t A0, uint32_t B0, uint32_t A1, uint32_t B1,
                                               uint32_t A2, uint32_t B2, uint32_t A3, uint32_t B3)
{
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t bit = 16u; bit != 0u; bit >>= 1) {
        const uint32_t y = pos + bit;                                     // offset y - 1 is tested
        uint32_t c = dn_count_add(dn_low_bits(B0, y), dn_count_add(dn_low_bits(A0, y), dn_count_add(dn_low_bits(u1, y), base)));
        if (NP >= 1u) c += 2u * dn_count_add(dn_low_bits(B1, y), (uint32_t)__popc(dn_low_bits(A1, y)));
        if (NP >= 3u)
            c += 4u * dn_count_add(dn_low_bits(B2, y), (uint32_t)__popc(dn_low_bits(A2, y))) +
                 8u * dn_count_add(dn_low_bits(B3, y), (uint32_t)__popc(dn_low_bits(A3, y)));
        pos = c < s ? y : pos;
    }
    return pos;
}

// The word in which the union reaches its s-th element: the smallest bit position t with f(t) >= s, where f(t) = what lies
// before the word (fprev) + union bits below t + extras of either row with offset <= t; the COMMON bits below t are what comes
// back (mab: the two rows' masks ANDed).
// The extras of the word as four bit planes per row (bit o of plane j: bit j of the number of extras at offset o), or -- a gap
// of the word holds sixteen and more in one of the rows -- from the rows' lists.
// Called by every lane of the wave (uniform control flow: the ballots below are taken over all of them); a lane with nothing
// to resolve passes lists = false, na = nb = 0 and a word that exists, and ignores what comes back.
__device__ __forceinline__ uint32_t dn_resolve(unsigned long long un, unsigned long long mab, uint32_t fprev, uint32_t s, uint32_t w, bool lists,
                                               const unsigned long long *xpa, const unsigned long long *xpb, const uint16_t *xa,
                                               const uint16_t *xb, uint32_t ca0, uint32_t na, uint32_t cb0, uint32_t nb)
{
    if (__ballot(lists) == 0) {                              // uniform
        // (xpa / xpb: the row's and the column's lane in the planes of their blocks, [W][4][128]: lanes that resolve the same
        //  word -- near-copies nearly all do -- read a line per plane between them)
        const unsigned long long a0 = xpa[512u * w], a1 = xpa[512u * w + 128u], a2 = xpa[512u * w + 256u], a3 = xpa[512u * w + 384u];
        const unsigned long long b0 = xpb[512u * w], b1 = xpb[512u * w + 128u], b2 = xpb[512u * w + 256u], b3 = xpb[512u * w + 384u];
        // Which half of the word: f(31) = fprev + the union's bits 0 .. 30 + the extras at offsets 0 .. 31; from there on every
        // count is over 32-bit halves (a 64-bit mask and count is two shifts with carries and two counts per operand).
        // (near-copies have a few dozen extras over a thousand gaps: where no gap of the whole wave holds two, a count is three
        //  population counts, five where none holds four, else nine)
        const bool p1 = __ballot((a1 | b1) != 0ull) != 0, p23 = __ballot((a2 | a3 | b2 | b3) != 0ull) != 0;      // uniform
        const uint32_t ulo = (uint32_t)un, uhi = (uint32_t)(un >> 32);
        uint32_t c31 = dn_count_add((uint32_t)b0, dn_count_add((uint32_t)a0, dn_count_add(ulo & 0x7FFFFFFFu, fprev)));
        if (p1 || p23) c31 += 2u * dn_count_add((uint32_t)b1, (uint32_t)__popc((uint32_t)a1));
        if (p23)
            c31 += 4u * dn_count_add((uint32_t)b2, (uint32_t)__popc((uint32_t)a2)) + 8u * dn_count_add((uint32_t)b3, (uint32_t)__popc((uint32_t)a3));
        const bool upper = c31 < s;                          // s is reached at offset 32 or above: the lower half lies before
        const uint32_t base = upper ? c31 + (ulo >> 31) : fprev;
        const uint32_t u1 = (upper ? uhi : ulo) << 1;
        const uint32_t A0 = upper ? (uint32_t)(a0 >> 32) : (uint32_t)a0, B0 = upper ? (uint32_t)(b0 >> 32) : (uint32_t)b0;
        uint32_t pos;
        if (!p1 && !p23) {
            pos = dn_descend<0>(base, s, u1, A0, B0, 0, 0, 0, 0, 0, 0);
        } else {
            const uint32_t A1 = upper ? (uint32_t)(a1 >> 32) : (uint32_t)a1, B1 = upper ? (uint32_t)(b1 >> 32) : (uint32_t)b1;
            if (!p23) {
                pos = dn_descend<1>(base, s, u1, A0, B0, A1, B1, 0, 0, 0, 0);
            } else {
                const uint32_t A2 = upper ? (uint32_t)(a2 >> 32) : (uint32_t)a2, B2 = upper ? (uint32_t)(b2 >> 32) : (uint32_t)b2;
                const uint32_t A3 = upper ? (uint32_t)(a3 >> 32) : (uint32_t)a3, B3 = upper ? (uint32_t)(b3 >> 32) : (uint32_t)b3;
                pos = dn_descend<3>(base, s, u1, A0, B0, A1, B1, A2, B2, A3, B3);
            }
        }
        const uint32_t mlo = (uint32_t)mab, mhi = (uint32_t)(mab >> 32);
        return upper ? dn_count_add(dn_low_bits(mhi, pos), (uint32_t)__popc(mlo)) : (uint32_t)__popc(dn_low_bits(mlo, pos));
    }
    uint32_t lo = 0, hi = 63;
    // (some lane's word is flagged: every lane takes its rows' lists, which are exact for all)
    const uint32_t wbase = w << 6;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        uint32_t c = fprev + (uint32_t)__popcll(un & ((1ull << mid) - 1ull));
        for (uint32_t k = 0; k < na; k++) c += ((uint32_t)xa[ca0 + k] - wbase <= mid) ? 1u : 0u;
        for (uint32_t k = 0; k < nb; k++) c += ((uint32_t)xb[cb0 + k] - wbase <= mid) ? 1u : 0u;
        if (c >= s) hi = mid; else lo = mid + 1u;
    }
    return (uint32_t)__popcll(mab & ((1ull << lo) - 1ull));
}

// The column block's words are read from global memory (L2) word by word, the next word requested before the current one
// is worked on; only the rows' words sit in LDS.  (Staging the block in LDS was measured slower for every width: 26 KB per
// two waves leave a CU a handful of waves, and the loop lives on having many -- C3 1.47 against 1.10 ms, one clade of
// 32 768 rows 17.8 against 12.6 ms; universes of hundreds of words, s = 10 000, would not fit anyway.)
template <uint32_t DN_ROWS, uint32_t DN_IL>                // DN_IL: rows a wave works on side by side
__global__ __launch_bounds__(128) void dn_pairs_kernel(const DenseTile *tiles, const DenseGroup *groups, const unsigned long long *gdata,
                                                       uint32_t use_lists,
                                                       const uint16_t *ext, uint32_t xs, uint32_t s, uint32_t row_begin, uint32_t row_end,
                                                       uint64_t out_base, const uint32_t *inv, uint2 *out, DenseList list)
{
    MG_DYN_SHARED(unsigned long long, dl);
    const DenseTile T = tiles[blockIdx.x];
    const DenseGroup G = groups[T.group];
    const uint32_t W = G.W, tid = threadIdx.x;
    const uint64_t bw = dense_block_words(W);
    const unsigned long long *Bm = gdata + G.data_off + (uint64_t)T.cblk * bw;            // [W][128]
    const uint16_t *Bcx = reinterpret_cast<const uint16_t *>(Bm + 128ull * W);             // [W + 1][128]
    const uint16_t *Btot = Bcx + (W + 1u) * 128u;                                         // [W + 1][128]
    unsigned long long *Am = dl;                                                          // [W][DN_ROWS]
    uint16_t *Acx = reinterpret_cast<uint16_t *>(Am + (uint64_t)DN_ROWS * W);             // [W + 1][DN_ROWS]
    uint16_t *Atot = Acx + (W + 1u) * DN_ROWS;                                            // [W + 1][DN_ROWS]
    const uint32_t ra = T.row0 - G.g0;                                    // (a multiple of DN_ROWS: the tile's rows share a block)
    const unsigned long long *asrc = gdata + G.data_off + (uint64_t)(ra >> 7) * bw;
    const uint32_t la0 = ra & 127u;
    for (uint32_t i = tid; i < W * DN_ROWS; i += 128u) Am[i] = asrc[(i / DN_ROWS) * 128u + la0 + (i % DN_ROWS)];
    {
        // (cx and tot lie behind each other in the block and in LDS: one loop over 2 (W + 1) lines)
        const uint16_t *acs = reinterpret_cast<const uint16_t *>(asrc + 128ull * W);
        for (uint32_t i = tid; i < 2u * (W + 1u) * DN_ROWS; i += 128u) Acx[i] = acs[(i / DN_ROWS) * 128u + la0 + (i % DN_ROWS)];
    }
    __syncthreads();
    // the largest total of every batch of DN_IL rows before every word boundary: while that and the column's total stay
    // within s, no pair of the batch has reached its s-th union element whatever its rows have in common
    constexpr uint32_t DN_NB = DN_ROWS / DN_IL;
    uint16_t *Amax = Atot + (W + 1u) * DN_ROWS;                                           // [DN_NB][W + 1]
    for (uint32_t i = tid; i < DN_NB * (W + 1u); i += 128u) {
        const uint32_t bt = i / (W + 1u), w = i - bt * (W + 1u);
        uint32_t m = 0;
        for (uint32_t k = 0; k < DN_IL; k++)
            if (T.row0 + bt * DN_IL + k < G.g1) {                          // (rows behind the group's end are not written)
                const uint32_t v = Atot[w * DN_ROWS + bt * DN_IL + k];
                m = v > m ? v : m;
            }
        Amax[i] = (uint16_t)m;
    }
    __syncthreads();
    const uint32_t b = G.g0 + T.cblk * 128u + tid;                         // this lane's column
    // (a lane behind the group's end works on the group's last row: what it reads is written, what it computes is not stored)
    const uint32_t lb = b < G.g1 ? tid : G.g1 - 1u - (G.g0 + T.cblk * 128u);
    const uint16_t *xb = ext + (uint64_t)(G.xrow0 + (b < G.g1 ? b - G.g0 : 0u)) * xs;
    const unsigned long long *xpb = Bm + dense_block_planes(W) + lb;
    // where this lane's pairs land: the column's table row and its triangle base once per tile -- per pair a
    // comparison and an addition are left, the row's share is uniform
    const uint32_t colrow = inv ? inv[G.g0 + T.cblk * 128u + lb] : G.g0 + T.cblk * 128u + lb;
    const uint64_t tri_col = (colrow ? (uint64_t)colrow * (colrow - 1u) / 2u : 0ull) - out_base;
    // DN_IL rows at a time: the column block's word is loaded once and serves all of them (their words are broadcast reads
    // from LDS), and a wave has several independent pairs per lane in flight instead of one.
    // Per pair and word: the intersection's bits are counted (common), and what the union holds up to the word's end follows
    // from the two rows' running totals -- |A u B| = |A| + |B| - |A n B| -- instead of from a second population count:
    // F = totA + totB - common.  F never falls from word to word, so the loop needs no state but two counters per pair:
    // while F <= s the word lies before the s-th union element and its common bits count; the number of such words IS the
    // word in which F passes s.  That word is resolved bit by bit behind the loop (dn_resolve), once per pair and with all
    // lanes of the wave at it together.  (A pair whose F equals s at a word's end meets the resolve with nothing left to
    // count, and one that is not a pair of the job -- a column not below the row -- is computed like the others and not stored.)
    // The first words -- about half of them for near-copies: totA + totB reaches s where either row holds s / 2 -- need no
    // comparison at all (Amax): two ANDs and two counts per pair and word, ten instructions afterwards.
    for (uint32_t ai = 0; ai < DN_ROWS; ai += DN_IL) {
        if (T.row0 + ai >= G.g1 || T.row0 + ai >= row_end) break;         // uniform
        const uint16_t *amax = Amax + (ai / DN_IL) * (W + 1u);
        uint32_t common[DN_IL], nle[DN_IL];
#pragma unroll
        for (uint32_t k = 0; k < DN_IL; k++) common[k] = nle[k] = 0;
        unsigned long long mb_next = Bm[lb];
        uint32_t tb_next = Btot[128u + lb];
        uint32_t wi = 0;
        // the words no pair of the batch can end in (tb_next: the column's total at the end of word wi)
        for (; wi < W; wi++) {
            if (__ballot((uint32_t)amax[wi + 1u] + tb_next > s) != 0) break;          // uniform
            const unsigned long long mb = mb_next;
            if (wi + 1u < W) {                                             // the next word is on its way while this one is worked on
                mb_next = Bm[(wi + 1u) * 128u + lb];
                tb_next = Btot[(wi + 2u) * 128u + lb];
            }
            const uint32_t mlo = (uint32_t)mb, mhi = (uint32_t)(mb >> 32);
#pragma unroll
            for (uint32_t k = 0; k < DN_IL; k++) {
                const unsigned long long am = Am[wi * DN_ROWS + ai + k];
                common[k] = dn_count_add((uint32_t)am & mlo, dn_count_add((uint32_t)(am >> 32) & mhi, common[k]));
            }
        }
        const uint32_t nfast = wi;
        for (; wi < W; wi++) {
            const unsigned long long mb = mb_next;
            const int32_t tbs = (int32_t)tb_next - (int32_t)s;            // F <= s  <=>  totA + (totB - s) <= common
            if (wi + 1u < W) {
                mb_next = Bm[(wi + 1u) * 128u + lb];
                tb_next = Btot[(wi + 2u) * 128u + lb];
            }
            const uint32_t mlo = (uint32_t)mb, mhi = (uint32_t)(mb >> 32);
            bool any = false;
#pragma unroll
            for (uint32_t k = 0; k < DN_IL; k++) {
                const unsigned long long am = Am[wi * DN_ROWS + ai + k];
                const uint32_t cnew = dn_count_add((uint32_t)am & mlo, dn_count_add((uint32_t)(am >> 32) & mhi, common[k]));
                const bool before = (int32_t)Atot[(wi + 1u) * DN_ROWS + ai + k] + tbs <= (int32_t)cnew;
                common[k] = before ? cnew : common[k];
                nle[k] += before ? 1u : 0u;
                any = any || before;
            }
            if (__ballot(any) == 0) break;                                // uniform: every pair is past its s-th element
        }
#pragma unroll
        for (uint32_t k = 0; k < DN_IL; k++) {
            const uint32_t a = T.row0 + ai + k;
            // (uniform: the row exists, belongs to the job, and the column block starts below it)
            const bool present = a < G.g1 && a < row_end && a >= row_begin && G.g0 + T.cblk * 128u < a;
            if (!present) continue;                                       // uniform
            const bool valid = b < a;
            uint32_t denom = s;
            const uint32_t w = nfast + nle[k];
            const bool need = valid && w < W;                             // s is reached inside word w
            if (__ballot(need) != 0) {                                    // uniform: the resolve is entered by the whole wave
                const uint32_t wc = need ? w : 0u;
                const unsigned long long ma = Am[wc * DN_ROWS + ai + k], mb = Bm[wc * 128u + lb];
                const uint32_t ca0 = Acx[wc * DN_ROWS + ai + k] & DN_CX_MASK, ca1r = Acx[(wc + 1u) * DN_ROWS + ai + k];
                const uint32_t cb0 = Bcx[wc * 128u + lb] & DN_CX_MASK, cb1r = Bcx[(wc + 1u) * 128u + lb];
                const uint32_t fprev = (uint32_t)Atot[wc * DN_ROWS + ai + k] + (uint32_t)Btot[wc * 128u + lb] - common[k];
                const bool lists = need && (use_lists || ((ca1r | cb1r) & DN_CX_FLAG) != 0);
                const uint32_t add = dn_resolve(ma | mb, ma & mb, fprev, s, wc, lists, asrc + dense_block_planes(W) + la0 + ai + k, xpb,
                                               ext + (uint64_t)(G.xrow0 + (a - G.g0)) * xs, xb, ca0, need ? (ca1r & DN_CX_MASK) - ca0 : 0u, cb0,
                                               need ? (cb1r & DN_CX_MASK) - cb0 : 0u);
                if (need) common[k] += add;
            }
            if (!valid) continue;
            if (w >= W) {                                                 // the union ends before s (short sketches)
                const uint32_t total = (uint32_t)Atot[W * DN_ROWS + ai + k] + (uint32_t)Btot[W * 128u + lb] - common[k];
                denom = total < s ? total : s;
            }
            if (list.rc) {                                                // (uniform) a list job: the tail of row a's list
                const uint32_t ra = a - list.row_first;
                const uint32_t at = list.row_base[ra] + list.row_cnt[ra] - (a - b);
                list.rc[at] = make_uint2(a, b);
                list.counts[at] = make_uint2(common[k], denom);
            } else {
                const uint32_t arow = inv ? inv[a] : a;                   // (uniform)
                const uint64_t tri_a = (uint64_t)arow * (arow - 1u) / 2u - out_base;
                out[arow > colrow ? tri_a + colrow : tri_col + arow] = make_uint2(common[k], denom);
            }
        }
    }
}

// LDS of a tile: its rows' words and counts
uint32_t dense_max_words() { return 400; }                 // 32 rows x (8 W + 4 W + 4) bytes

size_t dense_pairs_lds(uint32_t W, uint32_t rows) { return (size_t)rows * W * 8 + (size_t)(W + 1u) * rows * 4 + (size_t)(W + 1u) * 8 * 2 + 16; }

// rows of a tile: 32 when that still leaves enough tiles to fill the device (a tile is two waves; the rows of a tile are taken
// one after the other), else 8 -- a collection of small clusters has few column blocks per row block
uint32_t dense_rows_per_tile(uint64_t wave_rows) { return wave_rows / 32u >= 16384u ? 32u : 8u; }

hipError_t launch_dense_pairs(const DenseTile *tiles, uint32_t ntiles, uint32_t rows_per_tile, const DenseGroup *groups,
                              const unsigned long long *gdata, bool use_lists, const uint16_t *ext, uint32_t xs,
                              uint32_t s, uint32_t wmax, uint32_t row_begin, uint32_t row_end, uint64_t out_base, const uint32_t *inv, uint2 *out,
                              hipStream_t stream, const DenseList *list)
{
    if (ntiles == 0) return hipSuccess;
    const size_t smem = dense_pairs_lds(wmax, rows_per_tile);
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(ntiles), dim3(128), smem, stream, tiles, groups, gdata, use_lists ? 1u : 0u, ext, xs, s, row_begin, row_end,
                           out_base, inv, out, list ? *list : DenseList());
        return hipGetLastError();
    };
    const uint32_t il = 8;                                 // (measured on one clade of 32 768 rows, warm: 7.8 / 7.3 / 7.2 ms at 4 / 8 / 16)
    if (rows_per_tile == 32u) return il >= 16u ? go(dn_pairs_kernel<32, 16>) : il >= 8u ? go(dn_pairs_kernel<32, 8>) : go(dn_pairs_kernel<32, 4>);
    return il >= 8u ? go(dn_pairs_kernel<8, 8>) : go(dn_pairs_kernel<8, 4>);
}

#ifndef MG_HIP_EMU
// ------------------------------------------------------------------------------------------------
// which rows belong together, whatever their order in the table (collections are not always listed by species): the
// LABEL of a row is the smallest row that holds one of its first four hashes -- rows of a clade agree on it with high
// probability (a member lacks all four of the clade's smallest values only rarely), two jumps label -> label of the
// label close the chains -- and the rows sorted by (label, row) are the order in which the index is built.  Any
// order would be correct; this one puts near-copies next to each other so that they form dense groups.
constexpr uint32_t CL_FIRST = 4;

__global__ __launch_bounds__(256) void cl_emit_kernel(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt, uint32_t n,
                                                      unsigned long long *key, uint32_t *row_out, uint32_t *label)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n * CL_FIRST) return;
    const uint32_t row = i / CL_FIRST, f = i % CL_FIRST;
    key[i] = f < cnt[row] ? (unsigned long long)hashes[(uint64_t)row * stride + f] : ~0ull;
    row_out[i] = row;
    if (f == 0) label[row] = row;
}

__global__ __launch_bounds__(256) void cl_minrow_kernel(const unsigned long long *key_sorted, const uint32_t *row_sorted, uint32_t m,
                                                        uint32_t *label)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const unsigned long long k = key_sorted[i];
    if (k == ~0ull) return;
    uint32_t lo = 0, hi = i;                               // first position holding k (the sort is stable: its row is the smallest)
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (key_sorted[mid] < k) lo = mid + 1; else hi = mid;
    }
    atomicMin(&label[row_sorted[i]], row_sorted[lo]);
}

__global__ __launch_bounds__(256) void cl_jump_kernel(const uint32_t *in, uint32_t *out, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = in[in[i]];
}

// (split: the rows from `split` on form a segment of their own behind the others -- a triangle job over the rows [split, n) then
//  finds its rows side by side in the index's order, and the rows of a cluster side by side inside each segment)
__global__ __launch_bounds__(256) void cl_order_keys_kernel(const uint32_t *label, uint32_t n, uint32_t split, unsigned long long *key)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) key[i] = ((unsigned long long)(i >= split ? 1u : 0u) << 63) | ((unsigned long long)label[i] << 32) | i;
}

__global__ __launch_bounds__(256) void cl_split_keys_kernel(const unsigned long long *key_sorted, uint32_t n, uint32_t *inv, uint32_t *label_sorted)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) {
        inv[i] = (uint32_t)(key_sorted[i] & 0xFFFFFFFFull);
        label_sorted[i] = (uint32_t)(key_sorted[i] >> 32) & 0x7FFFFFFFu;
    }
}

size_t dense_cluster_temp_bytes(uint32_t n)
{
    size_t a = 0, b = 0;
    rocprim::radix_sort_pairs(nullptr, a, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (const uint32_t *)nullptr,
                              (uint32_t *)nullptr, (size_t)n * CL_FIRST, 0u, 64u, (hipStream_t) nullptr);
    rocprim::radix_sort_keys(nullptr, b, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (size_t)n, 0u, 64u,
                             (hipStream_t) nullptr);
    return a > b ? a : b;
}

// scratch: key_a / key_b [4 n] u64, row_a / row_b [4 n] u32, lab_a / lab_b [n] u32.  Out: inv[n] (index row -> table row),
// label_sorted[n] (the label of every index row: equal labels = one cluster).
hipError_t dense_cluster_rows(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt, uint32_t n, void *temp, size_t temp_bytes,
                              unsigned long long *key_a, unsigned long long *key_b, uint32_t *row_a, uint32_t *row_b, uint32_t *lab_a,
                              uint32_t *lab_b, uint32_t *inv, uint32_t *label_sorted, hipStream_t stream, uint32_t split)
{
    if (n == 0) return hipSuccess;
    const uint32_t m = n * CL_FIRST;
    hipLaunchKernelGGL(cl_emit_kernel, dim3((m + 255u) / 256u), dim3(256), 0, stream, hashes, stride, cnt, n, key_a, row_a, lab_a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(temp, temp_bytes, (const unsigned long long *)key_a, key_b, (const uint32_t *)row_a, row_b, (size_t)m, 0u, 64u,
                                  stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cl_minrow_kernel, dim3((m + 255u) / 256u), dim3(256), 0, stream, (const unsigned long long *)key_b, (const uint32_t *)row_b, m,
                       lab_a);
    hipLaunchKernelGGL(cl_jump_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, (const uint32_t *)lab_a, lab_b, n);
    hipLaunchKernelGGL(cl_jump_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, (const uint32_t *)lab_b, lab_a, n);
    hipLaunchKernelGGL(cl_order_keys_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, (const uint32_t *)lab_a, n, split, key_a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_keys(temp, temp_bytes, (const unsigned long long *)key_a, key_b, (size_t)n, 0u, 64u, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cl_split_keys_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, (const unsigned long long *)key_b, n, inv, label_sorted);
    return hipGetLastError();
}

// the table in index order: out[a] = row inv[a] (whole rows, padding included)
__global__ __launch_bounds__(256) void cl_gather_rows_kernel(const uint64_t *hashes, uint64_t stride, const uint32_t *inv, uint64_t *out)
{
    const uint32_t a = blockIdx.x;
    const uint64_t *src = hashes + (uint64_t)inv[a] * stride;
    uint64_t *dst = out + (uint64_t)a * stride;
    for (uint64_t p = threadIdx.x; p < stride; p += 256u) dst[p] = src[p];
}

hipError_t launch_dense_gather_rows(const uint64_t *hashes, uint64_t stride, const uint32_t *inv, uint32_t n, uint64_t *out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(cl_gather_rows_kernel, dim3(n), dim3(256), 0, stream, hashes, stride, inv, out);
    return hipGetLastError();
}

#endif  // !MG_HIP_EMU

// Per row of the table the candidate group it stands in: grp_of[r] (0xFFFFFFFF: none) and lead_rows[r] = {group, its first
// row, one past its last, 0} -- what the index build's leader search gathers per entry (IxLeaders).  groups: disjoint,
// ascending.  (Made here from the few groups: 20 bytes per row that the host neither fills nor copies.)
__global__ __launch_bounds__(256) void dn_group_rows_kernel(const DenseGroup *groups, uint32_t ng, uint32_t n, uint32_t *grp_of, uint4 *lead_rows)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n) return;
    uint32_t lo = 0, hi = ng;                            // the groups below lo start at or before r, those from hi on behind it
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (groups[mid].g0 <= r) lo = mid + 1u;
        else hi = mid;
    }
    uint32_t g = 0xFFFFFFFFu, g0 = 0, g1 = 0;
    if (lo > 0 && r < groups[lo - 1u].g1) {
        g = lo - 1u;
        g0 = groups[g].g0;
        g1 = groups[g].g1;
    }
    grp_of[r] = g;
    lead_rows[r] = make_uint4(g, g0, g1, 0u);
}

hipError_t launch_dense_group_rows(const DenseGroup *groups, uint32_t ng, uint32_t n, uint32_t *grp_of, uint32_t *lead_rows, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(dn_group_rows_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, groups, ng, n, grp_of, reinterpret_cast<uint4 *>(lead_rows));
    return hipGetLastError();
}

}  // namespace mg
