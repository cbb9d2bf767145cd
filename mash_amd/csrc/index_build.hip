// index_build.hip -- gfx950: the inverted index of a sketch table without a general-purpose sort (see index_build.h for
// the plan: window offsets, counts, one partition pass over tiles of (512 rows x a window of buckets), an LDS sort per
// bucket that also finds the groups of equal values, and the tiles again to write the images back row segment by row segment).
//
// The kernels are written so that tools/hipemu/hipemu.h can run them on host threads (MG_HIP_EMU; tests/test_index_emu.py:
// every array against a std::stable_sort statement of the index, also under ThreadSanitizer): a workgroup leaves a kernel as a
// whole or not at all, wave operations sit in uniform control flow, and no step relies on the lock step of a wave.
#ifdef MG_HIP_EMU
#include "hipemu.h"
#else
#include <hip/hip_runtime.h>
#define MG_DYN_SHARED(T, name)                                   \
    extern __shared__ __align__(16) unsigned char name##_raw[]; \
    T *name = reinterpret_cast<T *>(name##_raw)
#endif
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "index_build.h"

// One LDS atomic instruction whose lanes may hit the same counter: the hardware serves them in lane order (that is relied
// on for SPEED only, see ix_bucket_sort_body); the emulator's lanes are independent fibers, so there the statement is
// executed lane by lane.
#ifdef MG_HIP_EMU
#define IX_IN_LANE_ORDER(stmt)                                   \
    for (uint32_t l_ = 0; l_ < 64u; l_++) {                     \
        if ((threadIdx.x & 63u) == l_) { stmt; }                \
        hipemu::wave_sync();                                     \
    }
#else
#define IX_IN_LANE_ORDER(stmt) { stmt; }
#endif

namespace mg {

// -DIX_PHASE_CLOCKS: work-item 0 of every workgroup adds the cycles between two marks of a kernel to a table (a tuning build:
// tools/r5_clocks.sh prints it; the marks compile to nothing otherwise)
#if defined(IX_PHASE_CLOCKS) && !defined(MG_HIP_EMU)
__device__ unsigned long long ix_clk[64];
#define IX_CLK_BEGIN() unsigned long long clk_prev_ = threadIdx.x == 0 ? clock64() : 0ull
#define IX_CLK(i)                                                          \
    if (threadIdx.x == 0) {                                               \
        const unsigned long long t_ = clock64();                          \
        atomicAdd(&ix_clk[i], t_ - clk_prev_);                            \
        clk_prev_ = t_;                                                   \
    }
#else
#define IX_CLK_BEGIN() ((void)0)
#define IX_CLK(i) ((void)0)
#endif

constexpr uint32_t IX_RB = 512;                 // rows of a block of rows
constexpr uint32_t IX_NT = 512;                 // work-items of the tile kernels (one per row of the block)
constexpr uint32_t IX_M = 7;                    // entries a work-item owns in the tile's sort (odd: its strided LDS reads spread over the banks)
constexpr uint32_t IX_PCAP = IX_NT * IX_M;      // entries of a tile sorted at a time (a larger tile is taken in pieces)
constexpr uint32_t IX_LPR = 8;                  // lanes that read one row's segment
constexpr uint32_t IX_BW_MAX = 512;             // buckets per window, at most
constexpr uint32_t IX_CAP = 6144;               // entries of a bucket the LDS sort takes
constexpr uint32_t IX_NT4 = 512;
constexpr uint32_t IX_SUBBITS = 13;             // the counting sort's key: the next 13 bits below the bucket
constexpr uint32_t IX_NSUB = 1u << IX_SUBBITS;
constexpr uint32_t IX_SUBLOW = IX_SUBBITS - 3u;      // a sub-bucket's bits below the three that name the wave owning it (IX_NT4 / 64 = 8 waves)
static_assert(IX_SUBBITS == 13 && IX_NT4 == 512, "the ticket slots hold 3 + 13 bits");
// expected entries of the fullest bucket IF values were held by one row each.  Collections are not like that: a value
// of a cluster is held by ~80 rows at once, so a bucket's fill varies like sqrt(values) x 80, not sqrt(entries) -- C3 at an
// expected 4 464 had a bucket of 7 046.  Half the capacity is headroom.
constexpr double IX_TMAX = 2560.0;
constexpr double IX_TILE_TARGET = 3072.0;       // expected entries of a tile
constexpr uint32_t IX_STAT_SLOTS = 1024;

struct IxStatSlot { unsigned long long inc; uint32_t max_group, groups; };

// ------------------------------------------------------------------------------------------------
// helpers

// exclusive prefix sum over the workgroup (any whole number of waves up to 16); `part`: 16 u32 of LDS; two barriers inside
__device__ __forceinline__ uint32_t ix_block_scan_sum(uint32_t x, uint32_t *part, uint32_t &total)
{
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t incl = x;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
    if (lane == 63u) part[wid] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t k = 0; k < nw; k++) {
        const uint32_t p = part[k];
        if (k < wid) base += p;
        tot += p;
    }
    __syncthreads();
    total = tot;
    return base + incl - x;
}

// exclusive prefix maximum over the workgroup (0 for the first work-item)
__device__ __forceinline__ uint32_t ix_block_scan_max(uint32_t x, uint32_t *part)
{
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint32_t incl = x;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t y = __shfl_up(incl, d);
        if (lane >= d && y > incl) incl = y;
    }
    if (lane == 63u) part[wid] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < wid; k++) base = part[k] > base ? part[k] : base;
    __syncthreads();
    const uint32_t before = __shfl_up(incl, 1u);        // the inclusive maximum of the lane before
    uint32_t ex = lane > 0 ? before : 0u;
    return ex > base ? ex : base;
}

// The tiles in the order they are worked on: groups of `wgrp` windows, inside a group block-major, window-minor -- the
// tiles that follow each other read neighbouring pieces of the same rows (K3) / write them (K5), and the tiles of
// neighbouring blocks, whose pieces of a bucket are neighbours in memory, are a few tiles apart.  The hardware deals
// workgroups to the eight XCDs round-robin: XCD x takes the x-th contiguous eighth of the sequence, so that what follows
// each other meets in ONE L2.
__device__ __forceinline__ bool ix_tile_id(const IxGeom &g, uint32_t &blk, uint32_t &w)
{
    const uint32_t per = gridDim.x >> 3;                  // (the launch has 8 x per workgroups)
    const uint32_t q = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (q >= g.nseq) return false;
    const uint32_t gsz = g.nblk * g.wgrp;
    const uint32_t wg = q / gsz, rem = q - wg * gsz;
    blk = rem / g.wgrp;
    w = wg * g.wgrp + (rem - blk * g.wgrp);
    return w < g.NW;
}

// ------------------------------------------------------------------------------------------------
// K0: lb[row][w] = the first position of the row whose value lies in window w or above (w = 0 .. NW; lb[row][NW] = count).
// One wave per row.
// K0 inside the copy of the table in clustered order (host_compare.cpp: the index is then built on the copy): out[a] = table
// row inv[a], whole rows with their padding, and the window offsets of row a from the same read -- one pass over the table
// less.  cnt: the rows' entry counts in the TABLE's order.  One workgroup per row.
// (the table, its copy and the offsets never overlap: said so, a row's four loads leave before its first store)
__global__ __launch_bounds__(256) void ix_gather_offsets_kernel(IxGeom g, const uint64_t *__restrict__ H, const uint32_t *__restrict__ inv,
                                                                const uint32_t *__restrict__ cnt_table, uint64_t *__restrict__ out,
                                                                uint16_t *__restrict__ lb)
{
    const uint32_t a = blockIdx.x, tid = threadIdx.x;
    const uint32_t r = inv[a];
    const uint32_t cnt = cnt_table[r];
    const uint64_t *src = H + (uint64_t)r * g.stride;
    uint64_t *dst = out + (uint64_t)a * g.stride;
    uint16_t *o = lb + (uint64_t)a * (g.NW + 1u);
    const uint32_t wsh = g.shift + g.bw_log;
    for (uint64_t p = tid; p < g.stride; p += 256u) {
        const uint64_t v = src[p];
        dst[p] = v;
        if (p < cnt) {
            const int w = (int)(uint32_t)(v >> wsh);
            const int wp = p > 0 ? (int)(uint32_t)(src[p - 1] >> wsh) : -1;
            for (int x = wp + 1; x <= w; x++) o[x] = (uint16_t)p;
        }
    }
    const int wl = cnt > 0 ? (int)(uint32_t)(src[cnt - 1] >> wsh) : -1;
    for (uint32_t x = (uint32_t)(wl + 1) + tid; x <= g.NW; x += 256u) o[x] = (uint16_t)cnt;
}

__global__ __launch_bounds__(256) void ix_window_offsets_kernel(IxGeom g, const uint64_t *H, const uint32_t *off, uint16_t *lb)
{
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (row < g.n) {
        const uint32_t cnt = off[row + 1] - off[row];
        const uint64_t *src = H + (uint64_t)row * g.stride;
        uint16_t *out = lb + (uint64_t)row * (g.NW + 1u);
        const uint32_t wsh = g.shift + g.bw_log;
        for (uint32_t p = lane; p < cnt; p += 64u) {
            const int w = (int)(uint32_t)(src[p] >> wsh);
            const int wp = p > 0 ? (int)(uint32_t)(src[p - 1] >> wsh) : -1;
            for (int x = wp + 1; x <= w; x++) out[x] = (uint16_t)p;
        }
        const int wl = cnt > 0 ? (int)(uint32_t)(src[cnt - 1] >> wsh) : -1;
        for (uint32_t x = (uint32_t)(wl + 1) + lane; x <= g.NW; x += 64u) out[x] = (uint16_t)cnt;
    }
}

// ------------------------------------------------------------------------------------------------
// K1: cnt[blk][bucket] = entries of the block's rows in the bucket, for the buckets of the tile's window.
// The loads of the eight rounds of rows are issued before any is used (a round at a time cost a memory latency each).
constexpr uint32_t IX_ROUNDS = IX_RB / (IX_NT / IX_LPR);       // 8 rounds of 64 rows

__global__ __launch_bounds__(IX_NT) void ix_tile_count_kernel(IxGeom g, const uint64_t *__restrict__ H, const uint16_t *__restrict__ lb,
                                                              uint32_t *__restrict__ cnt)
{
    __shared__ uint32_t s_hist[IX_BW_MAX];
    uint32_t blk = 0, w = 0;
    if (!ix_tile_id(g, blk, w)) return;                  // uniform
    const uint32_t tid = threadIdx.x, sub = tid % IX_LPR;
    const uint32_t row0 = blk * IX_RB, nrows = g.n - row0 < IX_RB ? g.n - row0 : IX_RB;
    if (tid < g.BW) s_hist[tid] = 0;
    uint32_t lo[IX_ROUNDS], hi[IX_ROUNDS];
#pragma unroll
    for (uint32_t it = 0; it < IX_ROUNDS; it++) {
        const uint32_t r = it * (IX_NT / IX_LPR) + tid / IX_LPR;
        const uint16_t *p = lb + (uint64_t)(row0 + (r < nrows ? r : 0u)) * (g.NW + 1u) + w;
        const uint32_t a = p[0], b = p[1];
        lo[it] = a;
        hi[it] = r < nrows ? b : a;
    }
    uint64_t v[IX_ROUNDS];
#pragma unroll
    for (uint32_t it = 0; it < IX_ROUNDS; it++) {
        const uint32_t r = it * (IX_NT / IX_LPR) + tid / IX_LPR;
        const bool in = lo[it] + sub < hi[it];
        v[it] = H[in ? (uint64_t)(row0 + r) * g.stride + lo[it] + sub : (uint64_t)row0 * g.stride];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t it = 0; it < IX_ROUNDS; it++) {
        const uint32_t r = it * (IX_NT / IX_LPR) + tid / IX_LPR;
        if (lo[it] + sub < hi[it]) {
            atomicAdd(&s_hist[(uint32_t)(v[it] >> g.shift) & (g.BW - 1u)], 1u);
            const uint64_t *src = H + (uint64_t)(row0 + r) * g.stride;         // (a row with more than IX_LPR entries in the window)
            for (uint32_t k = lo[it] + sub + IX_LPR; k < hi[it]; k += IX_LPR) atomicAdd(&s_hist[(uint32_t)(src[k] >> g.shift) & (g.BW - 1u)], 1u);
        }
    }
    __syncthreads();
    if (tid < g.BW) cnt[(uint64_t)blk * g.Bp + (uint64_t)w * g.BW + tid] = s_hist[tid];
}

// K2a: per bucket the blocks' counts become exclusive prefixes over the blocks; tot[bucket] = the bucket's entries
__global__ __launch_bounds__(256) void ix_col_scan_kernel(IxGeom g, uint32_t *cnt, uint32_t *tot)
{
    // (sixteen blocks' counts are read before the first prefix is written: a load behind every store -- the two may alias, the
    //  compiler keeps their order -- was a round trip per block, 82 us for C3's 196 blocks and 260 beside the fill)
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b < g.Bp) {
        uint32_t run = 0;
        for (uint32_t blk0 = 0; blk0 < g.nblk; blk0 += 16u) {
            uint32_t c[16];
#pragma unroll
            for (uint32_t k = 0; k < 16u; k++) c[k] = blk0 + k < g.nblk ? cnt[(uint64_t)(blk0 + k) * g.Bp + b] : 0u;
#pragma unroll
            for (uint32_t k = 0; k < 16u; k++)
                if (blk0 + k < g.nblk) {
                    cnt[(uint64_t)(blk0 + k) * g.Bp + b] = run;
                    run += c[k];
                }
        }
        tot[b] = run;
    }
}

// K2b (one workgroup): start[b] = exclusive prefix of the buckets' entries (in place), start[Bp] = E; the fullest bucket;
// the buckets that the LDS sort does not take (flags[IXF_NBIG] of them, in no particular order).
// Rounds of 4 096 buckets, four consecutive ones per work-item: the workgroup reads and writes whole lines (round 5 gave every
// work-item a contiguous slice of the buckets -- 380 loads a lane, each lane in a line of its own: 1.24 ms on C5's 390 000
// buckets for 6 MB of traffic).
__global__ __launch_bounds__(1024) void ix_bucket_scan_kernel(IxGeom g, uint32_t *start, uint32_t *flags, uint32_t *biglist)
{
    __shared__ uint32_t s_part[16], s_max[16];
    const uint32_t tid = threadIdx.x;
    uint32_t carry = 0, mx = 0;
    for (uint32_t base = 0; base < g.Bp; base += 4096u) {   // uniform
        const uint32_t b = base + 4u * tid;
        uint32_t c[4], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            c[k] = b + k < g.Bp ? start[b + k] : 0u;
            sum += c[k];
            mx = c[k] > mx ? c[k] : mx;
            // (a bucket beyond the LDS sort's capacity -- a value held by thousands of rows stands in it: ix_big_bucket_kernel's;
            //  the list has room for E / IX_CAP + 1 of them, more cannot exist)
            if (c[k] > IX_CAP) biglist[atomicAdd(&flags[IXF_NBIG], 1u)] = b + k;
        }
        uint32_t total = 0;
        uint32_t run = carry + ix_block_scan_sum(sum, s_part, total);       // (two barriers inside)
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            if (b + k < g.Bp) start[b + k] = run;
            run += c[k];
        }
        carry += total;
    }
    if (tid == 0) start[g.Bp] = carry;
#pragma unroll
    for (uint32_t d = 32; d > 0; d >>= 1) {
        const uint32_t o = __shfl_xor(mx, d);
        mx = o > mx ? o : mx;
    }
    if ((tid & 63u) == 0) s_max[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
        uint32_t m = 0;
        for (uint32_t k = 0; k < 16u; k++) m = s_max[k] > m ? s_max[k] : m;
        flags[IXF_MAXBUCKET] = m;
    }
}

// ------------------------------------------------------------------------------------------------
// The tile's stable sort by bucket: src[0 .. np) holds (local bucket << 12 | index in the piece) in the order the entries were
// read (row by row, positions ascending); LSD passes of 4 bits.  A work-item owns the entries [M t, M t + M) of the current
// order and sixteen private counters, so the order inside a digit is kept without any ranking among lanes.
// cnt: 16 x IX_NT u16 of LDS.  Returns the array that holds the result.  The caller has a barrier in front.
__device__ __forceinline__ uint32_t *ix_tile_sort(uint32_t *src, uint32_t *dst, uint16_t *cnt, uint32_t *part, uint32_t np, uint32_t npass)
{
    const uint32_t tid = threadIdx.x, i0 = tid * IX_M;
    for (uint32_t pass = 0; pass < npass; pass++) {
        const uint32_t sh = 12u + 4u * pass;
#pragma unroll
        for (uint32_t d = 0; d < 16u; d++) cnt[d * IX_NT + tid] = 0;
        uint32_t e[IX_M];
#pragma unroll
        for (uint32_t k = 0; k < IX_M; k++) {
            e[k] = i0 + k < np ? src[i0 + k] : 0xFFFFFFFFu;
            if (i0 + k < np) cnt[((e[k] >> sh) & 15u) * IX_NT + tid]++;
        }
        __syncthreads();
        // exclusive prefix over the 16 x NT counters, digit-major: work-item t owns the counters [16 t, 16 t + 16)
        uint32_t c16[16], sum = 0;
#pragma unroll
        for (uint32_t x = 0; x < 16u; x++) {
            c16[x] = cnt[16u * tid + x];
            sum += c16[x];
        }
        uint32_t total = 0;
        uint32_t run = ix_block_scan_sum(sum, part, total);
#pragma unroll
        for (uint32_t x = 0; x < 16u; x++) {
            cnt[16u * tid + x] = (uint16_t)run;
            run += c16[x];
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < IX_M; k++)
            if (i0 + k < np) {
                uint16_t *c = &cnt[((e[k] >> sh) & 15u) * IX_NT + tid];
                dst[*c] = e[k];
                *c = (uint16_t)(*c + 1u);
            }
        __syncthreads();
        uint32_t *t = src;
        src = dst;
        dst = t;
    }
    return src;
}

// LDS of the tile kernels (bytes): entries as packed words, two arrays of sort items, the sort's counters, the rows'
// segment starts, the buckets' counts and where they go
constexpr uint32_t IXL_PACK = 0, IXL_PA = IXL_PACK + IX_PCAP * 8u, IXL_PB = IXL_PA + IX_PCAP * 4u, IXL_CNT = IXL_PB + IX_PCAP * 4u,
                   IXL_ROWPRE = IXL_CNT + 16u * IX_NT * 2u, IXL_LO = IXL_ROWPRE + (IX_RB + 2u) * 4u, IXL_HIST = IXL_LO + IX_RB * 2u,
                   IXL_GBASE = IXL_HIST + (IX_BW_MAX + 2u) * 4u, IXL_PART = IXL_GBASE + IX_BW_MAX * 4u, IXL_BYTES = IXL_PART + 64u;

// K3: the tile's entries, sorted by bucket, go to pk as one piece per bucket (stable: the block's rows in order, positions
// ascending); where each entry went is left in the position image (K5 reads {code, position} back from there).
__global__ __launch_bounds__(IX_NT, 4) void ix_tile_partition_kernel(IxGeom g, const uint64_t *__restrict__ H, const uint16_t *__restrict__ lb,
                                                                  const uint32_t *__restrict__ colpre, const uint32_t *__restrict__ start,
                                                                  const uint32_t *__restrict__ flags, uint64_t *__restrict__ pk,
                                                                  uint32_t *__restrict__ slot_img)
{
    MG_DYN_SHARED(unsigned char, lds);
    uint64_t *s_pack = reinterpret_cast<uint64_t *>(lds + IXL_PACK);
    uint32_t *s_pa = reinterpret_cast<uint32_t *>(lds + IXL_PA), *s_pb = reinterpret_cast<uint32_t *>(lds + IXL_PB);
    uint16_t *s_cnt = reinterpret_cast<uint16_t *>(lds + IXL_CNT);
    uint32_t *s_rowpre = reinterpret_cast<uint32_t *>(lds + IXL_ROWPRE);
    uint16_t *s_lo = reinterpret_cast<uint16_t *>(lds + IXL_LO);
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(lds + IXL_HIST);
    uint32_t *s_gbase = reinterpret_cast<uint32_t *>(lds + IXL_GBASE);
    uint32_t *s_part = reinterpret_cast<uint32_t *>(lds + IXL_PART);
    uint32_t blk = 0, w = 0;
    if (!ix_tile_id(g, blk, w)) return;                  // uniform
    const uint32_t tid = threadIdx.x, sub = tid % IX_LPR;
    IX_CLK_BEGIN();
    const uint32_t row0 = blk * IX_RB, nrows = g.n - row0 < IX_RB ? g.n - row0 : IX_RB;
    uint32_t len = 0, gb = 0;
    if (tid < g.BW) {                                    // (asked for before the scan waits for the segments' bounds)
        const uint32_t bg = w * g.BW + tid;
        gb = start[bg] + colpre[(uint64_t)blk * g.Bp + bg];
    }
    if (tid < nrows) {
        const uint16_t *p = lb + (uint64_t)(row0 + tid) * (g.NW + 1u) + w;
        const uint32_t lo = p[0];
        len = (uint32_t)p[1] - lo;
        s_lo[tid] = (uint16_t)lo;
    }
    uint32_t total = 0;
    const uint32_t pre = ix_block_scan_sum(len, s_part, total);
    if (total == 0) return;                              // uniform
    s_rowpre[tid] = pre;
    if (tid == IX_NT - 1u) s_rowpre[IX_NT] = total;
    if (tid < g.BW) {
        s_gbase[tid] = gb;
        s_hist[tid] = 0;
    }
    __syncthreads();
    IX_CLK(0);
    const uint64_t lowmask = (1ull << g.shift) - 1ull;    // (shift <= 63)
    for (uint32_t i0 = 0; i0 < total; i0 += IX_PCAP) {
        const uint32_t np = total - i0 < IX_PCAP ? total - i0 : IX_PCAP;
        // the rows' segments, IX_LPR lanes per row: entry `rp + k` of the tile is entry k of the row's segment.  The first
        // entry of every lane in all eight rounds of rows is requested before any is used.
        uint32_t at[IX_ROUNDS];                           // index in the piece of the lane's first entry of the round, ~0: none
        uint64_t v[IX_ROUNDS];
#pragma unroll
        for (uint32_t it = 0; it < IX_ROUNDS; it++) {
            const uint32_t r = it * (IX_NT / IX_LPR) + tid / IX_LPR, rc = r < nrows ? r : 0u;
            const uint32_t rp = s_rowpre[rc], ln = r < nrows ? s_rowpre[rc + 1u] - rp : 0u;
            const uint32_t i = rp + sub;
            const bool in = sub < ln && i >= i0 && i < i0 + np;
            at[it] = in ? i - i0 : 0xFFFFFFFFu;
            v[it] = H[in ? (uint64_t)(row0 + r) * g.stride + s_lo[rc] + sub : (uint64_t)row0 * g.stride];
        }
#pragma unroll
        for (uint32_t it = 0; it < IX_ROUNDS; it++) {
            const uint32_t r = it * (IX_NT / IX_LPR) + tid / IX_LPR;
            if (at[it] != 0xFFFFFFFFu) {
                const uint32_t bl = (uint32_t)(v[it] >> g.shift) & (g.BW - 1u);
                atomicAdd(&s_hist[bl], 1u);
                s_pack[at[it]] = ((v[it] & lowmask) << g.rb) | (uint64_t)(row0 + r);
                s_pa[at[it]] = (bl << 12) | at[it];
            }
        }
        // (what a row holds beyond IX_LPR entries in this window: rare, a plain loop)
#pragma unroll 1
        for (uint32_t r = tid / IX_LPR; r < nrows; r += IX_NT / IX_LPR) {
            const uint32_t rp = s_rowpre[r], ln = s_rowpre[r + 1u] - rp;
            if (ln <= IX_LPR || rp + ln <= i0 || rp >= i0 + np) continue;
            const uint64_t *src = H + (uint64_t)(row0 + r) * g.stride + s_lo[r];
            for (uint32_t k = sub + IX_LPR; k < ln; k += IX_LPR) {
                const uint32_t i = rp + k;
                if (i < i0 || i >= i0 + np) continue;
                const uint64_t x = src[k];
                const uint32_t bl = (uint32_t)(x >> g.shift) & (g.BW - 1u);
                atomicAdd(&s_hist[bl], 1u);
                s_pack[i - i0] = ((x & lowmask) << g.rb) | (uint64_t)(row0 + r);
                s_pa[i - i0] = (bl << 12) | (i - i0);
            }
        }
        __syncthreads();
        IX_CLK(1);
        uint32_t *srt = ix_tile_sort(s_pa, s_pb, s_cnt, s_part, np, g.npass);
        IX_CLK(2);
        uint32_t *s_slot = srt == s_pa ? s_pb : s_pa;     // (the sort's other array: where every entry of the piece goes)
        {   // where every bucket starts in the sorted piece (in place; entry BW = the piece's size)
            const uint32_t c = tid < g.BW ? s_hist[tid] : 0u;
            uint32_t tot = 0;
            const uint32_t ex = ix_block_scan_sum(c, s_part, tot);
            if (tid < g.BW) s_hist[tid] = ex;
            if (tid == 0) s_hist[g.BW] = np;
        }
        __syncthreads();
        IX_CLK(3);
        for (uint32_t j = tid; j < np; j += IX_NT) {
            const uint32_t e = srt[j], b = e >> 12, idx = e & 4095u;
            const uint32_t slot = s_gbase[b] + (j - s_hist[b]);
            pk[slot] = s_pack[idx];
            s_slot[idx] = slot;
        }
        __syncthreads();
        IX_CLK(4);
        {   // where every entry went, into the position image: the lanes' first entries of the eight rounds together ...
            uint32_t sl[IX_ROUNDS], lo8[IX_ROUNDS];
#pragma unroll
            for (uint32_t it = 0; it < IX_ROUNDS; it++) {
                const uint32_t r = it * (IX_NT / IX_LPR) + tid / IX_LPR;
                sl[it] = s_slot[at[it] != 0xFFFFFFFFu ? at[it] : 0u];
                lo8[it] = s_lo[r < nrows ? r : 0u];
            }
#pragma unroll
            for (uint32_t it = 0; it < IX_ROUNDS; it++) {
                const uint32_t r = it * (IX_NT / IX_LPR) + tid / IX_LPR;
                if (at[it] != 0xFFFFFFFFu) slot_img[(uint64_t)(row0 + r) * g.rs + lo8[it] + sub] = sl[it];
            }
        }
        // ... and what a row holds beyond IX_LPR entries in this window
#pragma unroll 1
        for (uint32_t r = tid / IX_LPR; r < nrows; r += IX_NT / IX_LPR) {
            const uint32_t rp = s_rowpre[r], ln = s_rowpre[r + 1u] - rp;
            if (ln <= IX_LPR || rp + ln <= i0 || rp >= i0 + np) continue;
            const uint64_t at0 = (uint64_t)(row0 + r) * g.rs + s_lo[r];
            for (uint32_t k = sub + IX_LPR; k < ln; k += IX_LPR) {
                const uint32_t i = rp + k;
                if (i >= i0 && i < i0 + np) slot_img[at0 + k] = s_slot[i - i0];
            }
        }
        IX_CLK(5);
        // the next piece of the tile goes behind this one in every bucket
        uint32_t add = 0;
        if (tid < g.BW) add = s_hist[tid + 1] - s_hist[tid];
        __syncthreads();
        if (tid < g.BW) {
            s_gbase[tid] += add;
            s_hist[tid] = 0;
        }
        __syncthreads();
    }
}

// K5: {code, position} of every entry, read from where K3 put the entry (its slot stands in the position image) and written
// into the images.  A plain pass over the images, four consecutive positions per work-item (16-byte loads of the slots,
// 16-byte stores of whole lines: the first version wrote a tile's row segments, 23 bytes each, and moved 2 x its payload),
// in pieces of (a block of 512 rows x 32 positions): the reads of tc are gathers, but the rows of a block hold, at the same
// positions, neighbouring values -- neighbouring buckets, whose pieces of this block K3 wrote side by side, and the pieces
// of the next block right behind them.  Pieces follow each other block by block, positions ascending, an XCD taking a
// contiguous eighth of them.
constexpr uint32_t IX5_CH = 32;                           // positions of a piece
constexpr uint32_t IX5_UNR = 4;                           // rows a work-item has in flight

__global__ __launch_bounds__(256) void ix_images_kernel(IxGeom g, const uint32_t *__restrict__ off, const uint32_t *__restrict__ flags,
                                                        const uint2 *__restrict__ tc, uint32_t *__restrict__ code_img, uint32_t *pos_img, uint32_t nchunk)
{
    const uint32_t per = gridDim.x >> 3;
    const uint32_t q = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (q >= g.nblk * nchunk) return;                    // uniform
    const uint32_t blk = q / nchunk, ch = q - blk * nchunk;
    const uint32_t tid = threadIdx.x, quad = tid & 7u, rsub = tid >> 3;
    const uint32_t p0 = ch * IX5_CH + quad * 4u;
    const uint32_t row0 = blk * IX_RB, nrows = g.n - row0 < IX_RB ? g.n - row0 : IX_RB;
    // The slots of the NEXT sweep are requested before this sweep's gathers are waited for: pos_img is read (the slots) and
    // written (the positions) through one pointer, so the compiler keeps a sweep's loads behind the stores of the sweep before --
    // two round trips per sweep, one behind the other (1.0 ms on C3, 1.5 beside the fill).
    uint32_t cnt_n[IX5_UNR];
    uint32_t sl_n[IX5_UNR][4];
    auto slots = [&](uint32_t rb) {
#pragma unroll
        for (uint32_t u = 0; u < IX5_UNR; u++) {
            const uint32_t r = rb + 32u * u;
            const uint32_t row = row0 + (r < nrows ? r : 0u);
            const uint32_t c = r < nrows ? off[row + 1u] - off[row] : 0u;
            cnt_n[u] = c;
            const uint32_t *src = pos_img + (uint64_t)row * g.rs + (p0 < c ? p0 : 0u);      // (rs and p0 are multiples of four: 16-byte aligned)
            sl_n[u][0] = src[0];
            sl_n[u][1] = src[1];
            sl_n[u][2] = src[2];
            sl_n[u][3] = src[3];
        }
    };
    if (rsub < nrows) slots(rsub);
    for (uint32_t rb = rsub; rb < nrows; rb += 32u * IX5_UNR) {          // (32 rows per sweep of the workgroup)
        uint32_t cnt[IX5_UNR];
        uint32_t sl[IX5_UNR][4];
#pragma unroll
        for (uint32_t u = 0; u < IX5_UNR; u++) {
            cnt[u] = cnt_n[u];
#pragma unroll
            for (uint32_t e = 0; e < 4u; e++) sl[u][e] = sl_n[u][e];
        }
        uint2 cp[IX5_UNR][4];
#pragma unroll
        for (uint32_t u = 0; u < IX5_UNR; u++)
#pragma unroll
            for (uint32_t e = 0; e < 4u; e++) cp[u][e] = tc[p0 + e < cnt[u] ? sl[u][e] : 0u];
        if (rb + 32u * IX5_UNR < nrows) slots(rb + 32u * IX5_UNR);
#pragma unroll
        for (uint32_t u = 0; u < IX5_UNR; u++) {
            const uint32_t r = rb + 32u * u;
            if (r < nrows && p0 < g.rs) {
                const uint64_t at = (uint64_t)(row0 + r) * g.rs + p0;
#pragma unroll
                for (uint32_t e = 0; e < 4u; e++) {
                    // (behind the row's last entry: the code image's padding -- larger than every code, so chunked loads past a
                    //  row's end are harmless --, the position image stays whatever it was)
                    code_img[at + e] = p0 + e < cnt[u] ? cp[u][e].x : 0xFFFFFFFFu;
                    if (p0 + e < cnt[u]) pos_img[at + e] = cp[u][e].y;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4: one workgroup per bucket.  Its entries (packed words, in the order K3 wrote them) are sorted by (value, row) -- the
// word's own order: a STABLE counting sort on the 13 bits below the bucket (see ix_bucket_sort_body), then the sub-buckets
// that hold two different values are put in order by comparison.  The groups of equal values are found in LDS; out go the
// values, the rows, the groups' ends, optionally every position's group start, and tc[start + j] = {code, position} of
// the entry that ARRIVED as the bucket's j-th (K5 reads them from there).
// the counting sort's key: the IX_SUBBITS bits of a packed word right below the bucket (all there are if the bucket is narrower)
__device__ __forceinline__ uint32_t ix_subshift(const IxGeom &g) { return g.shift >= IX_SUBBITS ? g.shift + g.rb - IX_SUBBITS : g.rb; }

constexpr uint32_t IX4_PK = 0, IX4_JX = IX4_PK + IX_CAP * 8u, IX4_H = IX4_JX + IX_CAP * 2u, IX4_MIX = IX4_H + IX_NSUB * 2u,
                   IX4_PART = IX4_MIX + IX_NSUB / 8u, IX4_BYTES = IX4_PART + 64u;

// the body for buckets of up to PER x IX_NT4 entries (a bucket of 2 000 entries does not run twelve rounds)
//
// Equal values are the rule, not the exception (a value of a cluster is held by ~80 of its rows), and they must come out
// in row order.  They ARRIVE in row order (K3's partition is stable), so the counting sort is made stable instead of
// ranking equal entries against each other afterwards (80 x 80 comparisons per value: 60 % of this kernel's time in its
// first version): the entries are laid down in LDS in arrival order, wave w owns the sub-buckets whose top three bits are w
// and sweeps ALL entries 64 at a time, taking a ticket for those that are its own -- one wave per counter, its sweeps in
// program order, and lanes that hit one counter in the same instruction served in lane order.  That last point is how
// the LDS behaves, not what the ISA promises: the finished order is therefore CHECKED (strictly ascending words) and a
// bucket that fails it flags the table for the general sort.  What is left to rank by comparison are the sub-buckets that
// hold two different values.
//
// PART: the entries are a PART of a big bucket (ix_big_bucket_kernel): src[j] is the part's j-th word, arr[j] its place in
// the order the BUCKET's entries arrived in (tc is indexed by that, from tcbase), and src / arr are the output arrays'
// own memory (read completely before anything is written).  false: the table was flagged, leave.
// (a whole bucket's entries come from pk and nothing written here is read again: the compiler may know)
template <bool PART> struct IxBodyPtr { using In = const uint64_t *__restrict__; using Keys = uint64_t *__restrict__; using Rows = uint32_t *__restrict__; };
template <> struct IxBodyPtr<true> { using In = const uint64_t *; using Keys = uint64_t *; using Rows = uint32_t *; };

template <uint32_t PER, bool PART>
__device__ __forceinline__ bool ix_bucket_sort_body(const IxGeom &g, typename IxBodyPtr<PART>::In src, const uint32_t *arr, uint32_t tcbase, uint32_t subshift,
                                                    typename IxBodyPtr<PART>::Keys keys_sorted, typename IxBodyPtr<PART>::Rows sorted_rows, uint32_t *__restrict__ gend,
                                                    uint32_t *__restrict__ gs_of, uint2 *__restrict__ tc, IxStatSlot *stat, uint32_t *flags,
                                                    const IxLeaders &lead, uint64_t *s_pk, uint16_t *s_jx, uint32_t *s_h, uint32_t *s_mixed,
                                                    uint32_t *s_part, uint32_t b, uint32_t G0, uint32_t N)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    IX_CLK_BEGIN();
    for (uint32_t x = tid; x < IX_NSUB / 2u; x += IX_NT4) s_h[x] = 0;
    if (tid < IX_NSUB / 32u) s_mixed[tid] = 0;
    // ---- the entries in arrival order (registers and LDS)
    // (the per-entry registers are assigned unconditionally: a conditional element write makes the compiler carry the whole
    //  array through every branch -- 256 VGPRs and spills)
    uint64_t v[PER];
    uint32_t sa[PER];
    uint32_t am[PART ? PER : 1u];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t j = tid + k * IX_NT4;
        const bool in = j < N;
        const uint64_t x = src[in ? j : 0u];
        v[k] = x;
        if constexpr (PART) am[k] = arr[in ? j : 0u];
        const uint32_t sub0 = (uint32_t)(x >> subshift) & (IX_NSUB - 1u);
        if (in) s_jx[j] = (uint16_t)(((sub0 >> IX_SUBLOW) << 13) | (sub0 & ((1u << IX_SUBLOW) - 1u)));
    }
    __syncthreads();
    IX_CLK(32);
    // ---- tickets, in arrival order (two u16 counters per word).  Slot j: bits 15..13 the wave that owns the entry's
    // sub-bucket (its top three bits), below them the sub-bucket's other ten bits until that wave has served the entry, then
    // the ticket (13 bits).  A wave that does not own the entry looks at the top three bits only, whenever it comes by.
    uint16_t *s_tk = s_jx;
    for (uint32_t j0 = 0; j0 < N; j0 += 256u) {           // uniform; four sweeps of 64 in flight (the LDS serves a wave's requests in order)
        uint32_t tag[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
            const uint32_t j = j0 + 64u * u + lane;
            tag[u] = j < N ? (uint32_t)s_tk[j] : 0xFFFFu;    // (beyond the bucket: nobody's -- wave 7 checks j as well)
        }
        uint32_t old[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
            const uint32_t j = j0 + 64u * u + lane;
            const bool mine = (tag[u] >> 13) == wave && j < N;
            const uint32_t sub = (wave << IX_SUBLOW) | (tag[u] & ((1u << IX_SUBLOW) - 1u));
            const uint32_t h16 = (sub & 1u) * 16u;
            uint32_t o = 0;
            IX_IN_LANE_ORDER(if (mine) o = atomicAdd(&s_h[sub >> 1], 1u << h16));
            old[u] = (o >> h16) & 0xFFFFu;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
            const uint32_t j = j0 + 64u * u + lane;
            if ((tag[u] >> 13) == wave && j < N) s_tk[j] = (uint16_t)((wave << 13) | old[u]);
        }
    }
    __syncthreads();
    IX_CLK(33);
    {   // exclusive prefix over the 8192 counters, in place; work-item t owns the words [8 t, 8 t + 8)
        uint32_t wv[8], sum = 0;
#pragma unroll
        for (uint32_t x = 0; x < 8u; x++) {
            wv[x] = s_h[8u * tid + x];
            sum += (wv[x] & 0xFFFFu) + (wv[x] >> 16);
        }
        uint32_t total = 0;
        uint32_t run = ix_block_scan_sum(sum, s_part, total);
#pragma unroll
        for (uint32_t x = 0; x < 8u; x++) {
            const uint32_t lo = wv[x] & 0xFFFFu, hi = wv[x] >> 16;
            s_h[8u * tid + x] = run | ((run + lo) << 16);
            run += lo + hi;
        }
    }
    const uint16_t *cs = reinterpret_cast<const uint16_t *>(s_h);      // cs[sub]: the sub-bucket's first sorted position
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t j = tid + k * IX_NT4;
        sa[k] = (uint32_t)s_tk[j < N ? j : 0u] & 0x1FFFu;               // (the ticket: 13 bits below the owner's three)
    }
    __syncthreads();                                     // (the counters are prefixes, every ticket is in a register: the arrival copy may go)
    IX_CLK(34);
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t j = tid + k * IX_NT4;
        if (j < N) {
            const uint32_t sub = (uint32_t)(v[k] >> subshift) & (IX_NSUB - 1u);
            const uint32_t q = (uint32_t)cs[sub] + sa[k];
            s_pk[q] = v[k];
            s_jx[q] = (uint16_t)j;
        }
    }
    __syncthreads();
    IX_CLK(35);
    // ---- sub-buckets that hold more than one value: marked ...  (all LDS reads first: an atomic between them would order them)
    {
        uint32_t diff = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t q = tid + k * IX_NT4;
            const uint64_t me = s_pk[q < N ? q : 0u];
            const uint32_t sub = (uint32_t)(me >> subshift) & (IX_NSUB - 1u);
            if (q < N && (s_pk[cs[sub]] >> g.rb) != (me >> g.rb)) diff |= 1u << k;
        }
        if (diff) {
#pragma unroll
            for (uint32_t k = 0; k < PER; k++)
                if ((diff >> k) & 1u) {
                    const uint32_t sub = (uint32_t)(s_pk[tid + k * IX_NT4] >> subshift) & (IX_NSUB - 1u);
                    atomicOr(&s_mixed[sub >> 5], 1u << (sub & 31u));
                }
        }
    }
    __syncthreads();
    // ... and put in order: values ascending, the entries of one value as they stand (stable)
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t q = tid + k * IX_NT4;
        const bool in = q < N;
        const uint32_t qc = in ? q : 0u;
        const uint64_t me = s_pk[qc];
        const uint32_t jv = s_jx[qc];
        const uint32_t sub = (uint32_t)(me >> subshift) & (IX_NSUB - 1u);
        uint32_t nq = qc;
        if (in && ((s_mixed[sub >> 5] >> (sub & 31u)) & 1u)) {
            const uint32_t a = cs[sub], e = sub + 1u < IX_NSUB ? (uint32_t)cs[sub + 1u] : N;
            {
                // (the entries of a sub-bucket agree in everything above `subshift`: what is compared are the value's bits below.
                //  No limit on e - a: two values of one clade in a sub-bucket are 2 x 1 000 entries -- a few of a collection's
                //  buckets -- and even a bucket that is ONE sub-bucket costs 6 144 steps per entry, a fraction of a millisecond.)
                uint32_t r = 0;
                if (subshift - g.rb <= 32u) {
                    const uint32_t low = (uint32_t)(me >> g.rb);
                    for (uint32_t x = a; x < e; x++) {
                        const uint32_t lx = (uint32_t)(s_pk[x] >> g.rb);
                        r += (lx < low || (lx == low && x < qc)) ? 1u : 0u;
                    }
                } else {
                    const uint64_t low = me >> g.rb;
                    for (uint32_t x = a; x < e; x++) {
                        const uint64_t lx = s_pk[x] >> g.rb;
                        r += (lx < low || (lx == low && x < qc)) ? 1u : 0u;
                    }
                }
                nq = a + r;
            }
        }
        v[k] = me;
        sa[k] = nq | (jv << 16);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t q = tid + k * IX_NT4;
        if (q < N) {
            s_pk[sa[k] & 0xFFFFu] = v[k];
            s_jx[sa[k] & 0xFFFFu] = (uint16_t)(sa[k] >> 16);
        }
    }
    __syncthreads();
    IX_CLK(36);
    // ---- where the bucket's j-th arrival stands now (the counters are dead: their space holds the inverse), and where the
    // groups of equal values start: one bit per position (a wave's 64 consecutive positions = one ballot = one word).  The
    // same pass checks the stable counting sort: words strictly ascending.
    uint16_t *inv = reinterpret_cast<uint16_t *>(s_h);
    unsigned long long *s_hb = reinterpret_cast<unsigned long long *>(s_mixed);      // (the marks are dead too; N / 64 words <= 1 KB)
    int unordered = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t q = tid + k * IX_NT4;
        const bool in = q < N;
        const uint32_t qc = in ? q : 0u;
        const uint64_t w = s_pk[qc], prev = s_pk[qc > 0 ? qc - 1u : 0u];
        const bool head = in && (q == 0 || (w >> g.rb) != (prev >> g.rb));
        if (in && q > 0 && w <= prev) unordered = 1;
        if (in) inv[s_jx[q]] = (uint16_t)q;
        const unsigned long long bal = __ballot(head);
        if (lane == 0 && q < N) s_hb[q >> 6] = bal;
    }
    unordered = __syncthreads_or(unordered);
    if (unordered) {                                     // uniform
        if (tid == 0) flags[IXF_DEGENERATE] = 1u;
        return false;
    }
    IX_CLK(37);
    // the start of q's group: the highest head bit at or below q
    auto group_start = [&](uint32_t q) -> uint32_t {
        uint32_t w = q >> 6;
        unsigned long long m = s_hb[w] & (~0ull >> (63u - (q & 63u)));
        while (m == 0) m = s_hb[--w];                      // (position 0 is a head)
        return (w << 6) + 63u - (uint32_t)__builtin_clzll(m);
    };
    auto is_head = [&](uint32_t q) -> bool { return (s_hb[q >> 6] >> (q & 63u)) & 1ull; };      // q < N
    const uint64_t rowmask = (1ull << g.rb) - 1ull;
    const uint64_t vbase = (uint64_t)b << g.shift;
    // ---- the statistics; the dense groups' leaders (see IxLeaders); out, by sorted position.  First everything that is
    // READ (LDS, and the rows' groups from global memory: all requests in flight together), then what is written -- a
    // gather between two stores waits for its own round trip, PER times in a row (27 % of the kernel before).
    unsigned long long inc = 0;
    uint32_t heads = 0, glen = 0, lmask = 0;
    uint32_t gsr[PER], grpr[PER], g0r[PER], g1r[PER];
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t q = tid + k * IX_NT4;
        const bool in = q < N;
        const uint32_t qc = in ? q : 0u;
        const uint32_t gs = group_start(qc);
        const bool last = qc + 1u == N || is_head(qc + 1u);
        gsr[k] = gs;
        // (a value held by one row has no leader: its row's group is not looked up)
        const bool look = lead.grp_of != nullptr && in && !(last && qc == gs);
        const uint32_t *gp = lead.grp_of + 4ull * (uint32_t)(s_pk[qc] & rowmask);
        grpr[k] = look ? gp[0] : 0xFFFFFFFFu;
        g0r[k] = look ? gp[1] : 0u;
        g1r[k] = look ? gp[2] : 0u;
        inc += in ? q - gs : 0u;
        heads += (in && q == gs) ? 1u : 0u;
        const uint32_t len = (in && last) ? q + 1u - gs : 0u;
        glen = len > glen ? len : glen;
    }
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t q = tid + k * IX_NT4;
        if (q < N) {
            const uint64_t w = s_pk[q];
            const uint32_t gs = gsr[k];
            const bool last = q + 1u == N || is_head(q + 1u);
            keys_sorted[G0 + q] = vbase | (w >> g.rb);
            sorted_rows[G0 + q] = (uint32_t)(w & rowmask);
            if (gs_of) gs_of[G0 + q] = G0 + gs;
            if (last) gend[G0 + gs] = G0 + q + 1u;
            if (grpr[k] != 0xFFFFFFFFu) {
                const bool first = q == gs || (uint32_t)(s_pk[q - 1u] & rowmask) < g0r[k];
                const bool more = !last && (uint32_t)(s_pk[q + 1u] & rowmask) < g1r[k];
                if (first && more) lmask |= 1u << k;
            }
        }
    }
    IX_CLK(38);
    // statistics of the index: one atomic per workgroup and number, spread over the slots
#pragma unroll
    for (uint32_t d = 32; d > 0; d >>= 1) {
        inc += __shfl_xor(inc, d);
        heads += __shfl_xor(heads, d);
        const uint32_t o = __shfl_xor(glen, d);
        glen = o > glen ? o : glen;
    }
    __shared__ unsigned long long s_inc[IX_NT4 / 64];
    __shared__ uint32_t s_len[IX_NT4 / 64], s_heads[IX_NT4 / 64];
    if (lane == 0) {
        s_inc[wave] = inc;
        s_len[wave] = glen;
        s_heads[wave] = heads;
    }
    uint32_t ltotal = 0, lbase = 0;
    if (lead.grp_of) lbase = ix_block_scan_sum((uint32_t)__popc(lmask), s_part, ltotal);       // uniform (barriers inside)
    else __syncthreads();
    if (tid == 0) {
        // (the leaders' place in their list first: its round trip to the L2 runs while everybody writes tc below --
        //  the whole workgroup waiting for it was 19 % of the kernel)
        if (ltotal) s_part[15] = atomicAdd(&lead.cnt[b & (lead.nsub - 1u)], ltotal);
        unsigned long long t = 0;
        uint32_t m = 0, h = 0;
        for (uint32_t k = 0; k < IX_NT4 / 64u; k++) {
            t += s_inc[k];
            m = s_len[k] > m ? s_len[k] : m;
            h += s_heads[k];
        }
        IxStatSlot *sl = stat + (b & (IX_STAT_SLOTS - 1u));
        if (t) atomicAdd(&sl->inc, t);
        if (m > 1u) atomicMax(&sl->max_group, m);
        atomicAdd(&sl->groups, h);
    }
    IX_CLK(39);
    // ---- by arrival: what K5 carries into the images
#pragma unroll
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t j = tid + k * IX_NT4;
        if (j < N) {
            const uint32_t q = inv[j], gs = group_start(q);
            const bool last = q + 1u == N || is_head(q + 1u);
            const uint32_t shared = (gs == q && last) ? 0u : 1u;
            uint32_t at = j;
            if constexpr (PART) at = am[k];
            tc[tcbase + at] = make_uint2(((G0 + gs) << 1) | shared, G0 + q);
        }
    }
    IX_CLK(40);
    if (ltotal) {                                        // uniform
        __syncthreads();
        const uint32_t sub = b & (lead.nsub - 1u);
        uint32_t at = s_part[15] + lbase;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            if ((lmask >> k) & 1u) {
                const uint32_t q = tid + k * IX_NT4;
                if (at < lead.cap_sub) {
                    const uint64_t slot = (uint64_t)sub * lead.cap_sub + at;
                    lead.key[slot] = ((unsigned long long)grpr[k] << 32) | (unsigned long long)(G0 + gsr[k]);
                    lead.val[slot] = G0 + q;
                }
                at++;
            }
        }
    }
    IX_CLK(41);
    return true;
}

__global__ __launch_bounds__(IX_NT4, 4) void ix_bucket_sort_kernel(IxGeom g, const uint64_t *pk, const uint32_t *start, uint64_t *keys_sorted,
                                                                uint32_t *sorted_rows, uint32_t *gend, uint32_t *gs_of, uint2 *tc,
                                                                IxStatSlot *stat, uint32_t *flags, IxLeaders lead)
{
    MG_DYN_SHARED(unsigned char, lds);
    uint64_t *s_pk = reinterpret_cast<uint64_t *>(lds + IX4_PK);
    uint16_t *s_jx = reinterpret_cast<uint16_t *>(lds + IX4_JX);
    uint32_t *s_h = reinterpret_cast<uint32_t *>(lds + IX4_H);
    uint32_t *s_mixed = reinterpret_cast<uint32_t *>(lds + IX4_MIX);
    uint32_t *s_part = reinterpret_cast<uint32_t *>(lds + IX4_PART);
    const uint32_t b = blockIdx.x;
    const uint32_t G0 = start[b], N = start[b + 1] - G0;
    if (N == 0 || N > IX_CAP) return;                    // uniform  (beyond the capacity: ix_big_bucket_kernel's)
    const uint32_t per = (N + IX_NT4 - 1u) / IX_NT4;      // uniform
    const uint32_t subshift = ix_subshift(g);
#define IX_BODY(P)                                                                                                                          \
    (void)ix_bucket_sort_body<P, false>(g, pk + G0, nullptr, G0, subshift, keys_sorted, sorted_rows, gend, gs_of, tc, stat, flags, lead, s_pk, s_jx, \
                                        s_h, s_mixed, s_part, b, G0, N)
    if (per <= 3u) IX_BODY(3);
    else if (per <= 4u) IX_BODY(4);
    else if (per <= 5u) IX_BODY(5);
    else if (per <= 6u) IX_BODY(6);
    else if (per <= 8u) IX_BODY(8);
    else IX_BODY(12);
#undef IX_BODY
}

// ------------------------------------------------------------------------------------------------
// K4b: the buckets beyond the LDS sort's capacity.  Sketches of one species share most of their values: a value held by
// 16 000 rows makes a bucket of 16 000 + the usual entries.  One workgroup takes such a bucket:
//   1. the same STABLE counting sort on the 13 bits below the bucket, but through global memory (ix_big_split: a histogram,
//      its prefix, then the entries in arrival order -- one wave per counter range as in the body, the counter's old value
//      IS the entry's place) into the bucket's own range of the OUTPUT arrays: keys_sorted / sorted_rows hold {word, arrival
//      index} until their final content is written over it, part by part;
//   2. the sorted sub-buckets are walked (ix_big_walk): a run of them that fits the LDS goes through ix_bucket_sort_body (a
//      sub-bucket alone: on the NEXT 13 bits).  A sub-bucket that does not fit holds a value with thousands of holders and,
//      one time in four, a few strangers: it is split once more on the next 13 bits (into the bucket's range of pk / gend,
//      which nobody needs any more) and walked; what does not fit THEN is one value, already in row order, written out as it
//      stands (ix_big_value) -- or two values that 26 bits do not tell apart: flagged, the general sort's.
constexpr uint32_t IXB_CS1 = IX4_BYTES, IXB_CS2 = IXB_CS1 + IX_NSUB * 4u, IXB_BYTES = IXB_CS2 + IX_NSUB * 4u;
constexpr uint32_t IXB_CH = 4096;               // entries staged per round of the ordered pass: words, arrival indices, tags
constexpr uint32_t IXB_W = 0, IXB_ARR = IXB_W + IXB_CH * 8u, IXB_TAG = IXB_ARR + IXB_CH * 4u;
static_assert(IXB_TAG + IXB_CH * 2u <= IX4_H, "the staging area lies in the body's entry arrays");

struct IxOut {
    uint64_t *keys_sorted;
    uint32_t *sorted_rows, *gend, *gs_of;
    uint2 *tc;
    IxStatSlot *stat;
    uint32_t *flags;
};

// the counters after ix_big_split: cs[k] = one past sub-bucket k's last place
__device__ __forceinline__ uint32_t ix_big_first(const uint32_t *cs, uint32_t k) { return k ? cs[k - 1u] : 0u; }

// Stable counting sort of `count` words by the IX_SUBBITS bits at `sshift`: src_w[j] (and src_arr[j], or j itself) goes to
// dst_w / dst_arr [its place].  cs: IX_NSUB counters of LDS; lds: the staging area.
__device__ __forceinline__ void ix_big_split(const uint64_t *src_w, const uint32_t *src_arr, uint64_t *dst_w, uint32_t *dst_arr, uint32_t count,
                                             uint32_t sshift, uint32_t *cs, unsigned char *lds, uint32_t *s_part)
{
    uint64_t *s_w = reinterpret_cast<uint64_t *>(lds + IXB_W);
    uint32_t *s_arr = reinterpret_cast<uint32_t *>(lds + IXB_ARR);
    uint16_t *s_tag = reinterpret_cast<uint16_t *>(lds + IXB_TAG);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t x = tid; x < IX_NSUB; x += IX_NT4) cs[x] = 0;
    __syncthreads();
    for (uint32_t j = tid; j < count; j += IX_NT4) atomicAdd(&cs[(uint32_t)(src_w[j] >> sshift) & (IX_NSUB - 1u)], 1u);
    __syncthreads();
    {
        constexpr uint32_t W = IX_NSUB / IX_NT4;
        uint32_t c[W], sum = 0;
#pragma unroll
        for (uint32_t x = 0; x < W; x++) {
            c[x] = cs[W * tid + x];
            sum += c[x];
        }
        uint32_t total = 0;
        uint32_t run = ix_block_scan_sum(sum, s_part, total);
#pragma unroll
        for (uint32_t x = 0; x < W; x++) {
            cs[W * tid + x] = run;
            run += c[x];
        }
    }
    for (uint32_t j0 = 0; j0 < count; j0 += IXB_CH) {    // uniform
        __syncthreads();
        for (uint32_t x = tid; x < IXB_CH; x += IX_NT4) {
            const uint32_t j = j0 + x, jc = j < count ? j : 0u;
            const uint64_t w = src_w[jc];
            const uint32_t sub0 = (uint32_t)(w >> sshift) & (IX_NSUB - 1u);
            s_w[x] = w;
            s_arr[x] = src_arr ? src_arr[jc] : j;
            s_tag[x] = (uint16_t)(((sub0 >> IX_SUBLOW) << 13) | (sub0 & ((1u << IX_SUBLOW) - 1u)));
        }
        __syncthreads();
        const uint32_t xe = count - j0 < IXB_CH ? count - j0 : IXB_CH;
        for (uint32_t x0 = 0; x0 < xe; x0 += 256u) {      // uniform
            uint32_t tag[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) tag[u] = s_tag[x0 + 64u * u + lane];
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) {
                const uint32_t x = x0 + 64u * u + lane;
                const bool mine = (tag[u] >> 13) == wave && x < xe;
                const uint32_t sub = (wave << IX_SUBLOW) | (tag[u] & ((1u << IX_SUBLOW) - 1u));
                uint32_t at = 0;
                IX_IN_LANE_ORDER(if (mine) at = atomicAdd(&cs[sub], 1u));
                if (mine) {
                    dst_w[at] = s_w[x];
                    dst_arr[at] = s_arr[x];
                }
            }
        }
    }
    __threadfence();
    __syncthreads();
}

// do the words src_w[0 .. m) hold ONE value?  (uniform)
__device__ __forceinline__ bool ix_big_one_value(const IxGeom &g, const uint64_t *src_w, uint32_t m)
{
    const uint64_t low0 = src_w[0] >> g.rb;
    int bad = 0;
    for (uint32_t q = threadIdx.x; q < m; q += IX_NT4) bad |= (src_w[q] >> g.rb) != low0 ? 1 : 0;
    return __syncthreads_or(bad) == 0;
}

// One value (ix_big_one_value), m > IX_CAP holders: its words src_w[0 .. m) in row order, src_arr their arrival indices; gs:
// its first sorted position.  (src may be the output arrays' own memory: a round reads all its entries before it writes any.)
__device__ __forceinline__ bool ix_big_value(const IxGeom &g, const uint64_t *src_w, const uint32_t *src_arr, const IxOut &o, const IxLeaders &lead,
                                             uint32_t *s_row, uint32_t *s_part, uint32_t b, uint32_t tcbase, uint32_t gs, uint32_t m)
{
    const uint32_t tid = threadIdx.x;
    const uint64_t rowmask = (1ull << g.rb) - 1ull;
    const uint64_t vbase = (uint64_t)b << g.shift;
    int bad = 0;
    for (uint32_t q0 = 0; q0 < m; q0 += IX_NT4) {         // uniform
        const uint32_t q = q0 + tid;
        const bool in = q < m, more_after = q + 1u < m;
        const uint64_t w = src_w[in ? q : 0u];
        const uint32_t ar = src_arr[in ? q : 0u];
        const uint32_t next_row = (uint32_t)(src_w[more_after ? q + 1u : 0u] & rowmask);
        const uint32_t row = (uint32_t)(w & rowmask);
        s_row[tid + 1u] = row;
        __syncthreads();                                 // (everything of this round is read: its places may be written)
        const uint32_t prev_row = s_row[tid];            // (work-item 0: the round before's last row)
        if (in && q > 0 && prev_row >= row) bad = 1;      // the stable order that the splits are trusted to give: checked
        bool isl = false;
        uint32_t grp = 0xFFFFFFFFu;
        if (lead.grp_of != nullptr && in) {
            const uint32_t *gp = lead.grp_of + 4ull * row;
            grp = gp[0];
            const bool first = q == 0 || prev_row < gp[1];
            isl = grp != 0xFFFFFFFFu && first && more_after && next_row < gp[2];
        }
        __syncthreads();
        if (tid == IX_NT4 - 1u) s_row[0] = row;
        if (in) {
            o.keys_sorted[gs + q] = vbase | (w >> g.rb);
            o.sorted_rows[gs + q] = row;
            if (o.gs_of) o.gs_of[gs + q] = gs;
            o.tc[tcbase + ar] = make_uint2((gs << 1) | 1u, gs + q);
        }
        if (lead.grp_of != nullptr) {                    // uniform
            uint32_t ltotal = 0;
            const uint32_t lbase = ix_block_scan_sum(isl ? 1u : 0u, s_part, ltotal);
            if (ltotal) {                                // uniform
                const uint32_t sub = b & (lead.nsub - 1u);
                if (tid == 0) s_part[15] = atomicAdd(&lead.cnt[sub], ltotal);
                __syncthreads();
                const uint32_t at = s_part[15] + lbase;
                if (isl && at < lead.cap_sub) {
                    const uint64_t slot = (uint64_t)sub * lead.cap_sub + at;
                    lead.key[slot] = ((unsigned long long)grp << 32) | (unsigned long long)gs;
                    lead.val[slot] = gs + q;
                }
                __syncthreads();
            }
        }
    }
    bad = __syncthreads_or(bad);
    if (bad) {                                           // uniform
        if (tid == 0) o.flags[IXF_DEGENERATE] = 1u;
        return false;
    }
    if (tid == 0) {
        o.gend[gs] = gs + m;
        atomicAdd(&o.flags[IXF_NSTREAMED], 1u);
        IxStatSlot *sl = o.stat + (b & (IX_STAT_SLOTS - 1u));
        atomicAdd(&sl->inc, (unsigned long long)m * (unsigned long long)(m - 1u) / 2ull);
        atomicMax(&sl->max_group, m);
        atomicAdd(&sl->groups, 1u);
    }
    return true;
}

// The sub-buckets of a split (cs, made on the bits at sshift) in order: src_w / src_arr [0 .. count) are the split's result,
// gpos the sorted position of its first entry; alt_w / alt_arr: free memory of the same extent for one more split (LEVEL 1).
template <int LEVEL>
__device__ bool ix_big_walk(const IxGeom &g, const uint64_t *src_w, const uint32_t *src_arr, uint64_t *alt_w, uint32_t *alt_arr, uint32_t gpos,
                            uint32_t count, uint32_t sshift, const uint32_t *cs, uint32_t *cs_next, const IxOut &o, const IxLeaders &lead,
                            unsigned char *lds, uint32_t b, uint32_t tcbase)
{
    uint64_t *s_pk = reinterpret_cast<uint64_t *>(lds + IX4_PK);
    uint16_t *s_jx = reinterpret_cast<uint16_t *>(lds + IX4_JX);
    uint32_t *s_h = reinterpret_cast<uint32_t *>(lds + IX4_H);
    uint32_t *s_mixed = reinterpret_cast<uint32_t *>(lds + IX4_MIX);
    uint32_t *s_part = reinterpret_cast<uint32_t *>(lds + IX4_PART);
    const uint32_t sshift_next = sshift - g.rb >= IX_SUBBITS ? sshift - IX_SUBBITS : g.rb;
    uint32_t sub = 0;
    while (sub < IX_NSUB) {                              // uniform
        const uint32_t a = ix_big_first(cs, sub);
        if (a == count) break;
        uint32_t lo = sub, hi = IX_NSUB;                 // first(lo) - a <= IX_CAP < first(hi) - a, unless everything left fits
        if (count - a <= IX_CAP) lo = hi;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (ix_big_first(cs, mid) - a <= IX_CAP) lo = mid;
            else hi = mid;
        }
        bool ok = true;
        if (lo == sub) {
            const uint32_t m = cs[sub] - a;
            if (ix_big_one_value(g, src_w + a, m)) {     // uniform
                ok = ix_big_value(g, src_w + a, src_arr + a, o, lead, reinterpret_cast<uint32_t *>(lds + IX4_PK), s_part, b, tcbase, gpos + a, m);
            } else if (LEVEL == 1 && sshift > g.rb) {    // uniform: strangers beside the value -- the next 13 bits part them
                if constexpr (LEVEL == 1) {
                    ix_big_split(src_w + a, src_arr + a, alt_w + a, alt_arr + a, m, sshift_next, cs_next, lds, s_part);
                    ok = ix_big_walk<2>(g, alt_w + a, alt_arr + a, nullptr, nullptr, gpos + a, m, sshift_next, cs_next, nullptr, o, lead, lds, b, tcbase);
                }
            } else {                                     // two values that 26 bits below the bucket do not tell apart
                if (threadIdx.x == 0) o.flags[IXF_DEGENERATE] = 1u;
                ok = false;
            }
            sub++;
        } else {
            const uint32_t cnt = ix_big_first(cs, lo) - a;
            if (cnt)
                ok = ix_bucket_sort_body<IX_CAP / IX_NT4, true>(g, src_w + a, src_arr + a, tcbase, lo == sub + 1u ? sshift_next : sshift, o.keys_sorted,
                                                                o.sorted_rows, o.gend, o.gs_of, o.tc, o.stat, o.flags, lead, s_pk, s_jx, s_h, s_mixed, s_part,
                                                                b, gpos + a, cnt);
            sub = lo;
        }
        if (!ok) return false;                           // uniform (the table is flagged)
        __syncthreads();
    }
    return true;
}

__global__ __launch_bounds__(IX_NT4, 2) void ix_big_bucket_kernel(IxGeom g, uint64_t *pk, const uint32_t *start, const uint32_t *biglist,
                                                               uint64_t *keys_sorted, uint32_t *sorted_rows, uint32_t *gend, uint32_t *gs_of, uint2 *tc,
                                                               IxStatSlot *stat, uint32_t *flags, IxLeaders lead)
{
    MG_DYN_SHARED(unsigned char, lds);
    uint32_t *s_part = reinterpret_cast<uint32_t *>(lds + IX4_PART);
    uint32_t *cs1 = reinterpret_cast<uint32_t *>(lds + IXB_CS1), *cs2 = reinterpret_cast<uint32_t *>(lds + IXB_CS2);
    const IxOut o{keys_sorted, sorted_rows, gend, gs_of, tc, stat, flags};
    const uint32_t nbig = flags[IXF_NBIG];
    const uint32_t sshift = ix_subshift(g);
    for (uint32_t i = blockIdx.x; i < nbig; i += gridDim.x) {            // uniform
        const uint32_t b = biglist[i];
        const uint32_t G0 = start[b], N = start[b + 1] - G0;
        ix_big_split(pk + G0, nullptr, keys_sorted + G0, sorted_rows + G0, N, sshift, cs1, lds, s_part);
        if (!ix_big_walk<1>(g, keys_sorted + G0, sorted_rows + G0, pk + G0, gend + G0, G0, N, sshift, cs1, cs2, o, lead, lds, b, G0)) return;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void ix_stat_reduce_kernel(const IxStatSlot *stat, unsigned long long *incidences, uint32_t *max_group, uint32_t *groups)
{
    unsigned long long inc = 0;
    uint32_t m = 0, gr = 0;
    for (uint32_t i = threadIdx.x; i < IX_STAT_SLOTS; i += 256u) {
        inc += stat[i].inc;
        m = stat[i].max_group > m ? stat[i].max_group : m;
        gr += stat[i].groups;
    }
    if (inc) atomicAdd(incidences, inc);
    if (m) atomicMax(max_group, m);
    if (gr) atomicAdd(groups, gr);
}

__global__ __launch_bounds__(256) void ix_fill_u32_kernel(uint32_t *p, uint64_t count, uint32_t v)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < count; i += stride) p[i] = v;
}

// Verification (MASHGPU_SPARSE_INDEX=verify): two builds of one index, word by word.  mode 0: every word; 1: where cond[i] == i
// (group starts: the groups' ends are defined there only); 2: where cond[i] != 0xFFFFFFFF (entries, not the images' padding).
// out[0] = words that differ, out[1] = the first of them.
__global__ __launch_bounds__(256) void ix_verify_kernel(const uint32_t *a, const uint32_t *b, const uint32_t *cond, uint32_t mode, uint64_t count,
                                                        unsigned long long *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < count; i += stride) {
        if (mode == 1u && cond[i] != (uint32_t)i) continue;
        if (mode == 2u && cond[i] == 0xFFFFFFFFu) continue;
        if (a[i] != b[i]) {
            atomicAdd(&out[0], 1ull);
            atomicMin(&out[1], (unsigned long long)i);
        }
    }
}

hipError_t index_verify_words(const uint32_t *a, const uint32_t *b, const uint32_t *cond, uint32_t mode, uint64_t count, unsigned long long *out2,
                              hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    uint64_t blocks = (count + 1023u) / 1024u;
    if (blocks > 8192u) blocks = 8192u;
    hipLaunchKernelGGL(ix_verify_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, a, b, cond, mode, count, out2);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// host side

size_t index_stat_scratch_bytes() { return sizeof(IxStatSlot) * IX_STAT_SLOTS; }

#if defined(IX_PHASE_CLOCKS) && !defined(MG_HIP_EMU)
void index_dump_clocks()
{
    unsigned long long h[64];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(ix_clk), sizeof h) != hipSuccess) return;
    static const char *k3[] = {"prologue (lb, scan, gbase)", "loads + pack", "tile sort", "bucket scan", "pk + slots to LDS", "slot image"};
    static const char *k4[] = {"zero + load", "tickets", "counter scan + ticket read", "scatter", "mixed mark + fix-up + write back", "heads + inverse", "out by position + leaders", "statistics + leaders' scan", "tc by arrival", "leaders out"};
    unsigned long long t3 = 0, t4 = 0;
    for (int i = 0; i < 6; i++) t3 += h[i];
    for (int i = 0; i < 10; i++) t4 += h[32 + i];
    for (int i = 0; i < 6; i++) fprintf(stderr, "ix clocks K3 %-34s %6.2f %%\n", k3[i], 100.0 * h[i] / (t3 ? t3 : 1));
    for (int i = 0; i < 10; i++) fprintf(stderr, "ix clocks K4 %-34s %6.2f %%\n", k4[i], 100.0 * h[32 + i] / (t4 ? t4 : 1));
    memset(h, 0, sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ix_clk), h, sizeof h);
}
#endif

IxPlan index_plan(uint32_t n, uint32_t E, uint32_t s, uint32_t rs, uint64_t stride, uint64_t maxv, double dens0, bool want_gs)
{
    IxPlan p;
    IxGeom &g = p.g;
    g = IxGeom{};
    if (n == 0 || E == 0) { p.why = "empty"; return p; }
    if (s > 65535u) { p.why = "sketch size beyond 16-bit positions"; return p; }
    g.n = n;
    g.E = E;
    g.rs = rs;
    g.stride = stride;
    g.want_gs = want_gs ? 1u : 0u;
    g.nblk = (n + IX_RB - 1u) / IX_RB;
    g.rb = 1;
    while (g.rb < 32u && (1ull << g.rb) < (uint64_t)n) g.rb++;
    // the widest bucket whose expected fill stays below IX_TMAX where the table is densest
    if (!(dens0 > 0.0)) { p.why = "no density"; return p; }
    uint32_t shift = 0;
    while (shift < 63u && dens0 * (double)(1ull << (shift + 1u)) <= IX_TMAX) shift++;
    if (shift > 64u - g.rb) shift = 64u - g.rb;           // value bits below the bucket | row: one 64-bit word
    if (shift > 63u) shift = 63u;
    g.shift = shift;
    const uint64_t B = (maxv >> shift) + 1ull;
    if (B > (1ull << 22)) { p.why = "too many buckets (values far from evenly spread)"; return p; }
    // buckets per window: the largest power of two that keeps a tile near IX_TILE_TARGET entries
    uint32_t bw_log = 0;
    while (bw_log < 9u && (1ull << (bw_log + 1u)) <= IX_BW_MAX && shift + bw_log + 1u <= 63u && (1ull << bw_log) < B &&
           (double)E * (double)(2u << bw_log) / ((double)g.nblk * (double)B) <= IX_TILE_TARGET)
        bw_log++;
    g.bw_log = bw_log;
    g.BW = 1u << bw_log;
    g.NW = (uint32_t)((B + g.BW - 1u) / g.BW);
    g.Bp = g.NW * g.BW;
    g.npass = (bw_log + 3u) / 4u;
    g.wgrp = 8;
    const uint64_t nseq = (uint64_t)((g.NW + g.wgrp - 1u) / g.wgrp) * g.wgrp * g.nblk;
    if (nseq >= (1ull << 30)) { p.why = "too many tiles"; return p; }
    g.nseq = (uint32_t)nseq;
    p.lb_bytes = (size_t)n * (g.NW + 1u) * 2u;
    p.cnt_bytes = (size_t)g.nblk * g.Bp * 4u;
    p.start_bytes = ((size_t)g.Bp + 1u) * 4u;
    p.pk_bytes = (size_t)E * 8u;
    p.tc_bytes = (size_t)E * 8u;
    p.big_bytes = ((size_t)E / IX_CAP + 2u) * 4u;
    if (p.lb_bytes > ((size_t)2 << 30) || p.cnt_bytes > ((size_t)2 << 30)) { p.why = "scratch of the tiles too large"; return p; }
    p.ok = true;
    return p;
}

hipError_t index_gather_rows(const IxPlan &plan, const uint64_t *hashes, const uint32_t *inv, const uint32_t *cnt_table, uint64_t *out, void *lb_v,
                             hipStream_t stream)
{
    if (!plan.ok) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ix_gather_offsets_kernel, dim3(plan.g.n), dim3(256), 0, stream, plan.g, hashes, inv, cnt_table, out, static_cast<uint16_t *>(lb_v));
    return hipGetLastError();
}

// TEST KNOB (MASHGPU_IX_DEBUG_SWAP, host_index.cpp): between the partition and the bucket sorts, swap the first two
// neighbouring entries of one bucket that hold the SAME value -- the stable counting sort then leaves that value's rows in
// descending order, exactly what a ticket served out of lane order would do, and the order check of ix_bucket_sort_body has to
// catch it (flags[IXF_DEGENERATE]; the host builds the index by the general sort).  One wave; done[0] = 1 if a pair was found.
__global__ __launch_bounds__(64) void ix_debug_swap_kernel(IxGeom g, uint64_t *pk, const uint32_t *start, uint32_t *done)
{
    if (threadIdx.x != 0) return;
    for (uint32_t b = 0; b < g.Bp; b++) {
        const uint32_t a = start[b], e = start[b + 1];
        if (e - a < 2u || e - a > IX_CAP) continue;
        for (uint32_t j = a; j + 1u < e; j++) {
            if ((pk[j] >> g.rb) == (pk[j + 1] >> g.rb)) {
                const uint64_t t = pk[j];
                pk[j] = pk[j + 1];
                pk[j + 1] = t;
                done[0] = 1u;
                return;
            }
        }
    }
}

hipError_t index_debug_swap(const IxPlan &plan, void *pk_v, const void *start_v, uint32_t *done, hipStream_t stream)
{
    if (!plan.ok) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ix_debug_swap_kernel, dim3(1), dim3(64), 0, stream, plan.g, static_cast<uint64_t *>(pk_v), static_cast<const uint32_t *>(start_v), done);
    return hipGetLastError();
}

hipError_t index_build(const IxPlan &plan, const uint64_t *hashes, const uint32_t *off, void *lb_v, void *cnt_v, void *start_v, void *big_v, void *pk_v, void *tc_v,
                       uint64_t *keys_sorted, uint32_t *sorted_rows, uint32_t *gend, uint32_t *gs_of, uint32_t *code_img, uint32_t *pos_img,
                       void *stat_scratch, unsigned long long *incidences, uint32_t *max_group, uint32_t *groups, uint32_t *flags,
                       const IxLeaders *leaders, hipStream_t stream, int stages)
{
    // (leaders: looked at by stage 2 only)
    if (!plan.ok) return hipErrorInvalidValue;
    const IxGeom g = plan.g;
    uint16_t *lb = static_cast<uint16_t *>(lb_v);
    uint32_t *cnt = static_cast<uint32_t *>(cnt_v), *start = static_cast<uint32_t *>(start_v);
    uint64_t *pk = static_cast<uint64_t *>(pk_v);
    uint2 *tc = static_cast<uint2 *>(tc_v);
    hipError_t e = hipSuccess;
    if (stages & 1) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(ix_tile_partition_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)IXL_BYTES);
        if (e != hipSuccess) return e;
        const uint32_t tiles = 8u * ((g.nseq + 7u) / 8u);
        if (!(stages & 8)) hipLaunchKernelGGL(ix_window_offsets_kernel, dim3((g.n + 3u) / 4u), dim3(256), 0, stream, g, hashes, off, lb);
        hipLaunchKernelGGL(ix_tile_count_kernel, dim3(tiles), dim3(IX_NT), 0, stream, g, hashes, (const uint16_t *)lb, cnt);
        hipLaunchKernelGGL(ix_col_scan_kernel, dim3((g.Bp + 255u) / 256u), dim3(256), 0, stream, g, cnt, start);
        hipLaunchKernelGGL(ix_bucket_scan_kernel, dim3(1), dim3(1024), 0, stream, g, start, flags, static_cast<uint32_t *>(big_v));
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(ix_tile_partition_kernel, dim3(tiles), dim3(IX_NT), IXL_BYTES, stream, g, hashes, (const uint16_t *)lb, (const uint32_t *)cnt,
                           (const uint32_t *)start, (const uint32_t *)flags, pk, pos_img);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (stages & 2) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(ix_bucket_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)IX4_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(ix_big_bucket_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)IXB_BYTES);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(stat_scratch, 0, index_stat_scratch_bytes(), stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(ix_bucket_sort_kernel, dim3(g.Bp), dim3(IX_NT4), IX4_BYTES, stream, g, (const uint64_t *)pk, (const uint32_t *)start, keys_sorted,
                           sorted_rows, gend, g.want_gs ? gs_of : (uint32_t *)nullptr, tc, static_cast<IxStatSlot *>(stat_scratch), flags,
                           leaders ? *leaders : IxLeaders());
        // (a workgroup per CU at most; with no big bucket -- the usual case -- they leave at once)
        hipLaunchKernelGGL(ix_big_bucket_kernel, dim3(256), dim3(IX_NT4), IXB_BYTES, stream, g, pk, (const uint32_t *)start,
                           (const uint32_t *)big_v, keys_sorted, sorted_rows, gend, g.want_gs ? gs_of : (uint32_t *)nullptr, tc,
                           static_cast<IxStatSlot *>(stat_scratch), flags, leaders ? *leaders : IxLeaders());
        hipLaunchKernelGGL(ix_stat_reduce_kernel, dim3(1), dim3(256), 0, stream, static_cast<const IxStatSlot *>(stat_scratch), incidences, max_group, groups);
    }
    if (!(stages & 4)) return hipGetLastError();
    {
        const uint32_t nchunk = (g.rs + IX5_CH - 1u) / IX5_CH;
        const uint32_t pieces = 8u * ((g.nblk * nchunk + 7u) / 8u);
        hipLaunchKernelGGL(ix_images_kernel, dim3(pieces), dim3(256), 0, stream, g, off, (const uint32_t *)flags, (const uint2 *)tc, code_img, pos_img, nchunk);
    }
    return hipGetLastError();
}

}  // namespace mg
