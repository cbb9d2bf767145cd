// compare_join.hip — gfx950: the JOIN engine, for collections in the middle of the similarity range.
//
// The inverted-index engine (compare_sparse.hip) pays per CANDIDATE: a pair that shares a hash is merged in full,
// ~2 s steps, one lane per pair -- what the reference pays for every pair (CommandDistance.cpp:347-365).  In a
// collection of one species every pair shares a tenth to a half of its values and none is a near-copy (no dense
// group, compare_dense.hip): every pair is a candidate and the engine is 300 x under its own headline.  This engine pays
// per SHARED VALUE instead.  Its arithmetic (tests/test_join_model.py: numpy model against the oracle):
//
//     walk the values two rows have in common in ascending order, c = the common values counted so far; value v at
//     position p_i of row i and p_j of row j has rank p_i + p_j - c in the union of the two rows, and the loop of
//     compareSketches counts v iff that rank is below s (its `denom` IS the rank of the value it looks at); after one
//     value has failed every later one fails.  numer = c, denom = min(s, |A| + |B| - c).
//
// Structure: rows in BLOCKS of 64.  Per block the LIST of its entries (value, row, position in the row) in value
// order, as groups of equal values (jn_* build kernels: one radix sort of (block, value id) keys -- the value ids are the
// codes of the inverted index, monotone in the values -- then heads, a scan, the group records).  A TILE is a pair of
// blocks (I, J); one wave takes a tile: it intersects the two lists' group ids 64 x 64 at a time (a bisection through
// the wave's lanes), and for every matched value updates c[i][j] of the value's holders -- the tile's 64 x 64 counters
// are u16 in LDS, the lanes stand for the holders on the longer side, the loop runs over the shorter one, so every
// instruction touches distinct counters (row stride of 33 dwords: distinct banks).  One value after the other, in
// ascending order: every pair sees its common values in the order the reference's merge does.  At the end the tile writes
// {c, min(s, |A| + |B| - c)} for EVERY pair of it -- pairs that share nothing come out as the fill would write them: this
// engine needs no fill, no discovery, no candidate list, no scatter.
// A tile stops early once no pair of it can count another value: T[b][k] = the largest value id any row of block b has
// at position k s / 16; beyond it every row of the block has passed that position, and with the tile's largest counter
// the rank of anything still to come is bounded from below.
// Cost: one LDS read-modify-write per (pair, shared value) + ~(groups of I + groups of J) / 64 intersection steps per
// tile; nothing per pair but its 8 bytes of output.
#ifdef MG_HIP_EMU
#include "hipemu.h"
#else
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#endif
#include <stdint.h>

#include "compare_internal.h"

namespace mg {

constexpr uint32_t JN_B = 64;            // rows per block
constexpr uint32_t JN_STRIDE = 66;       // u16 counters per row of a tile (33 dwords: rows fall into distinct banks)
constexpr uint32_t JN_LEVELS = 16;       // positions k s / 16 at which a block's progress is known (k = 1 .. 15)

// pointers into the tile's counters keep their address space (kept in arrays they would decay to flat pointers: flat stores)
#ifdef MG_HIP_EMU
#define JN_LDS
#else
#define JN_LDS __attribute__((address_space(3)))
#endif
#ifdef MG_HIP_EMU
static inline uint32_t jn_readlane(uint32_t v, uint32_t l) { return __shfl(v, l); }
static inline uint32_t jn_ctz64(uint64_t m) { return (uint32_t)__builtin_ctzll(m); }
// A wave's LDS accesses happen in program order on the hardware; the emulator runs the lanes one after the other between
// wave operations, so where a lane reads counters OTHER lanes have written it has to wait for them there.
static inline void jn_lanes_in_step() { (void)__ballot(1); }
#else
__device__ __forceinline__ void jn_lanes_in_step() {}
// l is uniform over the wave at every call
__device__ __forceinline__ uint32_t jn_readlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint32_t jn_ctz64(uint64_t m) { return (uint32_t)__builtin_ctzll(m); }
#endif

// ------------------------------------------------------------------------------------------------ the lists
// one key per SLOT (row, position < s): (block << cb) | value id; entries that are left out (beyond the row's count; in
// `only_shared` lists the values no other row holds: code bit 0 clear) get the id 2^cb - 1, above every real one, and
// gather at the end of their block.  val = (position << 8) | row in block.
__global__ __launch_bounds__(256) void jn_emit_kernel(const uint32_t *img, uint32_t rs, const uint32_t *cnt_off, const uint32_t *rep,
                                                      uint32_t nrows, uint32_t s, uint32_t cb, uint32_t only_shared,
                                                      unsigned long long *key, uint32_t *val)
{
    const uint64_t slot = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (slot >= (uint64_t)nrows * s) return;
    const uint32_t row = (uint32_t)(slot / s), pos = (uint32_t)(slot - (uint64_t)row * s);
    const uint32_t src = rep ? rep[row] : row;             // a copy of an earlier row has no entries of its own
    const uint32_t cnt = cnt_off[src + 1] - cnt_off[src];
    uint32_t id = (1u << cb) - 1u;
    if (pos < cnt) {
        const uint32_t code = img[(uint64_t)src * rs + pos];
        if (!only_shared || (code & 1u)) id = code >> 1;
    }
    key[slot] = ((unsigned long long)(row / JN_B) << cb) | id;
    val[slot] = (pos << 8) | (row & (JN_B - 1u));
}

__global__ __launch_bounds__(256) void jn_heads_kernel(const unsigned long long *key, uint64_t slots, uint32_t *head)
{
    const uint64_t e = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (e >= slots) return;
    head[e] = (e == 0 || key[e - 1] != key[e]) ? 1u : 0u;
}

// ginc: inclusive scan of the heads.  Group g = ginc[e] - 1 of a head e: {value id, e}; the first group of every block;
// where a block's real groups end (its last group is the one of the entries left out, if it has any)
__global__ __launch_bounds__(256) void jn_groups_kernel(const unsigned long long *key, const uint32_t *head, const uint32_t *ginc, uint64_t slots,
                                                        uint32_t cb, uint32_t nblocks, uint2 *grp, uint32_t *goff, uint32_t *gend)
{
    const uint64_t e = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (e >= slots) return;
    const uint32_t g = ginc[e] - 1u;
    if (head[e]) {
        const unsigned long long k = key[e];
        const uint32_t drop = (1u << cb) - 1u, id = (uint32_t)(k & drop), blk = (uint32_t)(k >> cb);
        grp[g] = make_uint2(id == drop ? 0xFFFFFFFFu : id, (uint32_t)e);
        if (e == 0 || (uint32_t)(key[e - 1] >> cb) != blk) goff[blk] = g;
        if (id == drop) gend[blk] = g;
    }
    if (e + 1u == slots) {
        grp[g + 1u] = make_uint2(0xFFFFFFFFu, (uint32_t)slots);
        goff[nblocks] = g + 1u;
    }
}

__global__ __launch_bounds__(256) void jn_gend_kernel(const uint32_t *goff, uint32_t nblocks, uint32_t *gend)
{
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b < nblocks && gend[b] == 0xFFFFFFFFu) gend[b] = goff[b + 1];
}

// thr[b * 16 + k] (k = 1 .. 15): the largest value id at position k s / 16 over the rows of block b; 0xFFFFFFFF if a row
// of the block is not that long (nothing is known about its pairs then).  One wave per block, lane = row.
__global__ __launch_bounds__(64) void jn_levels_kernel(const uint32_t *img, uint32_t rs, const uint32_t *cnt_off, const uint32_t *rep,
                                                       uint32_t nrows, uint32_t s, uint32_t *thr)
{
    const uint32_t b = blockIdx.x, lane = threadIdx.x, row = b * JN_B + lane;
    uint32_t src = 0, cnt = 0;
    if (row < nrows) {
        src = rep ? rep[row] : row;
        cnt = cnt_off[src + 1] - cnt_off[src];
    }
    for (uint32_t k = 1; k < JN_LEVELS; k++) {
        const uint32_t q = (uint32_t)((uint64_t)k * s / JN_LEVELS);
        uint32_t v = 0;
        if (row < nrows) v = cnt > q ? (img[(uint64_t)src * rs + q] >> 1) : 0xFFFFFFFFu;
        for (uint32_t d = 32; d; d >>= 1) {
            const uint32_t o = __shfl_xor(v, d);
            v = o > v ? o : v;
        }
        if (lane == 0) thr[b * JN_LEVELS + k] = v;
    }
    if (lane == 0) thr[b * JN_LEVELS] = 0;
}

// shared hashes of a job, exactly, from the images alone: the sum over the rows' entries of the lengths of their runs
// (what discovery would read).  One workgroup per row, a partial sum per workgroup slot.
__global__ __launch_bounds__(256) void jn_shared_kernel(const uint32_t *lo_img, const uint32_t *hi_img, uint32_t lo_shift, uint32_t rs,
                                                        const uint32_t *cnt_off, uint32_t row_begin, uint32_t row_end, unsigned long long *sum)
{
    __shared__ unsigned long long s_part[4];
    unsigned long long acc = 0;
    for (uint32_t row = row_begin + blockIdx.x; row < row_end; row += gridDim.x) {
        const uint32_t cnt = cnt_off[row + 1] - cnt_off[row];
        for (uint32_t p = threadIdx.x; p < cnt; p += 256u) {
            const uint64_t at = (uint64_t)row * rs + p;
            acc += hi_img[at] - (lo_img[at] >> lo_shift);
        }
    }
    for (uint32_t d = 32; d; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63u) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

// ------------------------------------------------------------------------------------------------ the rows' order
// Which rows share a block decides what a tile costs: 64 relatives hold the same values (few groups per block, long holder
// lists: the lanes of an update are busy), 64 strangers hold 64 different ones.  Any order is correct, so the lists are built
// on the rows in an order of their own: a row's LABEL at radius R is the smallest row that holds one of its values whose run
// has at most R holders (such a value marks a family of at most R rows; three jumps label -> label of the label close the
// chains), and the rows are sorted by (label at 4096, label at 512, label at 64, row) -- families inside families, as a tree
// of descent lists them.  One wave per row.
constexpr uint32_t JN_RADIUS0 = 64, JN_RADIUS1 = 512, JN_RADIUS2 = 4096;

__global__ __launch_bounds__(256) void jn_labels_kernel(const uint32_t *img, uint32_t rs, const uint32_t *cnt_off, const uint32_t *rep,
                                                        const uint32_t *gend, const uint32_t *sorted_rows, uint32_t nrows, uint32_t *lab0,
                                                        uint32_t *lab1, uint32_t *lab2)
{
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (row >= nrows) return;                              // (whole waves)
    const uint32_t src = rep ? rep[row] : row;             // (a copy takes its representative's family)
    const uint32_t cnt = cnt_off[src + 1] - cnt_off[src];
    uint32_t l0 = src, l1 = src, l2 = src;
    for (uint32_t p = lane; p < cnt; p += 64u) {
        const uint32_t code = img[(uint64_t)src * rs + p];
        if (code & 1u) {
            const uint32_t gs = code >> 1, len = gend[gs] - gs, f = sorted_rows[gs];
            if (len <= JN_RADIUS0) l0 = f < l0 ? f : l0;
            if (len <= JN_RADIUS1) l1 = f < l1 ? f : l1;
            if (len <= JN_RADIUS2) l2 = f < l2 ? f : l2;
        }
    }
    for (uint32_t d = 32; d; d >>= 1) {
        const uint32_t o0 = __shfl_xor(l0, d), o1 = __shfl_xor(l1, d), o2 = __shfl_xor(l2, d);
        l0 = o0 < l0 ? o0 : l0;
        l1 = o1 < l1 ? o1 : l1;
        l2 = o2 < l2 ? o2 : l2;
    }
    if (lane == 0) { lab0[row] = l0; lab1[row] = l1; lab2[row] = l2; }
}

__global__ __launch_bounds__(256) void jn_jump_kernel(const uint32_t *in0, const uint32_t *in1, const uint32_t *in2, uint32_t nrows, uint32_t *out0,
                                                      uint32_t *out1, uint32_t *out2)
{
    const uint32_t row = blockIdx.x * 256u + threadIdx.x;
    if (row >= nrows) return;
    out0[row] = in0[in0[row]];
    out1[row] = in1[in1[row]];
    out2[row] = in2[in2[row]];
}

// (split: the rows from `split` on in a segment of their own behind the others -- a job over the rows [split, n) is then a range
//  of the lists' rows, and every pair of it is a (row, earlier row) of the lists)
__global__ __launch_bounds__(256) void jn_order_keys_kernel(const uint32_t *lab0, const uint32_t *lab1, const uint32_t *lab2, uint32_t nrows, uint32_t bits,
                                                            uint32_t split, unsigned long long *key, uint32_t *val)
{
    const uint32_t row = blockIdx.x * 256u + threadIdx.x;
    if (row >= nrows) return;
    unsigned long long k = lab0[row];
    k |= (unsigned long long)lab1[row] << bits;
    const bool three = 3u * bits + 1u <= 64u;
    if (three) k |= (unsigned long long)lab2[row] << (2u * bits);
    k |= (unsigned long long)(row >= split ? 1u : 0u) << (three ? 3u * bits : 2u * bits);
    key[row] = k;
    val[row] = row;
}

// perm[a] = the index row that is row a of the lists; src[a] = the index row whose entries speak for it (a copy: its
// representative), map[a] = its row in the TABLE (inv: the index itself may stand on a permuted table)
__global__ __launch_bounds__(256) void jn_order_maps_kernel(const uint32_t *perm, const uint32_t *rep, const uint32_t *inv, uint32_t nrows, uint32_t *src,
                                                            uint32_t *map)
{
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a >= nrows) return;
    const uint32_t r = perm[a];
    src[a] = rep ? rep[r] : r;
    map[a] = inv ? inv[r] : r;
}

// ------------------------------------------------------------------------------------------------ the tiles
#ifndef MG_HIP_EMU
// c + (c > lim): a compare into the carry and an add with carry
__device__ __forceinline__ uint32_t jn_bump(uint32_t c, int lim)
{
    uint32_t r;
    asm volatile("v_cmp_lt_i32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, 0, %1, vcc" : "=v"(r) : "v"(c), "v"(lim) : "vcc");
    return r;
}
#endif

// the holders of one matched value: `lanes` entries stand in the lanes (l < nl), `loop` entries are walked (n of them, held
// by the lanes of `loopv`).  lanes_are_rows: the lanes' holders are rows of I (the loop's: columns of J), else the reverse.
// Every lane of the wave touches a counter of its own (distinct rows, or distinct columns), and the counters of two loop
// entries differ too: four of them are read before any is written back -- the reads' round trips overlap.
template <bool DIAG>
__device__ __forceinline__ void jn_update(JN_LDS uint16_t *cnt, uint32_t lane, uint32_t lanev, uint32_t nl, uint32_t loopv, uint32_t n,
                                          bool lanes_are_rows, int s)
{
    const uint32_t lr = lanev & 0xFFu;
    const int q = (int)(lanev >> 8) - s;                   // counted iff c > p_i + p_j - s  (rank p_i + p_j - c below s)
    const uint32_t base = lanes_are_rows ? lr * JN_STRIDE : lr;
    const uint32_t step = lanes_are_rows ? 1u : JN_STRIDE;
    // one entry of the loop side: its counter, and whether this lane's pair with it is one of the tile (the tile on the
    // diagonal joins a block with itself: a pair is (row, column below it))
#define JN_ENTRY(o, at, lim, ok)                                                  \
    JN_LDS uint16_t *at = cnt + base + ((o) & 0xFFu) * step;                      \
    const int lim = q + (int)((o) >> 8);                                          \
    const bool ok = !DIAG || (lanes_are_rows ? ((o) & 0xFFu) < lr : lr < ((o) & 0xFFu));
#ifdef MG_HIP_EMU
    for (uint32_t t = 0; t < n; t++) {
        const uint32_t o = jn_readlane(loopv, t);          // (a wave operation: every lane takes part)
        JN_ENTRY(o, at, lim, ok)
        if (lane < nl && ok) {
            const int c = (int)*at;
            if (c > lim) *at = (uint16_t)(c + 1);
        }
    }
#else
    n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
    if (lane < nl) {
        uint32_t t = 0;
        if (!DIAG && n >= 8u) {
            // eight entries in flight: the counter of entry t + 8 is requested as soon as the one of entry t is written back (the
            // entries of one value are distinct holders: distinct counters), so a round trip to the LDS is hidden behind seven
            // others instead of being waited for
            uint32_t o[8], c[8];
            JN_LDS uint16_t *at[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                o[u] = jn_readlane(loopv, (uint32_t)u);
                at[u] = cnt + base + (o[u] & 0xFFu) * step;
                c[u] = *at[u];
            }
            for (; t + 16u <= n; t += 8u) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    *at[u] = (uint16_t)jn_bump(c[u], q + (int)(o[u] >> 8));
                    o[u] = jn_readlane(loopv, t + 8u + (uint32_t)u);
                    at[u] = cnt + base + (o[u] & 0xFFu) * step;
                    c[u] = *at[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) *at[u] = (uint16_t)jn_bump(c[u], q + (int)(o[u] >> 8));
            t += 8u;
        }
        for (; t + 4u <= n; t += 4u) {
            const uint32_t o0 = jn_readlane(loopv, t), o1 = jn_readlane(loopv, t + 1u), o2 = jn_readlane(loopv, t + 2u), o3 = jn_readlane(loopv, t + 3u);
            JN_ENTRY(o0, at0, lim0, ok0)
            JN_ENTRY(o1, at1, lim1, ok1)
            JN_ENTRY(o2, at2, lim2, ok2)
            JN_ENTRY(o3, at3, lim3, ok3)
            const int c0 = (int)*at0, c1 = (int)*at1, c2 = (int)*at2, c3 = (int)*at3;
            *at0 = (uint16_t)(c0 + ((c0 > lim0 && ok0) ? 1 : 0));
            *at1 = (uint16_t)(c1 + ((c1 > lim1 && ok1) ? 1 : 0));
            *at2 = (uint16_t)(c2 + ((c2 > lim2 && ok2) ? 1 : 0));
            *at3 = (uint16_t)(c3 + ((c3 > lim3 && ok3) ? 1 : 0));
        }
        for (; t < n; t++) {
            const uint32_t o = jn_readlane(loopv, t);
            JN_ENTRY(o, at, lim, ok)
            const int c = (int)*at;
            *at = (uint16_t)(c + ((c > lim && ok) ? 1 : 0));
        }
    }
#endif
#undef JN_ENTRY
}

// One tile.  Three things are a round trip to memory away -- the next 64 group records of either list, the holders of a
// step's first matched value, the holders of the value after the one being worked on -- and each is requested one stage
// before it is needed: the windows of step k + 1 when those of step k have arrived (the two lists' largest ids decide which
// list moves on), the first holders of step k while the matches of step k - 1 are worked on, every further match one ahead.
template <bool DIAG>
__device__ __forceinline__ void jn_tile(const JoinArgs &a, JN_LDS uint16_t *cnt, uint32_t lane, uint32_t bi, uint32_t bj)
{
    const JoinSide &RI = a.rows, &CJ = a.cols;
    uint32_t ga = RI.goff[bi], gb = CJ.goff[bj];
    const uint32_t ga_end = RI.gend[bi], gb_end = CJ.gend[bj];
    const int s = (int)a.s;
    // progress levels (early stop)
    const uint32_t *thrI = RI.thr ? RI.thr + (uint64_t)bi * JN_LEVELS : nullptr, *thrJ = CJ.thr ? CJ.thr + (uint64_t)bj * JN_LEVELS : nullptr;
    uint32_t kI = 0, kJ = 0;
    uint32_t nextI = thrI ? thrI[1] : 0xFFFFFFFFu, nextJ = thrJ ? thrJ[1] : 0xFFFFFFFFu;
    bool done = false;

    // a window: 64 group records of either list (value id, first entry, one past the last entry)
    struct Win { uint32_t ax, ay, ae, bx, by, be, na, nb; };
    auto load_win = [&](uint32_t pa, uint32_t pb) {
        Win w;
        w.na = ga_end - pa < 64u ? ga_end - pa : 64u;
        w.nb = gb_end - pb < 64u ? gb_end - pb : 64u;
        w.ax = w.bx = 0xFFFFFFFFu;
        w.ay = w.ae = w.by = w.be = 0;
        if (lane < w.na) { const uint2 r = RI.grp[pa + lane]; w.ax = r.x; w.ay = r.y; w.ae = RI.grp[pa + lane + 1u].y; }
        if (lane < w.nb) { const uint2 r = CJ.grp[pb + lane]; w.bx = r.x; w.by = r.y; w.be = CJ.grp[pb + lane + 1u].y; }
        return w;
    };
    // the matches of a step: per lane of list I its group's entries and those of the group of list J that holds the same
    // value; m: the lanes that found one; cur*: the holders of the first of them, requested when the step was prepared
    struct Step { uint32_t id, a0, a1, b0, b1; uint64_t m; uint32_t curA, curB, cna, cnb, cid; };
    auto fetch = [&](const Step &S, uint32_t l, uint32_t &va, uint32_t &vb, uint32_t &na, uint32_t &nb, uint32_t &id) {
        const uint32_t a0 = jn_readlane(S.a0, l), b0 = jn_readlane(S.b0, l);
        na = jn_readlane(S.a1, l) - a0;
        nb = jn_readlane(S.b1, l) - b0;
        id = jn_readlane(S.id, l);
        va = vb = 0;
        if (lane < na) va = RI.ent[a0 + lane];
        if (lane < nb) vb = CJ.ent[b0 + lane];
    };
    auto prepare = [&](const Win &w) {
        // every lane looks its value id up among the 64 of the other list: lower bound by bisection through the lanes
        uint32_t pos = 0;
        for (uint32_t st = 32; st; st >>= 1) {
            const uint32_t v = __shfl(w.bx, pos + st - 1u);
            if (v < w.ax) pos += st;
        }
        Step S;
        S.id = w.ax; S.a0 = w.ay; S.a1 = w.ae;
        const uint32_t hitv = __shfl(w.bx, pos);
        S.b0 = __shfl(w.by, pos);
        S.b1 = __shfl(w.be, pos);
        S.m = __ballot(hitv == w.ax && w.ax != 0xFFFFFFFFu);
        S.curA = S.curB = S.cna = S.cnb = S.cid = 0;
        if (S.m) fetch(S, jn_ctz64(S.m), S.curA, S.curB, S.cna, S.cnb, S.cid);
        return S;
    };
    auto process = [&](Step &S) {
        // the matched values, ascending; the holders of the next one are requested before this one's are worked on
        uint64_t m = S.m;
        uint32_t curA = S.curA, curB = S.curB, cna = S.cna, cnb = S.cnb, cid = S.cid;
        while (m) {                                         // uniform
            m &= m - 1;
            uint32_t nxtA = 0, nxtB = 0, nna = 0, nnb = 0, nid = 0;
            if (m) fetch(S, jn_ctz64(m), nxtA, nxtB, nna, nnb, nid);
            // early stop: what the blocks' levels say about the positions of this and every later value
            if (cid > nextI || cid > nextJ) {
                while (kI + 1u < JN_LEVELS && cid > nextI) { kI++; nextI = kI + 1u < JN_LEVELS ? thrI[kI + 1u] : 0xFFFFFFFFu; }
                while (kJ + 1u < JN_LEVELS && cid > nextJ) { kJ++; nextJ = kJ + 1u < JN_LEVELS ? thrJ[kJ + 1u] : 0xFFFFFFFFu; }
                // every row of I holds more than kI s / 16 values below this one, every row of J more than kJ s / 16
                // (level 0 says nothing)
                const uint32_t floor_rank = (kI ? (uint32_t)((uint64_t)kI * a.s / JN_LEVELS) + 1u : 0u) + (kJ ? (uint32_t)((uint64_t)kJ * a.s / JN_LEVELS) + 1u : 0u);
                if (floor_rank >= a.s) {
                    jn_lanes_in_step();
                    uint32_t cm = 0;
                    const JN_LDS uint32_t *c32 = (const JN_LDS uint32_t *)cnt;
                    for (uint32_t u = lane; u < JN_B * JN_STRIDE / 2u; u += 64u) {
                        const uint32_t x = c32[u], lo = x & 0xFFFFu, hi = x >> 16;
                        cm = lo > cm ? lo : cm;
                        cm = hi > cm ? hi : cm;
                    }
                    for (uint32_t d = 32; d; d >>= 1) {
                        const uint32_t o = __shfl_xor(cm, d);
                        cm = o > cm ? o : cm;
                    }
                    if (floor_rank - cm >= a.s) { done = true; break; }
                }
            }
            if (cna >= cnb) jn_update<DIAG>(cnt, lane, curA, cna, curB, cnb, true, s);
            else jn_update<DIAG>(cnt, lane, curB, cnb, curA, cna, false, s);
            curA = nxtA; curB = nxtB; cna = nna; cnb = nnb; cid = nid;
        }
    };

    if (!(ga < ga_end && gb < gb_end)) return;             // uniform
    Win w = load_win(ga, gb);
    Step prev;
    prev.m = 0;
    prev.id = prev.a0 = prev.a1 = prev.b0 = prev.b1 = prev.curA = prev.curB = prev.cna = prev.cnb = prev.cid = 0;
    bool have = true;
    while (have && !done) {                                 // uniform
        // which list moves on: the one whose 64 ids end lower (both if they end alike)
        const uint32_t amax = jn_readlane(w.ax, w.na - 1u), bmax = jn_readlane(w.bx, w.nb - 1u);
        if (amax <= bmax) ga += w.na;
        if (bmax <= amax) gb += w.nb;
        have = ga < ga_end && gb < gb_end;
        Win wn = w;
        if (have) wn = load_win(ga, gb);
        Step cur = prepare(w);
        process(prev);
        prev = cur;
        w = wn;
    }
    if (!done) process(prev);
}

// TPW tiles (waves) per workgroup.  Nothing is shared between the waves of a workgroup but its life: with four tiles a
// workgroup's wave slots and LDS stay taken until its slowest tile is done -- tiles differ in what they hold -- and a CU
// averaged 12 of its 16 waves; with one tile per workgroup every slot is refilled by itself.
template <uint32_t TPW>
__global__ __launch_bounds__(64 * TPW) void jn_tile_kernel(JoinArgs a)
{
    __shared__ uint32_t s_cnt[TPW][JN_B * JN_STRIDE / 2u];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    // workgroups are dealt round-robin to the eight XCDs: XCD x takes the x-th contiguous eighth of the tile sequence, so
    // the tiles that share a block's list run in one L2
    uint64_t wg = blockIdx.x;
    {
        const uint32_t nwg = gridDim.x, q = nwg >> 3, r = nwg & 7u, x = blockIdx.x & 7u, k = blockIdx.x >> 3;
        wg = (uint64_t)x * q + (x < r ? x : r) + k;
    }
    const uint64_t tile = wg * TPW + w;
    if (tile >= a.ntiles) return;                           // (the whole wave; no workgroup barrier below)
    uint32_t bi, bj;
    if (a.triangle) {
        // row blocks bi0 .. ; block bi has the column blocks 0 .. bi
        const uint64_t t = tile + (uint64_t)a.bi0 * (a.bi0 + 1u) / 2u;
        uint64_t b = (uint64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while (b * (b + 1u) / 2u > t) b--;
        while ((b + 1u) * (b + 2u) / 2u <= t) b++;
        bi = (uint32_t)b;
        bj = (uint32_t)(t - b * (b + 1u) / 2u);
    } else {
        bi = a.bi0 + (uint32_t)(tile / a.ncb);
        bj = (uint32_t)(tile % a.ncb);
    }
    JN_LDS uint16_t *cnt = (JN_LDS uint16_t *)&s_cnt[w][0];
    for (uint32_t u = lane; u < JN_B * JN_STRIDE / 2u; u += 64u) s_cnt[w][u] = 0;
    const bool diag = a.triangle && bi == bj;
    if (diag) jn_tile<true>(a, cnt, lane, bi, bj);
    else jn_tile<false>(a, cnt, lane, bi, bj);
    // ---- the tile's results: lane = column, row after row
    jn_lanes_in_step();
    const uint32_t j = bj * JN_B + lane;
    uint32_t nj = 0, tj = j;
    if (j < a.ncols) {
        const uint32_t sj = a.col_rep ? a.col_rep[j] : j;
        nj = a.col_cnt_off[sj + 1] - a.col_cnt_off[sj];
        if (a.inv) tj = a.inv[j];
    }
    for (uint32_t r = 0; r < JN_B; r++) {
        const uint32_t i = bi * JN_B + r;
        if (i < a.row_begin || i >= a.row_end) continue;   // uniform
        const uint32_t si = a.rep ? a.rep[i] : i;
        const uint32_t ni = a.row_cnt_off[si + 1] - a.row_cnt_off[si];
        if (j >= a.ncols || (a.triangle && j >= i)) continue;
        const uint32_t c = cnt[r * JN_STRIDE + lane];
        const uint32_t un = ni + nj - c;
        const uint2 v = make_uint2(c, un < a.s ? un : a.s);
        uint64_t o;
        if (a.triangle) {
            uint32_t hi = i, lo = j;
            if (a.inv) {                                    // (an index built on a permuted table: back to the table's rows)
                const uint32_t ti = a.inv[i];
                hi = ti > tj ? ti : tj;
                lo = ti > tj ? tj : ti;
            }
            o = (uint64_t)hi * (hi - 1u) / 2u + lo - a.out_base;
        } else {
            o = (uint64_t)(i - a.row_begin) * a.ncols + j;
        }
        a.out[o] = v;
    }
}

hipError_t launch_join_tiles(const JoinArgs &a, hipStream_t stream)
{
    if (a.ntiles == 0) return hipSuccess;
    const uint32_t tpw = a.tiles_per_wg == 4u ? 4u : 1u;
    const uint64_t nwg = (a.ntiles + tpw - 1u) / tpw;
    if (nwg > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (tpw == 4u) hipLaunchKernelGGL(jn_tile_kernel<4>, dim3((uint32_t)nwg), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(jn_tile_kernel<1>, dim3((uint32_t)nwg), dim3(64), 0, stream, a);
    return hipGetLastError();
}

uint32_t join_block_rows() { return JN_B; }
uint32_t join_levels() { return JN_LEVELS; }

hipError_t launch_join_levels(const uint32_t *img, uint32_t rs, const uint32_t *cnt_off, const uint32_t *rep, uint32_t nrows, uint32_t s,
                              uint32_t *thr, hipStream_t stream)
{
    if (nrows == 0) return hipSuccess;
    hipLaunchKernelGGL(jn_levels_kernel, dim3((nrows + JN_B - 1u) / JN_B), dim3(64), 0, stream, img, rs, cnt_off, rep, nrows, s, thr);
    return hipGetLastError();
}

hipError_t launch_join_shared(const uint32_t *lo_img, const uint32_t *hi_img, uint32_t lo_shift, uint32_t rs, const uint32_t *cnt_off,
                              uint32_t row_begin, uint32_t row_end, unsigned long long *sum, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(sum, 0, 8, stream);
    if (e != hipSuccess || row_begin >= row_end) return e;
    const uint32_t nb = row_end - row_begin < 2048u ? row_end - row_begin : 2048u;
    hipLaunchKernelGGL(jn_shared_kernel, dim3(nb), dim3(256), 0, stream, lo_img, hi_img, lo_shift, rs, cnt_off, row_begin, row_end, sum);
    return hipGetLastError();
}

#ifdef MG_HIP_EMU
}  // namespace mg
#include <algorithm>
#include <numeric>
#include <vector>
namespace mg {
size_t join_build_temp_bytes(uint64_t) { return 16; }
#else
size_t join_build_temp_bytes(uint64_t slots)
{
    size_t a = 0, b = 0;
    rocprim::double_buffer<unsigned long long> k(nullptr, nullptr);
    rocprim::double_buffer<uint32_t> v(nullptr, nullptr);
    rocprim::radix_sort_pairs(nullptr, a, k, v, (size_t)slots, 0u, 64u, (hipStream_t) nullptr);
    rocprim::inclusive_scan(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)slots, rocprim::plus<uint32_t>(), (hipStream_t) nullptr);
    return a > b ? a : b;
}
#endif

size_t join_order_temp_bytes(uint32_t nrows)
{
#ifdef MG_HIP_EMU
    (void)nrows;
    return 16;
#else
    size_t a = 0;
    rocprim::radix_sort_pairs(nullptr, a, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                              (size_t)nrows, 0u, 64u, (hipStream_t) nullptr);
    return a;
#endif
}

// The rows' order for the lists (see jn_labels_kernel).  lab [6 nrows] u32 and key_a / key_b [nrows] u64, val_a [nrows] u32: scratch;
// perm / src / map [nrows]: out.
hipError_t join_order_rows(const uint32_t *img, uint32_t rs, const uint32_t *cnt_off, const uint32_t *rep, const uint32_t *inv, const uint32_t *gend,
                           const uint32_t *sorted_rows, uint32_t nrows, void *temp, size_t temp_bytes, uint32_t *lab, unsigned long long *key_a,
                           unsigned long long *key_b, uint32_t *val_a, uint32_t *perm, uint32_t *src, uint32_t *map, hipStream_t stream, uint32_t split)
{
    if (nrows == 0) return hipSuccess;
    uint32_t *a0 = lab, *a1 = lab + nrows, *a2 = lab + 2ull * nrows, *b0 = lab + 3ull * nrows, *b1 = lab + 4ull * nrows, *b2 = lab + 5ull * nrows;
    const uint32_t g = (nrows + 255u) / 256u;
    hipLaunchKernelGGL(jn_labels_kernel, dim3((nrows + 3u) / 4u), dim3(256), 0, stream, img, rs, cnt_off, rep, gend, sorted_rows, nrows, a0, a1, a2);
    hipLaunchKernelGGL(jn_jump_kernel, dim3(g), dim3(256), 0, stream, (const uint32_t *)a0, (const uint32_t *)a1, (const uint32_t *)a2, nrows, b0, b1, b2);
    hipLaunchKernelGGL(jn_jump_kernel, dim3(g), dim3(256), 0, stream, (const uint32_t *)b0, (const uint32_t *)b1, (const uint32_t *)b2, nrows, a0, a1, a2);
    hipLaunchKernelGGL(jn_jump_kernel, dim3(g), dim3(256), 0, stream, (const uint32_t *)a0, (const uint32_t *)a1, (const uint32_t *)a2, nrows, b0, b1, b2);
    uint32_t bits = 1;
    while ((1ull << bits) < nrows) bits++;
    hipLaunchKernelGGL(jn_order_keys_kernel, dim3(g), dim3(256), 0, stream, (const uint32_t *)b0, (const uint32_t *)b1, (const uint32_t *)b2, nrows, bits, split,
                       key_a, val_a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint32_t key_bits = (3u * bits + 1u <= 64u ? 3u * bits : 2u * bits) + 1u;
#ifdef MG_HIP_EMU
    {
        std::vector<uint32_t> p(nrows);
        std::iota(p.begin(), p.end(), 0u);
        std::stable_sort(p.begin(), p.end(), [&](uint32_t x, uint32_t y) { return key_a[x] < key_a[y]; });
        for (uint32_t a = 0; a < nrows; a++) perm[a] = val_a[p[a]];
    }
    (void)key_b; (void)temp; (void)temp_bytes; (void)key_bits;
#else
    e = rocprim::radix_sort_pairs(temp, temp_bytes, (const unsigned long long *)key_a, key_b, (const uint32_t *)val_a, perm, (size_t)nrows, 0u, key_bits,
                                  stream);
    if (e != hipSuccess) return e;
#endif
    hipLaunchKernelGGL(jn_order_maps_kernel, dim3(g), dim3(256), 0, stream, (const uint32_t *)perm, rep, inv, nrows, src, map);
    return hipGetLastError();
}

// The lists of one side.  key_a / key_b [slots] u64 and val_a / val_b [slots] u32: the sort's buffers -- the entries end up
// in *ent_out (one of val_a / val_b), the keys' buffers are scratch afterwards (heads and their scan live in key_a or
// key_b, whichever the sort left free).  grp [slots + 1], goff / gend [nblocks + 1], thr [nblocks * 16].
hipError_t join_build_lists(const uint32_t *img, uint32_t rs, const uint32_t *cnt_off, const uint32_t *rep, uint32_t nrows, uint32_t s, uint32_t E,
                            bool only_shared, void *temp, size_t temp_bytes, unsigned long long *key_a, unsigned long long *key_b, uint32_t *val_a,
                            uint32_t *val_b, uint2 *grp, uint32_t *goff, uint32_t *gend, uint32_t *thr, const uint32_t **ent_out, hipStream_t stream)
{
    const uint64_t slots = (uint64_t)nrows * s;
    if (slots == 0 || slots >= (1ull << 32)) return hipErrorInvalidValue;
    const uint32_t nblocks = (nrows + JN_B - 1u) / JN_B;
    const uint32_t cb = 32u - (uint32_t)__builtin_clz(E | 1u);        // 2^cb > E > every value id
    uint32_t bb = 1;
    while ((1u << bb) < nblocks) bb++;
    const uint32_t grid = (uint32_t)((slots + 255u) / 256u);
    hipLaunchKernelGGL(jn_emit_kernel, dim3(grid), dim3(256), 0, stream, img, rs, cnt_off, rep, nrows, s, cb, only_shared ? 1u : 0u, key_a, val_a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
#ifdef MG_HIP_EMU
    // (the emulator's stand-in for the device sort: the same stable order)
    {
        std::vector<uint32_t> perm(slots);
        std::iota(perm.begin(), perm.end(), 0u);
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) { return key_a[x] < key_a[y]; });
        for (uint64_t e2 = 0; e2 < slots; e2++) { key_b[e2] = key_a[perm[e2]]; val_b[e2] = val_a[perm[e2]]; }
    }
    const unsigned long long *keys = key_b;
    uint32_t *scratch = reinterpret_cast<uint32_t *>(key_a);
    uint32_t *head = scratch, *ginc = scratch + slots;
    *ent_out = val_b;
    (void)bb; (void)temp; (void)temp_bytes;
#else
    rocprim::double_buffer<unsigned long long> k(key_a, key_b);
    rocprim::double_buffer<uint32_t> v(val_a, val_b);
    e = rocprim::radix_sort_pairs(temp, temp_bytes, k, v, (size_t)slots, 0u, cb + bb, stream);
    if (e != hipSuccess) return e;
    const unsigned long long *keys = k.current();
    uint32_t *scratch = reinterpret_cast<uint32_t *>(k.alternate());  // 2 x slots u32: heads, their inclusive scan
    uint32_t *head = scratch, *ginc = scratch + slots;
    *ent_out = v.current();
#endif
    hipLaunchKernelGGL(jn_heads_kernel, dim3(grid), dim3(256), 0, stream, keys, slots, head);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
#ifdef MG_HIP_EMU
    std::partial_sum(head, head + slots, ginc);
#else
    e = rocprim::inclusive_scan(temp, temp_bytes, (const uint32_t *)head, ginc, (size_t)slots, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
#endif
    e = hipMemsetAsync(gend, 0xFF, (size_t)(nblocks + 1u) * 4u, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(jn_groups_kernel, dim3(grid), dim3(256), 0, stream, keys, (const uint32_t *)head, (const uint32_t *)ginc, slots, cb, nblocks, grp,
                       goff, gend);
    hipLaunchKernelGGL(jn_gend_kernel, dim3((nblocks + 255u) / 256u), dim3(256), 0, stream, (const uint32_t *)goff, nblocks, gend);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_join_levels(img, rs, cnt_off, rep, nrows, s, thr, stream);
}

}  // namespace mg
