// sketch.hip — gfx950 sketching kernels: canonical k-mer streaming, MurmurHash3 in
// registers, bottom-s distinct selection in LDS.
//
// Replaces addMinHashes + getHash + MinHashHeap::tryInsert + HashSet::toHashList
// (Sketch.cpp:512-583, hash.cpp:10-38, MinHashHeap.cpp:68-145, HashSet.cpp:78-118).
// The reference's result is order independent: the sketch is the s
// smallest DISTINCT hashes over all valid k-mers — so any parallel selection that
// yields that set is bit-exact.
//
// Work decomposition: a work item is a CHUNK of one sketch's byte range (a whole
// sequence when there are enough sequences to fill the chip).  One workgroup per
// chunk streams it in tiles of NT*60 k-mer start positions:
//   global --16B coalesced--> LDS tile --odd-stride ds_read_b32--> per-lane run
// Each lane rolls its ASCII/2-bit windows over 60 (28) consecutive k-mers (kmer_hash.h),
// hashes, and appends hashes below the current threshold T to an LDS candidate
// buffer (wave ballot + one LDS atomic per wave).  Every 8 k-mers the workgroup
// checks capacity; when the buffer could overflow it is compacted: bitonic sort in
// LDS, adjacent-unique, keep s, T := s-th smallest distinct.  Chunks of one sketch
// share T through a global atomicMin (any subset's s-th smallest is an upper bound
// of the final one, so a stale value is only less selective, never wrong).
// Multi-chunk sketches are finished by merge_chunks_kernel.
//
// Roofline: integer-ALU bound (10 64-bit multiplies per 21-mer), ~1 B/base of HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kmer_hash.h"
#include "kmer_stream.h"
#include "screen_internal.h"
#include "sketch_internal.h"

// This file is compiled four times (-DSK_PART=0..3, see Makefile): every part instantiates the
// K-templated kernels for eight k-mer sizes, part 0 also holds everything that is not
// templated on K.  One translation unit with all 32 x 3 x (2 + 1 + 1) kernels takes ~5 min.
#ifndef SK_PART
#define SK_PART 0
#endif
#define SK_K0 (SK_PART * 8)
#define SK_CAT2(a, b) a##b
#define SK_CAT(a, b) SK_CAT2(a, b)
#define SK_PARTFN(base) SK_CAT(base, SK_PART)

namespace mg {

// Geometry per workgroup size.  NT=256 (s <= 2048): 60 k-mer starts per lane per tile
// (15-dword lane stride, odd -> conflict-free ds_read_b32), capacity check every 8 k-mers.
// NT=1024 (s <= 12288, e.g. the s=10000 configuration): the candidate buffer takes
// 128 KB of LDS, so the tile shrinks to 28 starts per lane (7-dword stride) and the
// capacity check runs every 4 k-mers.
__host__ __device__ constexpr int sk_seg_dw(int nt) { return nt == 256 ? 2 : 1; }
constexpr int SK_E = 16;          // buffer elements per thread during unique-compaction
constexpr uint64_t HPAD = 0xFFFFFFFFFFFFFFFFULL;

// ---------------------------------------------------------------------------
// block-wide exclusive scan of one uint per thread (NT threads); returns the
// exclusive prefix, *total = sum.  s_wsum: LDS scratch of NT/64 + 1 uints.
template <int NT>
__device__ __forceinline__ uint32_t block_exscan(uint32_t v, uint32_t *s_wsum, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) s_wsum[wid] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
        uint32_t x = s_wsum[w];
        if (w < wid) woff += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

struct SelState {            // LDS-resident selection state of one workgroup
    uint64_t T;              // s-th smallest distinct so far (valid when full)
    uint32_t count;          // entries in buf
    uint32_t full;           // >= s distinct values known (or a shared threshold adopted)
    uint32_t arrive;         // monotone wave-arrival ticket (segment boundaries)
    uint32_t snap;           // `count` as seen by the last wave to finish a segment
    uint64_t T0;             // seeded threshold (HPAD = none): only hashes below it are collected at all
};

// Sort + unique + truncate the candidate buffer.  All NT threads call.
// On return: buf[0..count) ascending distinct, count <= s, T/full updated, and the
// shared global threshold (if any) exchanged.
template <int NT>
__device__ void compact_buffer(uint64_t *buf, SelState *st, uint32_t *s_wsum, uint32_t s,
                               unsigned long long *g_T)
{
    __syncthreads();
    const uint32_t n = st->count;
    const int tid = threadIdx.x;
    uint32_t P = 2;
    while (P < n) P <<= 1;
    for (uint32_t i = n + tid; i < P; i += NT) buf[i] = HPAD;
    __syncthreads();
    // bitonic sort, ascending
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (P >> 1); t += NT) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const uint32_t l = i + j;
                const uint64_t a = buf[i], b = buf[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { buf[i] = b; buf[l] = a; }
            }
            __syncthreads();
        }
    }
    // adjacent-unique through registers (reads complete before any write)
    uint64_t v[SK_E];
    uint32_t keep = 0, cnt = 0;
    const uint32_t base = tid * SK_E;
    uint64_t prev = (base > 0 && base <= n) ? buf[base - 1] : 0;
#pragma unroll
    for (int e = 0; e < SK_E; e++) {
        const uint32_t i = base + e;
        v[e] = (i < n) ? buf[i] : HPAD;
        const bool f = (i < n) && (i == 0 || v[e] != prev);
        prev = v[e];
        keep |= (f ? 1u : 0u) << e;
        cnt += f ? 1u : 0u;
    }
    uint32_t total;
    uint32_t off = block_exscan<NT>(cnt, s_wsum, &total);   // contains the barriers
#pragma unroll
    for (int e = 0; e < SK_E; e++) {
        if ((keep >> e) & 1u) {
            if (off < s) buf[off] = v[e];
            off++;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t kept = total < s ? total : s;
        st->count = kept;
        uint64_t T = st->T0;                               // a seeded threshold stays in force
        uint32_t full = T != HPAD ? 1u : 0u;
        if (total >= s) { full = 1; T = buf[s - 1]; }      // (< T0: every candidate is)
        if (g_T) {
            // exchange with the other chunks of this sketch
            if (full) atomicMin(g_T, (unsigned long long)T);
            const uint64_t gt = __hip_atomic_load(g_T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gt < T || (!full && gt != HPAD)) { T = gt; full = 1; }
        }
        st->T = T;
        st->full = full;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// MODE 0: DNA canonical, 1: DNA forward-only (-n), 2: table alphabet forward-only
template <int K, int MODE, int NT, bool PROBE>
__global__ __launch_bounds__(NT) void sketch_chunks_kernel(SketchArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int SK_L = sk_L(NT);
    constexpr int SK_SEG_DW = sk_seg_dw(NT);
    constexpr int TILE = NT * SK_L;
    constexpr int TILE_DW = TILE / 4 + 32;                 // + slack for k-1 overlap & alignment
    uint64_t *buf = reinterpret_cast<uint64_t *>(smem);                          // [cap]
    uint32_t *tile = reinterpret_cast<uint32_t *>(smem + (size_t)a.cap * 8);     // [TILE_DW]
    uint8_t *alpha = reinterpret_cast<uint8_t *>(tile + TILE_DW);                // [256] (MODE 2)
    uint32_t *s_wsum = reinterpret_cast<uint32_t *>(alpha + 256);                // [NT/64 + 1]
    SelState *st = reinterpret_cast<SelState *>(s_wsum + NT / 64 + 2);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const SketchWork w = a.work[blockIdx.x];
    unsigned long long *g_T = w.nchunks > 1 ? (unsigned long long *)&a.g_T[w.sketch] : nullptr;

    if (tid == 0) {
        // Seeded start (SketchArgs::seed_T): with L k-mers in the sketch the s-th smallest hash
        // sits near s/L of the hash range, so a threshold a few times that spares the warm-up in
        // which the candidate buffer is sorted again and again while the threshold is still loose.
        // Exact as long as the sketch ends up with s hashes below the seed; the host re-runs the
        // few that do not (repeats, mostly invalid input) without a seed.
        const uint64_t t0 = a.seed_T ? a.seed_T[w.sketch] : HPAD;
        st->count = 0; st->full = t0 != HPAD ? 1u : 0u; st->T = t0; st->T0 = t0; st->arrive = 0; st->snap = 0;
        if (g_T) {
            const uint64_t gt = __hip_atomic_load(g_T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gt < st->T) { st->T = gt; st->full = 1; }
        }
    }
    if (MODE == 2) for (int i = tid; i < 256; i += NT) alpha[i] = a.alphabet[i];
    __syncthreads();

    const uint32_t s = a.sketch_size;
    const uint32_t cap = a.cap;
    const uint32_t seed = a.seed;
    const bool use64 = a.use64 != 0;
    const bool fold = a.fold_case != 0;
    constexpr int NBYTES = SK_L + K - 1;                   // bytes one lane consumes per tile
    constexpr int ND = (NBYTES + 3) / 4;
    constexpr uint32_t NW = NT / 64;
    uint32_t seg_no = 0;                                   // segments finished (uniform)
    const unsigned long long *probe_keys = a.probe_keys;
    uint32_t *probe_obs = a.probe_obs;
    const uint64_t probe_mask = a.probe_mask, probe_max = a.probe_max, probe_tier = a.probe_tier;
    constexpr bool probing = PROBE;                        // fused table probe (mash screen) compiled in or out

    for (uint64_t t0 = w.begin; t0 < w.end; t0 += TILE) {
        // ---- stage the tile: bytes [a0, a0 + TILE_DW*4), a0 = t0 rounded down to 16 ----
        const uint64_t a0 = t0 & ~15ULL;
        const uint32_t shift = (uint32_t)(t0 - a0);
        if (a.packed) {
            // packed input: 16 bases per lane = one dword of codes (two when the batch does not start at a dword), their
            // bits of the mask; written into the tile as the characters the ASCII path would have found there
            for (int q = tid; q < TILE_DW / 4; q += NT) {
                const uint64_t o = a0 + (uint64_t)q * 16;
                uint4 x = make_uint4(0, 0, 0, 0);
                if (o < w.limit) {
                    const uint64_t t = o >> 4;
                    uint64_t cw = a.packed[t];
                    if (a.pskip) cw |= (uint64_t)a.packed[t + 1] << 32;
                    const uint32_t bits = (uint32_t)(cw >> (2u * a.pskip));
                    uint32_t inv = 0;
                    if (a.pmask) {
                        const uint64_t gm = a.pmskip + o;
                        const uint32_t sh = (uint32_t)(gm & 31u);
                        uint64_t m = a.pmask[gm >> 5];
                        if (sh > 16u) m |= (uint64_t)a.pmask[(gm >> 5) + 1u] << 32;
                        inv = (uint32_t)(m >> sh) & 0xFFFFu;
                    }
                    const uint64_t left = w.limit - o;
                    uint32_t d[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        uint32_t word = 0;
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const int i = c * 4 + b;
                            uint32_t ch = (0x47544341u >> (8u * ((bits >> (2 * i)) & 3u))) & 0xFFu;      // "ACTG"[code]
                            if ((inv >> i) & 1u) ch = 'N';
                            if ((uint64_t)i >= left) ch = 0;                                              // (behind the sketch's bytes: as the ASCII path)
                            word |= ch << (8 * b);
                        }
                        d[c] = word;
                    }
                    x = make_uint4(d[0], d[1], d[2], d[3]);
                }
                reinterpret_cast<uint4 *>(tile)[q] = x;
            }
        } else {
        for (int q = tid; q < TILE_DW / 4; q += NT) {
            const uint64_t o = a0 + (uint64_t)q * 16;
            uint4 x = make_uint4(0, 0, 0, 0);
            if (o + 16 <= w.limit) {
                x = *reinterpret_cast<const uint4 *>(a.bases + o);
            } else if (o < w.limit) {
                uint32_t d[4] = {0, 0, 0, 0};
                for (int b = 0; b < 16 && o + b < w.limit; b++)
                    d[b >> 2] |= (uint32_t)a.bases[o + b] << (8 * (b & 3));
                x = make_uint4(d[0], d[1], d[2], d[3]);
            }
            reinterpret_cast<uint4 *>(tile)[q] = x;
        }
        }
        __syncthreads();

        // ---- per-lane run ----
        const uint32_t lane_byte0 = shift + (uint32_t)tid * SK_L;
        const uint32_t *lw = tile + (lane_byte0 >> 2);
        const uint32_t bsh = lane_byte0 & 3;               // == shift & 3 (uniform)
        // k-mer starts this lane may emit: start < remaining (chunk end)
        const uint64_t rem64 = w.end - t0;
        const uint32_t remaining = rem64 > (uint64_t)TILE ? (uint32_t)TILE : (uint32_t)rem64;
        const uint32_t lane_first = (uint32_t)tid * SK_L;

        // One pass of every lane over its run.  SEG = true: capacity is checked every
        // SK_SEG_DW dwords (needed while the threshold is still loose).  SEG = false: the
        // steady state — the threshold is set, candidates are rare, so the tile runs without
        // a single barrier; appends are bounds-checked and an overflow (pathological repeats)
        // discards the tile's candidates and re-runs it with SEG = true.
        auto run_tile = [&](auto seg_tag, const bool probe_now) {
            constexpr bool SEG = decltype(seg_tag)::value;
            KmerRoller<K, MODE == 0> r;
            r.reset();
            uint64_t T = st->T;
            bool full = st->full != 0;
            uint32_t cur = lw[0];
#pragma unroll 1
            for (int d = 0; d < ND; d++) {
                const uint32_t nxt = lw[d + 1];
                const uint32_t word = __builtin_amdgcn_alignbyte(nxt, cur, bsh);
                cur = nxt;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int pos = 4 * d + b;
                    uint32_t c = (word >> (8 * b)) & 0xFFu;
                    bool valid;
                    if (MODE == 2) {
                        if (fold) c = fold_upper(c);
                        valid = alpha[c] != 0;
                        r.push(c, valid);
                    } else {
                        if (fold) c &= 0xDFu;              // DNA: only membership of ACGT matters
                        uint32_t code, comp;
                        valid = dna_classify(c, code, comp);
                        r.push(c, valid, code, comp);
                    }
                    if (pos >= K - 1 && pos < NBYTES) {    // uniform
                        const uint32_t start = (uint32_t)(pos - (K - 1));
                        const uint64_t h = r.hash(seed, use64);
                        const bool mine = r.kmer_valid() && (lane_first + start < remaining);
                        const bool pass = mine && (!full || h < T);
                        if (probe_now && mine && h <= probe_max) {   // sketches are bottom-s sets: rare
                            bool look = true;
                            if (h > probe_tier) {                // second tier: the keys of the database's small genomes
                                const uint64_t b = __umul64hi(h - probe_tier - 1, a.probe_bits_scale);
                                look = (a.probe_bits[b >> 5] >> (b & 31)) & 1u;
                            }
                            uint64_t slot;
                            if (look && scr_find(probe_keys, probe_mask, h, &slot)) {
                                if (atomicAdd(&probe_obs[slot], 1u) == 0 && a.probe_touched) {
                                    const unsigned long long i = atomicAdd(a.probe_ntouched, 1ULL);
                                    if (i < a.probe_touched_cap) a.probe_touched[i] = (uint32_t)slot;
                                }
                            }
                        }
                        const uint64_t m = __ballot(pass);
                        if (m != 0) {
                            const uint32_t off = __builtin_amdgcn_mbcnt_hi(
                                (uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                            uint32_t bpos = 0;
                            if (pass && off == 0) bpos = atomicAdd(&st->count, (uint32_t)__popcll(m));
                            bpos = __shfl(bpos, __ffsll((unsigned long long)m) - 1);
                            if (pass && (SEG || bpos + off < cap)) buf[bpos + off] = h;
                        }
                    }
                }
                if (SEG && ((d % SK_SEG_DW) == SK_SEG_DW - 1 || d == ND - 1)) {
                    // Segment boundary.  The capacity decision must be identical in every
                    // thread, but `count` keeps moving as fast waves enter the next segment:
                    // the LAST wave to arrive (monotone ticket) snapshots it — by then every
                    // wave has issued its appends — and everybody reads the snapshot after
                    // the barrier (it cannot be overwritten before all waves pass the next one).
                    seg_no++;
                    if (lane == 0) {
                        const uint32_t t = atomicAdd(&st->arrive, 1u);
                        if (t == NW * seg_no - 1) st->snap = st->count;
                    }
                    __syncthreads();
                    if (st->snap + (uint32_t)NT * 4 * SK_SEG_DW > cap) {    // uniform
                        compact_buffer<NT>(buf, st, s_wsum, s, g_T);
                        T = st->T;
                        full = st->full != 0;
                    }
                }
            }
        };
        uint32_t count0 = st->count;
        const bool steady = st->full != 0 && count0 + (uint32_t)NT <= cap;
        __syncthreads();                                   // everybody has read count0 / full
        if (!steady && count0 + (uint32_t)NT * 4 * SK_SEG_DW > cap) {
            // A segmented pass checks the capacity AFTER each segment, so it must start with room for
            // one (steady tiles may leave the buffer fuller than that: they have no check at their end).
            // Without this, input whose k-mers keep passing the threshold -- a handful of distinct k-mers,
            // low-complexity runs -- wrote its first segment past the buffer into the staged tile.
            compact_buffer<NT>(buf, st, s_wsum, s, g_T);
            count0 = st->count;
        }
        if (steady) {
            run_tile(std::false_type{}, probing);
            __syncthreads();
            if (st->count > cap) {                         // uniform: overflow, redo the tile carefully
                __syncthreads();
                if (tid == 0) st->count = count0;
                compact_buffer<NT>(buf, st, s_wsum, s, g_T);
                run_tile(std::true_type{}, false);         // its k-mers were already probed
            }
        } else {
            run_tile(std::true_type{}, probing);
        }
        __syncthreads();      // tile may be overwritten
    }

    compact_buffer<NT>(buf, st, s_wsum, s, g_T);
    const uint32_t n = st->count;
    if (w.nchunks == 1) {
        uint64_t *out = a.hashes_out + (uint64_t)w.sketch * s;
        for (uint32_t i = tid; i < s; i += NT) out[i] = i < n ? buf[i] : HPAD;
        if (tid == 0) a.nhash_out[w.sketch] = n;
    } else {
        uint64_t *out = a.pool + (uint64_t)w.slot * s;
        for (uint32_t i = tid; i < n; i += NT) out[i] = buf[i];
        if (tid == 0) a.pool_n[w.slot] = n;
    }
}

#if SK_PART == 0
// One workgroup per multi-chunk sketch: bottom-s distinct of the union of its chunk lists.
template <int NT>
__global__ __launch_bounds__(NT) void merge_chunks_kernel(MergeArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *buf = reinterpret_cast<uint64_t *>(smem);
    uint32_t *s_wsum = reinterpret_cast<uint32_t *>(smem + (size_t)a.cap * 8);
    SelState *st = reinterpret_cast<SelState *>(s_wsum + NT / 64 + 2);
    const int tid = threadIdx.x;
    const MergeWork w = a.work[blockIdx.x];
    const uint32_t s = a.sketch_size, cap = a.cap;
    if (tid == 0) { st->count = 0; st->full = 0; st->T = HPAD; st->T0 = HPAD; st->arrive = 0; st->snap = 0; }
    __syncthreads();
    const uint32_t batch = cap - s;                        // room guaranteed after a compaction
    for (uint32_t c = 0; c < w.nchunks; c++) {
        const uint32_t slot = w.first_slot + c * w.stride;
        const uint32_t n = a.pool_n[slot];
        const uint64_t *src = a.pool + (uint64_t)slot * s;
        for (uint32_t b0 = 0; b0 < n; b0 += batch) {
            const uint32_t nb = (n - b0) < batch ? (n - b0) : batch;
            __syncthreads();
            if (st->count + nb > cap) compact_buffer<NT>(buf, st, s_wsum, s, nullptr);
            const uint64_t T = st->T;
            const bool full = st->full != 0;
            __syncthreads();
            for (uint32_t i = tid; i < nb; i += NT) {
                const uint64_t h = src[b0 + i];
                if (!full || h < T) buf[atomicAdd(&st->count, 1u)] = h;
            }
        }
    }
    compact_buffer<NT>(buf, st, s_wsum, s, nullptr);
    const uint32_t n = st->count;
    if (w.to_pool) {                                       // all reads of this group's slots are done
        uint64_t *out = a.pool + (uint64_t)w.first_slot * s;
        for (uint32_t i = tid; i < n; i += NT) out[i] = buf[i];
        if (tid == 0) a.pool_n[w.first_slot] = n;
        return;
    }
    uint64_t *out = a.hashes_out + (uint64_t)w.sketch * s;
    for (uint32_t i = tid; i < s; i += NT) out[i] = i < n ? buf[i] : HPAD;
    if (tid == 0) a.nhash_out[w.sketch] = n;
}

#endif  // SK_PART == 0

// ---------------------------------------------------------------------------
// Multiplicities (Sketch::Reference::counts, HashSet.cpp:78-118 / MinHashHeap.cpp:96-124).
// Every kept hash except the largest is counted on each occurrence in the reference, so its
// count is its exact multiplicity.  The largest kept hash h_max is only counted while the
// heap is not yet "full with h_max on top": occurrences after t* = max over kept hashes of
// their FIRST occurrence are dropped (MinHashHeap.cpp:70-74 tests `hash < top`).
// Phase 0 re-streams the chunk, looks every hash <= h_max up in the final sketch (LDS, binary
// search) and accumulates multiplicity + first position; launch_count_tstar derives t*; phase 1
// (only sketches whose h_max repeats) recounts h_max over positions <= t*.
template <int K, int MODE>
__global__ __launch_bounds__(256) void count_chunks_kernel(CountArgs a)
{
    constexpr int NT = 256;
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *hl = reinterpret_cast<uint64_t *>(smem);                              // [s]
    uint32_t *tile = reinterpret_cast<uint32_t *>(smem + (size_t)a.sketch_size * 8);
    uint8_t *alpha = reinterpret_cast<uint8_t *>(tile + sk_tile_dw(NT));

    const int tid = threadIdx.x;
    const SketchWork w = a.work[blockIdx.x];
    const uint32_t s = a.sketch_size;
    const uint32_t n = a.nhash[w.sketch];
    if (n == 0) return;
    const uint64_t *row = a.hashes + (uint64_t)w.sketch * s;
    for (uint32_t i = tid; i < n; i += NT) hl[i] = row[i];
    if (MODE == 2) for (int i = tid; i < 256; i += NT) alpha[i] = a.alphabet[i];
    __syncthreads();
    const uint64_t T = hl[n - 1];
    const uint64_t tstar = a.phase ? a.tstar[w.sketch] : 0;
    uint32_t *cnt = a.counts + (uint64_t)w.sketch * s;
    unsigned long long *fp = a.firstpos + (uint64_t)w.sketch * s;
    const uint32_t phase = a.phase;
    const unsigned long long *prev = a.prevpos ? a.prevpos + (uint64_t)w.sketch * s : nullptr;
    stream_chunk<K, MODE, NT>(a.bases, w, tile, alpha, a.fold_case != 0, a.seed, a.use64 != 0,
                              [&](uint64_t h, uint64_t kpos) {
        if (h > T) return;
        if (phase == 2) {                                  // position of the next occurrence after prev[]
            uint32_t lo = 0, hi = n;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (hl[mid] < h) lo = mid + 1; else hi = mid;
            }
            if (lo < n && hl[lo] == h && (unsigned long long)kpos > prev[lo])
                atomicMin(&fp[lo], (unsigned long long)kpos);
            return;
        }
        if (phase == 0) {
            uint32_t lo = 0, hi = n;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (hl[mid] < h) lo = mid + 1; else hi = mid;
            }
            if (lo < n && hl[lo] == h) {                   // minCov 1: always true (every value <= h_max is kept)
                atomicAdd(&cnt[lo], 1u);
                atomicMin(&fp[lo], (unsigned long long)kpos);
            }
        } else if (h == T && kpos <= tstar) {
            atomicAdd(&cnt[n - 1], 1u);
        }
    });
}

#if SK_PART == 0
// one thread per sketch: t* = latest first occurrence among the kept hashes of a FULL sketch;
// the largest hash needs the positional recount only if it repeats
__global__ void count_tstar_kernel(const uint32_t *nhash, uint32_t *counts, const unsigned long long *firstpos,
                                   unsigned long long *tstar, uint32_t *need_fix, uint32_t nsketch, uint32_t s)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsketch) return;
    const uint32_t n = nhash[i];
    uint32_t fix = 0;
    unsigned long long t = 0;
    if (n == s && counts[(uint64_t)i * s + n - 1] > 1) {
        for (uint32_t k = 0; k < n; k++) {
            const unsigned long long f = firstpos[(uint64_t)i * s + k];
            t = f > t ? f : t;
        }
        fix = 1;
        counts[(uint64_t)i * s + n - 1] = 0;
    }
    tstar[i] = t;
    need_fix[i] = fix;
}

#endif  // SK_PART == 0

// ---------------------------------------------------------------------------
// minCov >= 2 (`mash sketch -m`, MinHashHeap.cpp:96-118): a hash enters the sketch at its m-th
// occurrence.  The result is order independent -- it is the s smallest hashes of
// Q = {h : multiplicity(h) >= m} (a member of that set passes the `size < s || hash < top` test
// at every occurrence before its promotion, is never evicted, and the pending-set clean-up only
// drops hashes above an evicted top; checked against the reference's objects in
// tests/golden/ref_sketch_vectors_m.npz).  Q is found by exact counting: all k-mers whose
// hash lies in [lo, hi] are counted in an open-addressing table in HBM (the range is sized from
// the k-mer total so that it holds a bounded number of distinct hashes), survivors with count
// >= m are extracted, and the range moves up until s survivors exist or the hash space ends.
template <int K, int MODE>
__global__ __launch_bounds__(256) void range_count_kernel(RangeCountArgs a)
{
    constexpr int NT = 256;
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *tile = reinterpret_cast<uint32_t *>(smem);
    uint8_t *alpha = reinterpret_cast<uint8_t *>(tile + sk_tile_dw(NT));
    if (MODE == 2) {
        for (int i = threadIdx.x; i < 256; i += NT) alpha[i] = a.alphabet[i];
        __syncthreads();
    }
    const SketchWork w = a.work[blockIdx.x];
    unsigned long long *keys = a.keys;
    uint32_t *cnts = a.cnts;
    const uint64_t mask = a.mask, lo = a.lo, hi = a.hi;
    stream_chunk<K, MODE, NT>(a.bases, w, tile, alpha, a.fold_case != 0, a.seed, a.use64 != 0,
                              [&](uint64_t h, uint64_t) {
        if (h < lo || h > hi) return;
        uint64_t slot = scr_slot(h, mask);
        for (uint32_t step = 0; step < 4096; step++) {
            const unsigned long long old = atomicCAS(&keys[slot], SCR_EMPTY, (unsigned long long)h);
            if (old == SCR_EMPTY || old == h) { atomicAdd(&cnts[slot], 1u); return; }
            slot = (slot + 1) & mask;
        }
        atomicOr(a.overflow, 1u);                          // table too full: the host retries a narrower range
    });
}

#if SK_PART == 0
__global__ void range_extract_kernel(const unsigned long long *keys, const uint32_t *cnts, uint64_t slots,
                                     uint32_t min_copies, unsigned long long *out, unsigned long long *out_n,
                                     uint64_t out_cap)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += stride) {
        if (keys[i] != SCR_EMPTY && cnts[i] >= min_copies) {
            const unsigned long long at = atomicAdd(out_n, 1ull);
            if (at < out_cap) out[at] = keys[i];
        }
    }
}

#endif  // SK_PART == 0

// `mash sketch -r -c <cov>` stops reading once the heap's average multiplicity reaches the target
// (Sketch.cpp:1258) -- a property of the sequential heap.  It is reproduced exactly by replaying
// that heap on the host over a THINNED stream: only hashes below the heap's current top can ever
// change it (MinHashHeap.cpp:70-74), so the device emits {hash, position} of exactly those.
template <int K, int MODE>
__global__ __launch_bounds__(256) void hash_events_kernel(EventArgs a)
{
    constexpr int NT = 256;
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *tile = reinterpret_cast<uint32_t *>(smem);
    uint8_t *alpha = reinterpret_cast<uint8_t *>(tile + sk_tile_dw(NT));
    if (MODE == 2) {
        for (int i = threadIdx.x; i < 256; i += NT) alpha[i] = a.alphabet[i];
        __syncthreads();
    }
    const SketchWork w = a.work[blockIdx.x];
    const uint64_t bound = a.bound, cap = a.capacity;
    stream_chunk<K, MODE, NT>(a.bases, w, tile, alpha, a.fold_case != 0, a.seed, a.use64 != 0,
                              [&](uint64_t h, uint64_t kpos) {
        if (h > bound) return;
        const unsigned long long at = atomicAdd(a.count, 1ull);
        if (at < cap) a.out[at] = HashEvent{(unsigned long long)h, (unsigned long long)kpos};
    });
}

template <int K, int MODE>
static hipError_t launch_events_one(const EventArgs &a, uint32_t nwork, hipStream_t stream)
{
    const size_t smem = (size_t)sk_tile_dw(256) * 4 + 256 + 64;
    hipLaunchKernelGGL((hash_events_kernel<K, MODE>), dim3(nwork), dim3(256), smem, stream, a);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_events_k(int k, const EventArgs &a, uint32_t nwork, hipStream_t st)
{
    switch (k) {
#define MG_CASE(KK) case KK: return launch_events_one<KK, MODE>(a, nwork, st);
        MG_CASE(SK_K0 + 1) MG_CASE(SK_K0 + 2) MG_CASE(SK_K0 + 3) MG_CASE(SK_K0 + 4)
        MG_CASE(SK_K0 + 5) MG_CASE(SK_K0 + 6) MG_CASE(SK_K0 + 7) MG_CASE(SK_K0 + 8)
#undef MG_CASE
    }
    return hipErrorInvalidValue;
}

hipError_t SK_PARTFN(launch_events_part)(int k, int mode, const EventArgs &a, uint32_t nwork, hipStream_t stream)
{
    if (mode == 0) return launch_events_k<0>(k, a, nwork, stream);
    if (mode == 1) return launch_events_k<1>(k, a, nwork, stream);
    return launch_events_k<2>(k, a, nwork, stream);
}

#if SK_PART == 0
hipError_t launch_events_part1(int, int, const EventArgs &, uint32_t, hipStream_t);
hipError_t launch_events_part2(int, int, const EventArgs &, uint32_t, hipStream_t);
hipError_t launch_events_part3(int, int, const EventArgs &, uint32_t, hipStream_t);

hipError_t launch_hash_events(int k, int mode, const EventArgs &a, uint32_t nwork, hipStream_t stream)
{
    if (nwork == 0) return hipSuccess;
    switch ((k - 1) / 8) {
        case 0: return launch_events_part0(k, mode, a, nwork, stream);
        case 1: return launch_events_part1(k, mode, a, nwork, stream);
        case 2: return launch_events_part2(k, mode, a, nwork, stream);
        case 3: return launch_events_part3(k, mode, a, nwork, stream);
    }
    return hipErrorInvalidValue;
}
#endif  // SK_PART == 0

template <int K, int MODE>
static hipError_t launch_range_one(const RangeCountArgs &a, uint32_t nwork, hipStream_t stream)
{
    const size_t smem = (size_t)sk_tile_dw(256) * 4 + 256 + 64;
    hipLaunchKernelGGL((range_count_kernel<K, MODE>), dim3(nwork), dim3(256), smem, stream, a);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_range_k(int k, const RangeCountArgs &a, uint32_t nwork, hipStream_t st)
{
    switch (k) {
#define MG_CASE(KK) case KK: return launch_range_one<KK, MODE>(a, nwork, st);
        MG_CASE(SK_K0 + 1) MG_CASE(SK_K0 + 2) MG_CASE(SK_K0 + 3) MG_CASE(SK_K0 + 4)
        MG_CASE(SK_K0 + 5) MG_CASE(SK_K0 + 6) MG_CASE(SK_K0 + 7) MG_CASE(SK_K0 + 8)
#undef MG_CASE
    }
    return hipErrorInvalidValue;
}

hipError_t SK_PARTFN(launch_range_part)(int k, int mode, const RangeCountArgs &a, uint32_t nwork, hipStream_t stream)
{
    if (mode == 0) return launch_range_k<0>(k, a, nwork, stream);
    if (mode == 1) return launch_range_k<1>(k, a, nwork, stream);
    return launch_range_k<2>(k, a, nwork, stream);
}

#if SK_PART == 0
hipError_t launch_range_part1(int, int, const RangeCountArgs &, uint32_t, hipStream_t);
hipError_t launch_range_part2(int, int, const RangeCountArgs &, uint32_t, hipStream_t);
hipError_t launch_range_part3(int, int, const RangeCountArgs &, uint32_t, hipStream_t);

hipError_t launch_range_count(int k, int mode, const RangeCountArgs &a, uint32_t nwork, hipStream_t stream)
{
    if (nwork == 0) return hipSuccess;
    switch ((k - 1) / 8) {
        case 0: return launch_range_part0(k, mode, a, nwork, stream);
        case 1: return launch_range_part1(k, mode, a, nwork, stream);
        case 2: return launch_range_part2(k, mode, a, nwork, stream);
        case 3: return launch_range_part3(k, mode, a, nwork, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_range_extract(const unsigned long long *keys, const uint32_t *cnts, uint64_t slots,
                                uint32_t min_copies, unsigned long long *out, unsigned long long *out_n,
                                uint64_t out_cap, hipStream_t stream)
{
    uint64_t blocks = (slots + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(range_extract_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, keys, cnts, slots,
                       min_copies, out, out_n, out_cap);
    return hipGetLastError();
}
#endif  // SK_PART == 0

// ---------------------------------------------------------------------------
// dispatch tables

template <int K, int MODE, int NT>
static hipError_t launch_one(const SketchArgs &a, uint32_t nwork, size_t smem, hipStream_t stream)
{
    auto kern = a.probe_keys ? sketch_chunks_kernel<K, MODE, NT, true> : sketch_chunks_kernel<K, MODE, NT, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nwork), dim3(NT), smem, stream, a);
    return hipGetLastError();
}

template <int MODE, int NT>
static hipError_t launch_k(int k, const SketchArgs &a, uint32_t nwork, size_t smem, hipStream_t st)
{
    switch (k) {
#define MG_CASE(KK) case KK: return launch_one<KK, MODE, NT>(a, nwork, smem, st);
        MG_CASE(SK_K0 + 1) MG_CASE(SK_K0 + 2) MG_CASE(SK_K0 + 3) MG_CASE(SK_K0 + 4)
        MG_CASE(SK_K0 + 5) MG_CASE(SK_K0 + 6) MG_CASE(SK_K0 + 7) MG_CASE(SK_K0 + 8)
#undef MG_CASE
    }
    return hipErrorInvalidValue;
}

static size_t sketch_smem_bytes_impl(uint32_t cap, int nt)
{
    const size_t tile_dw = (size_t)nt * sk_L(nt) / 4 + 32;
    return (size_t)cap * 8 + tile_dw * 4 + 256 + ((size_t)nt / 64 + 2) * 4 + sizeof(SelState) + 16;
}

hipError_t SK_PARTFN(launch_sketch_part)(int k, int mode, int nt, const SketchArgs &a, uint32_t nwork,
                                         hipStream_t stream)
{
    const size_t smem = sketch_smem_bytes_impl(a.cap, nt);
    if (nt == 256) {
        if (mode == 0) return launch_k<0, 256>(k, a, nwork, smem, stream);
        if (mode == 1) return launch_k<1, 256>(k, a, nwork, smem, stream);
        return launch_k<2, 256>(k, a, nwork, smem, stream);
    }
    if (mode == 0) return launch_k<0, 1024>(k, a, nwork, smem, stream);
    if (mode == 1) return launch_k<1, 1024>(k, a, nwork, smem, stream);
    return launch_k<2, 1024>(k, a, nwork, smem, stream);
}

#if SK_PART == 0
size_t sketch_smem_bytes(uint32_t cap, int nt) { return sketch_smem_bytes_impl(cap, nt); }

uint32_t sketch_tile(int nt) { return (uint32_t)nt * sk_L(nt); }

// cap: power of two >= s + nt*8 (one segment of candidates always fits after a compaction)
bool sketch_geometry(uint64_t s, int *nt_out, uint32_t *cap_out)
{
    for (int nt : {256, 1024}) {
        uint64_t need = s + (uint64_t)nt * 4 * sk_seg_dw(nt);
        uint64_t cap = 2;
        while (cap < need) cap <<= 1;
        if (cap / nt > SK_E) continue;                     // unique-compaction holds cap/nt per thread
        if (sketch_smem_bytes((uint32_t)cap, nt) > 160 * 1024) continue;
        *nt_out = nt; *cap_out = (uint32_t)cap;
        return true;
    }
    return false;
}

hipError_t launch_sketch_part1(int, int, int, const SketchArgs &, uint32_t, hipStream_t);
hipError_t launch_sketch_part2(int, int, int, const SketchArgs &, uint32_t, hipStream_t);
hipError_t launch_sketch_part3(int, int, int, const SketchArgs &, uint32_t, hipStream_t);

hipError_t launch_sketch_chunks(int k, int mode, int nt, const SketchArgs &a, uint32_t nwork,
                                hipStream_t stream)
{
    switch ((k - 1) / 8) {
        case 0: return launch_sketch_part0(k, mode, nt, a, nwork, stream);
        case 1: return launch_sketch_part1(k, mode, nt, a, nwork, stream);
        case 2: return launch_sketch_part2(k, mode, nt, a, nwork, stream);
        case 3: return launch_sketch_part3(k, mode, nt, a, nwork, stream);
    }
    return hipErrorInvalidValue;
}

bool count_supported(uint64_t s)
{
    return s * 8 + ((size_t)256 * sk_L(256) / 4 + 32) * 4 + 256 + 64 <= 160 * 1024;
}
#endif  // SK_PART == 0

template <int K, int MODE>
static hipError_t launch_count_one(const CountArgs &a, uint32_t nwork, hipStream_t stream)
{
    const size_t smem = (size_t)a.sketch_size * 8 + ((size_t)256 * sk_L(256) / 4 + 32) * 4 + 256 + 64;
    auto kern = count_chunks_kernel<K, MODE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(nwork), dim3(256), smem, stream, a);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_count_k(int k, const CountArgs &a, uint32_t nwork, hipStream_t st)
{
    switch (k) {
#define MG_CASE(KK) case KK: return launch_count_one<KK, MODE>(a, nwork, st);
        MG_CASE(SK_K0 + 1) MG_CASE(SK_K0 + 2) MG_CASE(SK_K0 + 3) MG_CASE(SK_K0 + 4)
        MG_CASE(SK_K0 + 5) MG_CASE(SK_K0 + 6) MG_CASE(SK_K0 + 7) MG_CASE(SK_K0 + 8)
#undef MG_CASE
    }
    return hipErrorInvalidValue;
}

hipError_t SK_PARTFN(launch_count_part)(int k, int mode, const CountArgs &a, uint32_t nwork, hipStream_t stream)
{
    if (mode == 0) return launch_count_k<0>(k, a, nwork, stream);
    if (mode == 1) return launch_count_k<1>(k, a, nwork, stream);
    return launch_count_k<2>(k, a, nwork, stream);
}

#if SK_PART == 0
hipError_t launch_count_part1(int, int, const CountArgs &, uint32_t, hipStream_t);
hipError_t launch_count_part2(int, int, const CountArgs &, uint32_t, hipStream_t);
hipError_t launch_count_part3(int, int, const CountArgs &, uint32_t, hipStream_t);

hipError_t launch_count_chunks(int k, int mode, const CountArgs &a, uint32_t nwork, hipStream_t stream)
{
    if (nwork == 0) return hipSuccess;
    switch ((k - 1) / 8) {
        case 0: return launch_count_part0(k, mode, a, nwork, stream);
        case 1: return launch_count_part1(k, mode, a, nwork, stream);
        case 2: return launch_count_part2(k, mode, a, nwork, stream);
        case 3: return launch_count_part3(k, mode, a, nwork, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_count_tstar(const uint32_t *nhash, uint32_t *counts, const unsigned long long *firstpos,
                              unsigned long long *tstar, uint32_t *need_fix, uint32_t nsketch, uint32_t s,
                              hipStream_t stream)
{
    if (nsketch == 0) return hipSuccess;
    hipLaunchKernelGGL(count_tstar_kernel, dim3((nsketch + 255) / 256), dim3(256), 0, stream, nhash, counts, firstpos,
                       tstar, need_fix, nsketch, s);
    return hipGetLastError();
}

hipError_t launch_merge_chunks(int nt, const MergeArgs &a, uint32_t nwork, hipStream_t stream)
{
    const size_t smem = (size_t)a.cap * 8 + ((size_t)nt / 64 + 2) * 4 + sizeof(SelState) + 16;
    if (nt == 256) {
        auto kern = merge_chunks_kernel<256>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(nwork), dim3(256), smem, stream, a);
    } else {
        auto kern = merge_chunks_kernel<1024>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(nwork), dim3(1024), smem, stream, a);
    }
    return hipGetLastError();
}

#endif  // SK_PART == 0

}  // namespace mg
