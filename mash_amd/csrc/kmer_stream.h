// kmer_stream.h — tile streaming shared by the kernels that visit EVERY valid k-mer of a chunk
// (multiplicity pass, screen probe): global --16B coalesced--> LDS tile --odd-stride
// ds_read_b32--> one lane = sk_L(NT) consecutive k-mer starts (+K-1 warm-up bytes), rolled with
// KmerRoller.  `emit(hash, position)` is called for each valid k-mer; position = byte offset of
// the k-mer start in `bases`.  (sketch_chunks_kernel has its own copy of this loop because it
// interleaves the bottom-s capacity checks.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmer_hash.h"
#include "sketch_internal.h"

namespace mg {

// k-mer starts per lane per tile: dword lane stride 15 / 7 (odd -> conflict-free ds_read_b32)
__host__ __device__ constexpr int sk_L(int nt) { return nt == 256 ? 60 : 28; }
__host__ __device__ constexpr int sk_tile_dw(int nt) { return nt * sk_L(nt) / 4 + 32; }

template <int K, int MODE, int NT, class Emit>
__device__ __forceinline__ void stream_chunk(const uint8_t *bases, const SketchWork &w, uint32_t *tile,
                                             const uint8_t *alpha, bool fold, uint32_t seed, bool use64, Emit emit)
{
    constexpr int SK_L = sk_L(NT);
    constexpr int TILE = NT * SK_L;
    constexpr int TILE_DW = sk_tile_dw(NT);
    constexpr int NBYTES = SK_L + K - 1;
    constexpr int ND = (NBYTES + 3) / 4;
    const int tid = threadIdx.x;
    for (uint64_t t0 = w.begin; t0 < w.end; t0 += TILE) {
        const uint64_t a0 = t0 & ~15ULL;
        const uint32_t shift = (uint32_t)(t0 - a0);
        for (int q = tid; q < TILE_DW / 4; q += NT) {
            const uint64_t o = a0 + (uint64_t)q * 16;
            uint4 x = make_uint4(0, 0, 0, 0);
            if (o + 16 <= w.limit) {
                x = *reinterpret_cast<const uint4 *>(bases + o);
            } else if (o < w.limit) {
                uint32_t d[4] = {0, 0, 0, 0};
                for (int b = 0; b < 16 && o + b < w.limit; b++)
                    d[b >> 2] |= (uint32_t)bases[o + b] << (8 * (b & 3));
                x = make_uint4(d[0], d[1], d[2], d[3]);
            }
            reinterpret_cast<uint4 *>(tile)[q] = x;
        }
        __syncthreads();
        const uint32_t lane_byte0 = shift + (uint32_t)tid * SK_L;
        const uint32_t *lw = tile + (lane_byte0 >> 2);
        const uint32_t bsh = lane_byte0 & 3;
        const uint64_t rem64 = w.end - t0;
        const uint32_t remaining = rem64 > (uint64_t)TILE ? (uint32_t)TILE : (uint32_t)rem64;
        const uint32_t lane_first = (uint32_t)tid * SK_L;
        KmerRoller<K, MODE == 0> r;
        r.reset();
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
                if (MODE == 2) {
                    if (fold) c = fold_upper(c);
                    r.push(c, alpha[c] != 0);
                } else {
                    if (fold) c &= 0xDFu;
                    uint32_t code, comp;
                    const bool valid = dna_classify(c, code, comp);
                    r.push(c, valid, code, comp);
                }
                if (pos >= K - 1 && pos < NBYTES) {
                    const uint32_t start = (uint32_t)(pos - (K - 1));
                    const uint64_t h = r.hash(seed, use64);
                    if (r.kmer_valid() && (lane_first + start < remaining)) emit(h, t0 + lane_first + start);
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace mg
