// kmer_hash.h — per-lane k-mer state for the sketch kernel (host+device).
//
// One lane walks a run of consecutive bases.  It keeps, in registers:
//   * the ASCII text of the current forward k-mer as little-endian 64-bit words
//     (byte i of the k-mer = byte i of the 32-byte window) — this is exactly the
//     buffer MurmurHash3_x64_128 reads in the reference (hash.cpp:10-38 hashes the
//     k ASCII bytes, MurmurHash3.cpp:60-63 loads blocks little-endian);
//   * for canonical DNA, the ASCII text of the reverse complement, rolled the
//     other way (prepend complement of the incoming base, drop the last byte),
//     and both strands 2-bit packed with the first base most significant, so the
//     reference's `memcmp(fwd, rev, k) <= 0` (Sketch.cpp:569-571; ASCII order
//     A<C<G<T equals code order 0<1<2<3) is one 64-bit compare;
//   * the count of consecutive in-alphabet bytes ending at the current byte: a
//     k-mer is valid iff all its k bytes are in the alphabet (Sketch.cpp:544-567).
// Everything is templated on K so the window words, shifts and the murmur
// block/tail structure are static.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MG_HD __host__ __device__ __forceinline__
#else
#define MG_HD inline
#endif

namespace mg {

MG_HD uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

MG_HD uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

// MurmurHash3_x64_128 (MurmurHash3.cpp:255-335) of the K bytes held in w[0..3]
// (little-endian words, bytes >= K are zero).  Returns h1 (the only half getHash
// consumes, hash.cpp:28-35); h2 still has to be carried (h1 += h2 at the end).
template <int K>
MG_HD uint64_t murmur3_h1(const uint64_t w[4], uint32_t seed)
{
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = seed, h2 = seed;
    constexpr int NB = K / 16, REM = K & 15;
#pragma unroll
    for (int i = 0; i < NB; i++) {
        uint64_t k1 = w[2 * i], k2 = w[2 * i + 1];
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    if (REM > 8) {
        uint64_t k2 = w[(2 * NB + 1) & 3];
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
    }
    if (REM > 0) {
        uint64_t k1 = w[(2 * NB) & 3];
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint64_t)K; h2 ^= (uint64_t)K;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2;
    return h1;
}

// Case folding exactly as Sketch.cpp:524-530: only a..z are changed.
MG_HD uint32_t fold_upper(uint32_t c) { return c - (((c - 97u) < 26u) ? 32u : 0u); }

template <int K, bool CANON>
struct KmerRoller {
    static constexpr int NW = (K + 7) / 8;                  // window words in use
    static constexpr int TOPB = (K - 1) & 7;                // byte lane of k-mer byte K-1 in its word
    uint64_t fw[4];      // forward ASCII window
    uint64_t rw[4];      // reverse-complement ASCII window (CANON only)
    uint64_t f2, r2;     // 2-bit packed strands (CANON only), first base most significant
    uint32_t run;        // consecutive valid bytes ending here (saturates at K)

    MG_HD void reset()
    {
#pragma unroll
        for (int i = 0; i < 4; i++) { fw[i] = 0; rw[i] = 0; }
        f2 = r2 = 0; run = 0;
    }

    // Push one byte: `c` is the byte to hash (already case-folded), `valid` whether it
    // is in the alphabet.  For CANON, `code` is its 2-bit code (A0 C1 G2 T3) and
    // `comp` the ASCII of its complement.
    MG_HD void push(uint32_t c, bool valid, uint32_t code = 0, uint32_t comp = 0)
    {
        // forward: drop byte 0, append c as byte K-1
#pragma unroll
        for (int i = 0; i < NW - 1; i++) fw[i] = (fw[i] >> 8) | (fw[i + 1] << 56);
        fw[NW - 1] = (fw[NW - 1] >> 8) | ((uint64_t)c << (8 * TOPB));
        if (CANON) {
            // reverse complement: prepend comp as byte 0, drop byte K
#pragma unroll
            for (int i = NW - 1; i > 0; i--) rw[i] = (rw[i] << 8) | (rw[i - 1] >> 56);
            rw[0] = (rw[0] << 8) | (uint64_t)comp;
            if (TOPB != 7) rw[NW - 1] &= (~0ULL) >> (8 * (7 - TOPB));
            constexpr uint64_t M2 = (K == 32) ? ~0ULL : ((1ULL << (2 * (K & 31))) - 1ULL);
            f2 = ((f2 << 2) | (uint64_t)code) & M2;
            r2 = (r2 >> 2) | ((uint64_t)(3u - code) << (2 * (K - 1)));
        }
        run = valid ? (run < (uint32_t)K ? run + 1 : (uint32_t)K) : 0u;
    }

    MG_HD bool kmer_valid() const { return run >= (uint32_t)K; }

    // Hash of the k-mer ending at the last pushed byte (getHash: h1, or its low 32 bits).
    MG_HD uint64_t hash(uint32_t seed, bool use64) const
    {
        uint64_t h;
        if (CANON) {
            const bool use_f = f2 <= r2;                    // memcmp(fwd, rev, k) <= 0
            uint64_t w[4];
#pragma unroll
            for (int i = 0; i < 4; i++) w[i] = use_f ? fw[i] : rw[i];
            h = murmur3_h1<K>(w, seed);
        } else {
            h = murmur3_h1<K>(fw, seed);
        }
        return use64 ? h : (h & 0xFFFFFFFFULL);
    }
};

// DNA base classification without a table: idx = upper(c) - 'A'; members A,C,G,T
// are bits 0,2,6,19 of 0x80045.  code: A0 C1 G2 T3 from ASCII bits ((c>>1)&3 gives
// A0 C1 G3 T2; x ^ (x>>1) swaps the last two).  comp ASCII = "TGCA"[code].
MG_HD bool dna_classify(uint32_t c_folded, uint32_t &code, uint32_t &comp)
{
    const uint32_t idx = c_folded - 65u;
    const bool valid = (idx < 32u) && ((0x80045u >> idx) & 1u);
    uint32_t x = (c_folded >> 1) & 3u;
    code = x ^ (x >> 1);
    comp = (0x41434754u >> (8 * code)) & 0xFFu;    // bytes (LSB first): 'T','G','C','A'
    return valid;
}

}  // namespace mg
