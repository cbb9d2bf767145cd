// kmer_hash.h — per-lane k-mer state for the sketch kernel (host+device).
//
// One lane walks a run of consecutive bases.  It keeps, in registers:
//   * the ASCII text of the current forward k-mer as little-endian 32-bit words
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
// block/tail structure are static.  The arithmetic is written on 32-bit halves with
// explicit byte/bit aligns because gfx950 integer VALU issue is the kernel's limiter
// (~4.3 cycles per wave64 instruction, tools/ubench_valu.hip): a 64-bit rotate is two
// v_alignbit, a window roll one v_alignbyte per dword, a 64x64 multiply one
// v_mad_u64_u32 + two v_mul_lo_u32 + one v_add3.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MG_HD __host__ __device__ __forceinline__
#else
#define MG_HD inline
#endif

namespace mg {

// ({hi,lo} >> (8*n)) & 0xFFFFFFFF, n in 0..3
MG_HD uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, n);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * n));
#endif
}

// ({hi,lo} >> n) & 0xFFFFFFFF, n in 0..31
MG_HD uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, n);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> n);
#endif
}

struct u64x2 {               // a 64-bit value as two 32-bit halves
    uint32_t lo, hi;
};

MG_HD u64x2 make64(uint64_t v) { return {(uint32_t)v, (uint32_t)(v >> 32)}; }
MG_HD uint64_t to64(u64x2 v) { return ((uint64_t)v.hi << 32) | v.lo; }

template <int R>
MG_HD u64x2 rotl64h(u64x2 x)
{
    static_assert(R > 0 && R < 64 && R != 32, "rotation");
    if (R < 32) return {alignbit(x.lo, x.hi, 32 - R), alignbit(x.hi, x.lo, 32 - R)};
    return {alignbit(x.hi, x.lo, 64 - R), alignbit(x.lo, x.hi, 64 - R)};
}

MG_HD u64x2 mul64c(u64x2 x, uint64_t c)
{
    const uint32_t cl = (uint32_t)c, ch = (uint32_t)(c >> 32);
    const uint64_t p = (uint64_t)x.lo * cl;                       // v_mad_u64_u32
    return {(uint32_t)p, (uint32_t)(p >> 32) + x.lo * ch + x.hi * cl};
}

// 64-bit add on halves with an explicit carry (v_add_co / v_addc): no register-pair shuffling
MG_HD u64x2 add64(u64x2 a, u64x2 b)
{
    const uint32_t lo = a.lo + b.lo;
    return {lo, a.hi + b.hi + (uint32_t)(lo < a.lo)};
}

// h * 5 + c (c < 2^32): one 32x32->64 multiply-add for the low half, shift-add for the high half
MG_HD u64x2 mul5add(u64x2 h, uint32_t c)
{
    const uint64_t p = (uint64_t)h.lo * 5u + c;
    return {(uint32_t)p, (uint32_t)(p >> 32) + h.hi * 5u};
}
MG_HD u64x2 xor64(u64x2 a, u64x2 b) { return {a.lo ^ b.lo, a.hi ^ b.hi}; }

MG_HD u64x2 fmix64h(u64x2 k)
{
    k.lo ^= k.hi >> 1;                                            // k ^= k >> 33
    k = mul64c(k, 0xff51afd7ed558ccdULL);
    k.lo ^= k.hi >> 1;
    k = mul64c(k, 0xc4ceb9fe1a85ec53ULL);
    k.lo ^= k.hi >> 1;
    return k;
}

// MurmurHash3_x64_128 (MurmurHash3.cpp:255-335) of the K bytes held in w[0..7]
// (little-endian dwords, bytes >= K are zero).  Returns h1 (the only half getHash
// consumes, hash.cpp:28-35); h2 still has to be carried (h1 += h2 at the end).
template <int K>
MG_HD uint64_t murmur3_h1(const uint32_t w[8], uint32_t seed)
{
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    u64x2 h1 = {seed, 0}, h2 = {seed, 0};
    constexpr int NB = K / 16, REM = K & 15;
#pragma unroll
    for (int i = 0; i < NB; i++) {
        u64x2 k1 = {w[4 * i], w[4 * i + 1]}, k2 = {w[4 * i + 2], w[4 * i + 3]};
        k1 = mul64c(k1, c1); k1 = rotl64h<31>(k1); k1 = mul64c(k1, c2); h1 = xor64(h1, k1);
        h1 = rotl64h<27>(h1); h1 = add64(h1, h2);
        h1 = mul5add(h1, 0x52dce729u);
        k2 = mul64c(k2, c2); k2 = rotl64h<33>(k2); k2 = mul64c(k2, c1); h2 = xor64(h2, k2);
        h2 = rotl64h<31>(h2); h2 = add64(h2, h1);
        h2 = mul5add(h2, 0x38495ab5u);
    }
    if (REM > 8) {
        u64x2 k2 = {w[(4 * NB + 2) & 7], w[(4 * NB + 3) & 7]};
        k2 = mul64c(k2, c2); k2 = rotl64h<33>(k2); k2 = mul64c(k2, c1); h2 = xor64(h2, k2);
    }
    if (REM > 0) {
        u64x2 k1 = {w[(4 * NB) & 7], w[(4 * NB + 1) & 7]};
        k1 = mul64c(k1, c1); k1 = rotl64h<31>(k1); k1 = mul64c(k1, c2); h1 = xor64(h1, k1);
    }
    h1.lo ^= (uint32_t)K; h2.lo ^= (uint32_t)K;
    h1 = add64(h1, h2); h2 = add64(h2, h1);
    h1 = fmix64h(h1); h2 = fmix64h(h2);
    h1 = add64(h1, h2);
    return to64(h1);
}

// Case folding exactly as Sketch.cpp:524-530: only a..z are changed.
MG_HD uint32_t fold_upper(uint32_t c) { return c - (((c - 97u) < 26u) ? 32u : 0u); }

template <int K, bool CANON>
struct KmerRoller {
    static constexpr int NW = (K + 3) / 4;                  // window dwords in use
    static constexpr int TOPB = (K - 1) & 3;                // byte lane of k-mer byte K-1 in its dword
    uint32_t fw[8];      // forward ASCII window
    uint32_t rw[8];      // reverse-complement ASCII window (CANON only)
    u64x2 f2, r2;        // 2-bit packed strands (CANON only), first base most significant
    uint32_t run;        // consecutive valid bytes ending here

    MG_HD void reset()
    {
#pragma unroll
        for (int i = 0; i < 8; i++) { fw[i] = 0; rw[i] = 0; }
        f2 = {0, 0}; r2 = {0, 0}; run = 0;
    }

    // Push one byte: `c` is the byte to hash (already case-folded), `valid` whether it
    // is in the alphabet.  For CANON, `code` is its 2-bit code (A0 C1 G2 T3) and
    // `comp` the ASCII of its complement.
    MG_HD void push(uint32_t c, bool valid, uint32_t code = 0, uint32_t comp = 0)
    {
        // forward: drop byte 0, append c as byte K-1
#pragma unroll
        for (int i = 0; i < NW - 1; i++) fw[i] = alignbyte(fw[i + 1], fw[i], 1);
        if (TOPB == 0) fw[NW - 1] = c;
        else fw[NW - 1] = (fw[NW - 1] >> 8) | (c << (8 * TOPB));
        if (CANON) {
            // reverse complement: prepend comp as byte 0, drop byte K
#pragma unroll
            for (int i = NW - 1; i > 0; i--) rw[i] = alignbyte(rw[i], rw[i - 1], 3);
            rw[0] = (rw[0] << 8) | comp;
            if (TOPB != 3) rw[NW - 1] &= 0xFFFFFFFFu >> (8 * (3 - TOPB));
            // f2 = ((f2 << 2) | code) & mask(2K bits) ; r2 = (r2 >> 2) | ((3-code) << 2(K-1))
            constexpr uint64_t M2 = (K == 32) ? ~0ULL : ((1ULL << (2 * (K & 31))) - 1ULL);
            f2.hi = alignbit(f2.hi, f2.lo, 30) & (uint32_t)(M2 >> 32);
            f2.lo = ((f2.lo << 2) | code) & (uint32_t)M2;
            const uint32_t ccode = 3u - code;
            r2.lo = alignbit(r2.hi, r2.lo, 2);
            r2.hi >>= 2;
            constexpr int TOPSH = 2 * (K - 1);
            if (TOPSH >= 32) r2.hi |= ccode << (TOPSH - 32);
            else r2.lo |= ccode << TOPSH;
        }
        run = valid ? run + 1u : 0u;                         // < 2^32 steps between resets
    }

    MG_HD bool kmer_valid() const { return run >= (uint32_t)K; }

    // Hash of the k-mer ending at the last pushed byte (getHash: h1, or its low 32 bits).
    MG_HD uint64_t hash(uint32_t seed, bool use64) const
    {
        uint64_t h;
        if (CANON) {
            const bool use_f = to64(f2) <= to64(r2);        // memcmp(fwd, rev, k) <= 0
            uint32_t w[8];
#pragma unroll
            for (int i = 0; i < 8; i++) w[i] = use_f ? fw[i] : rw[i];
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
