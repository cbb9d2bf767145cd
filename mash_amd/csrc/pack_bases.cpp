// pack_bases.cpp -- host side of the packed nucleotide transport (include/mashgpu.h: mg_pack_bases; the device side is
// ingest.hip).  Plain C++, no device code: what a parse thread runs over the bytes kseq hands it (kseq.h:171-208) before
// they cross PCIe.  Eight bases per step as one 64-bit word: case fold, four byte-wise comparisons, the 2-bit codes
// gathered by shifts; 32 bases per step where the host has AVX2 + BMI2 (MASHGPU_PACK_PORTABLE=1 in the environment of the
// process keeps the portable steps: the tests run both).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>                                      // (outside every namespace: the header's own declarations belong to the global one)
#endif

#include "../../include/mashgpu.h"

namespace {

constexpr uint64_t kOnes = 0x0101010101010101ull, kLow7 = 0x7F7F7F7F7F7F7F7Full;

// 0x80 in every byte of x that equals c, 0 elsewhere (exact: no borrow between bytes)
inline uint64_t bytes_equal(uint64_t x, uint8_t c)
{
    const uint64_t z = x ^ (kOnes * c);
    return ~(((z & kLow7) + kLow7) | z | kLow7);
}

struct Packed8 { uint16_t codes; uint8_t invalid; };

inline Packed8 pack8(uint64_t x, bool fold)
{
    const uint64_t u = fold ? (x & 0xDFDFDFDFDFDFDFDFull) : x;      // 'a'..'z' -> 'A'..'Z'; nothing else becomes A, C, G or T
    const uint64_t ok = bytes_equal(u, 'A') | bytes_equal(u, 'C') | bytes_equal(u, 'G') | bytes_equal(u, 'T');
    // code = (ASCII >> 1) & 3 in every byte, then byte i's two bits to bits 2i of the result
    uint64_t t = (u >> 1) & (kOnes * 3u);
    t = (t | (t >> 6)) & 0x000F000F000F000Full;
    t = (t | (t >> 12)) & 0x000000FF000000FFull;
    t = (t | (t >> 24)) & 0xFFFFull;
    const uint64_t bad = ~ok & (kOnes * 0x80u);
    const uint8_t inv = (uint8_t)(((bad >> 7) * 0x0102040810204080ull) >> 56);
    return {(uint16_t)t, inv};
}

#if defined(__x86_64__)
// 32 bases per step on hosts with AVX2 + BMI2 (chosen at run time; the 64-bit steps above remain the portable form and
// do the tail): byte-wise compares for the four letters, movemask for the invalid bits, pext for the code bits
__attribute__((target("avx2,bmi2"))) uint64_t pack_avx2(const uint8_t *ascii, uint64_t groups32, bool fold, uint8_t *packed,
                                                         uint8_t *invalid_mask)
{
    const __m256i fold_mask = _mm256_set1_epi8(fold ? (char)0xDF : (char)0xFF);
    const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
    uint64_t ninv = 0;
    for (uint64_t g = 0; g < groups32; g++) {
        const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(ascii + 32u * g));
        const __m256i u = _mm256_and_si256(x, fold_mask);
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)),
                                           _mm256_or_si256(_mm256_cmpeq_epi8(u, cG), _mm256_cmpeq_epi8(u, cT)));
        const uint32_t inv = ~(uint32_t)_mm256_movemask_epi8(ok);
        memcpy(invalid_mask + 4u * g, &inv, 4);
        ninv += (uint64_t)__builtin_popcount(inv);
        alignas(32) uint64_t w[4];
        _mm256_store_si256(reinterpret_cast<__m256i *>(w), u);
        for (int q = 0; q < 4; q++) {                       // bits 1..2 of every byte: the code
            const uint16_t c = (uint16_t)_pext_u64(w[q], 0x0606060606060606ull);
            memcpy(packed + 8u * g + 2u * q, &c, 2);
        }
    }
    return ninv;
}
#endif

}  // namespace

extern "C" {

uint64_t mg_packed_bytes(uint64_t nbases) { return (nbases + 3u) / 4u; }
uint64_t mg_packed_mask_bytes(uint64_t nbases) { return (nbases + 7u) / 8u; }

int mg_pack_bases(const uint8_t *ascii, uint64_t nbases, int preserve_case, uint8_t *packed, uint8_t *invalid_mask,
                  uint64_t *ninvalid_out)
{
    if ((!ascii && nbases) || (!packed && nbases) || (!invalid_mask && nbases)) return MG_ERR_INVALID;
    const bool fold = !preserve_case;
    uint64_t ninv = 0;
    const uint64_t full = nbases / 8u;
    uint64_t g0 = 0;
#if defined(__x86_64__)
    static const bool wide = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && !getenv("MASHGPU_PACK_PORTABLE");
    if (wide) {
        ninv += pack_avx2(ascii, nbases / 32u, fold, packed, invalid_mask);
        g0 = (nbases / 32u) * 4u;
    }
#endif
    for (uint64_t g = g0; g < full; g++) {
        uint64_t x;
        memcpy(&x, ascii + 8u * g, 8);                      // (little-endian hosts: byte i of the input is byte i of the word)
        const Packed8 r = pack8(x, fold);
        packed[2u * g] = (uint8_t)(r.codes & 0xFFu);
        packed[2u * g + 1u] = (uint8_t)(r.codes >> 8);
        invalid_mask[g] = r.invalid;
        ninv += (uint64_t)__builtin_popcount(r.invalid);
    }
    const uint64_t rest = nbases - 8u * full;
    if (rest) {
        uint8_t tail[8] = {'A', 'A', 'A', 'A', 'A', 'A', 'A', 'A'};
        memcpy(tail, ascii + 8u * full, rest);
        uint64_t x;
        memcpy(&x, tail, 8);
        Packed8 r = pack8(x, fold);
        r.invalid &= (uint8_t)((1u << rest) - 1u);
        packed[2u * full] = (uint8_t)(r.codes & 0xFFu);
        if (rest > 4) packed[2u * full + 1u] = (uint8_t)(r.codes >> 8);
        invalid_mask[full] = r.invalid;
        ninv += (uint64_t)__builtin_popcount(r.invalid);
    }
    if (ninvalid_out) *ninvalid_out = ninv;
    return MG_OK;
}

}  // extern "C"
