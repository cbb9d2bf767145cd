// pack_bases.cpp -- host side of the packed nucleotide transport (include/mashgpu.h: mg_pack_bases; the device side is
// ingest.hip).  Plain C++, no device code: what a parse thread runs over the bytes kseq hands it (kseq.h:171-208) before
// they cross PCIe.  Eight bases per step as one 64-bit word: case fold, four byte-wise comparisons, the 2-bit codes
// gathered by shifts.
#include <cstdint>
#include <cstring>

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
    for (uint64_t g = 0; g < full; g++) {
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
