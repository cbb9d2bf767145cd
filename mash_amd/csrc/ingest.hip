// Packed nucleotide input (BASELINE.json north_star: "coalesced HBM reads of packed bases").
//
// The transport format of mg_sketch_host_packed / mg_sketch_dev_packed (include/mashgpu.h): two bits per base,
// four bases per byte, base i in bits 2(i % 4) .. of byte i / 4, code = (ASCII >> 1) & 3 -- A 0, C 1, T 2, G 3 --
// and one bit per base (bit i % 8 of byte i / 8) set where the input held anything else: N, IUPAC codes, the
// separator between two records, lower case under preserve_case.  That is everything addMinHashes looks at for
// the ACGT alphabet (Sketch.cpp:512-583: upper-cased unless preserveCase, a k-mer over a character outside the
// alphabet is skipped, the hash is taken over the k-mer's characters), at 0.375 B per base over PCIe instead of 1.
//
// MurmurHash3 runs over the k-mer's ASCII characters, so the sketch kernel's register windows are ASCII either way;
// this kernel turns a packed range back into the bytes the ASCII path would have been handed -- 'A' 'C' 'G' 'T', and
// 'N' where the mask is set -- one coalesced 16-byte store per lane (1.375 B of HBM traffic per base, 2 % of the
// sketch kernel's time), and sketch_chunks_kernel runs unchanged: the packed path cannot disagree with the ASCII one
// about anything but these bytes (tests/test_gpu_parity.py::test_sketch_packed_*).
#include <hip/hip_runtime.h>
#include <cstdint>

#include "sketch_internal.h"

namespace mg {

typedef uint32_t in_u32x4 __attribute__((ext_vector_type(4)));

// out[j] for j in [0, n): base number skip + j of `packed` (skip < 16), validity bit mskip + j of `mask` (mskip < 32;
// mask == nullptr: every base valid); both arrays 4-byte aligned.  One lane = 16 bases: one dword of codes (two when
// skip != 0), one or two dwords of the mask.  The buffers behind packed / mask are readable 8 bytes past their last
// used byte (callers pad); out holds a multiple of 16 bytes: positions >= n are written as 'N'.
__global__ __launch_bounds__(256) void unpack_bases_kernel(const uint32_t *packed, const uint32_t *mask, uint32_t skip, uint32_t mskip,
                                                           uint64_t n, uint8_t *out)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t j0 = t * 16u;
    if (j0 >= n) return;
    uint64_t w = packed[t];                                 // bases skip + j0 .. start in dword (skip + j0) / 16 = t
    if (skip) w |= (uint64_t)packed[t + 1] << 32;
    const uint32_t bits = (uint32_t)(w >> (2u * skip));
    uint32_t inv = 0;
    if (mask) {
        const uint64_t gm = mskip + j0;
        const uint32_t sh = (uint32_t)(gm & 31u);
        uint64_t m = mask[gm >> 5];
        if (sh > 16u) m |= (uint64_t)mask[(gm >> 5) + 1u] << 32;
        inv = (uint32_t)(m >> sh) & 0xFFFFu;
    }
    const uint64_t left = n - j0;
    if (left < 16u) inv |= 0xFFFFu << (uint32_t)left;
    in_u32x4 v;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t word = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int i = q * 4 + b;
            const uint32_t code = (bits >> (2 * i)) & 3u;
            uint32_t ch = (0x47544341u >> (8u * code)) & 0xFFu;       // "ACTG"[code]
            if ((inv >> i) & 1u) ch = 'N';
            word |= ch << (8 * b);
        }
        v[q] = word;
    }
    *reinterpret_cast<in_u32x4 *>(out + j0) = v;
}

hipError_t launch_unpack_bases(const uint8_t *packed, const uint8_t *mask, uint32_t skip, uint32_t mskip, uint64_t n, uint8_t *out,
                               hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint64_t lanes = (n + 15u) / 16u;
    const uint64_t blocks = (lanes + 255u) / 256u;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (skip >= 16u || mskip >= 32u || ((uintptr_t)packed & 3u) || ((uintptr_t)mask & 3u) || ((uintptr_t)out & 15u)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(unpack_bases_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, reinterpret_cast<const uint32_t *>(packed),
                       reinterpret_cast<const uint32_t *>(mask), skip, mskip, n, out);
    return hipGetLastError();
}

}  // namespace mg
