// screen.hip — gfx950 kernels for the containment path of `mash screen`
// (CommandScreen.cpp:99-114 hash table build, :484-599 hashSequence, :338-355 shared counts).
//
//   build : every distinct hash of the query-sketch table -> open-addressing table in HBM
//           (u64 keys, linear probing, load <= 0.5) with one u32 observation counter per slot;
//   probe : fused into sketch_chunks_kernel (sketch.hip, SketchArgs::probe_*): while the
//           mixture is sketched for its own bottom-s, every valid canonical k-mer hash that is
//           not above the largest key is looked up and counted (hashCounts[key]++,
//           CommandScreen.cpp:571-575) -- one pass over the mixture, one hash per k-mer;
//   gather: per (sketch, index) the observation count of that hash — `shared` and the
//           multiplicity medians follow on the host.
// The mixture's own bottom-s sketch (for estimateSetSize, CommandScreen.cpp:288-322) comes out
// of the same pass.  All integer work; the pass is ALU bound like sketching, the table probes
// are random 8-byte HBM/L2 reads (at most one 64-B line per k-mer that passes the key bound).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "screen_internal.h"

namespace mg {

__global__ void screen_build_kernel(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                                    unsigned long long *keys, uint64_t mask)
{
    const uint64_t total = n * s;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const uint64_t i = e / s, j = e - i * s;
        uint32_t k = nhash[i];
        if (k > s) k = (uint32_t)s;
        if (j >= k) continue;
        const unsigned long long key = hashes[e];
        if (key == SCR_EMPTY) continue;                    // reserved as the empty marker
        uint64_t slot = scr_slot(key, mask);
        for (;;) {
            const unsigned long long old = atomicCAS(&keys[slot], SCR_EMPTY, key);
            if (old == SCR_EMPTY || old == key) break;
            slot = (slot + 1) & mask;
        }
    }
}

__global__ void screen_gather_kernel(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                                     const unsigned long long *keys, const uint32_t *obs, uint64_t mask,
                                     uint32_t *counts_out)
{
    const uint64_t total = n * s;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const uint64_t i = e / s, j = e - i * s;
        uint32_t k = nhash[i];
        if (k > s) k = (uint32_t)s;
        uint32_t c = 0;
        if (j < k) {
            uint64_t slot;
            if (scr_find(keys, mask, hashes[e], &slot)) c = obs[slot];
        }
        counts_out[e] = c;
    }
}

// 6-frame translation of a nucleotide batch (CommandScreen.cpp:516-531 + translate/aaFromCodon,
// :617-809) for amino-acid query sketches.  The reference translates each `*`-joined chunk of
// reads as one string in frames 0..2 of the chunk and of its reverse complement; a codon that
// holds anything but upper-case ACGT (after the optional case fold) becomes '*', which no k-mer
// may contain -- so codons never bridge records and the set of k-mers does not depend on how
// reads were chunked.  Output: six segments of `seg` bytes, amino acids then MG_RECORD_SEP
// padding (not in the protein alphabet), ready for the table-alphabet sketch/probe pass.
__constant__ char kCodonTable[65] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";

__device__ __forceinline__ int nt_digit(uint32_t c, bool fold)
{
    if (fold) c &= 0xDFu;                                  // only a/c/g/t fold onto A/C/G/T
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return -1;
    }
}

__global__ void translate6_kernel(const uint8_t *in, uint64_t n, uint8_t *out, uint64_t seg, uint32_t fold)
{
    const uint64_t total = 6 * seg;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const uint32_t fr = (uint32_t)(idx / seg);
        const uint64_t j = idx - (uint64_t)fr * seg;
        const uint32_t f = fr % 3;
        const bool rev = fr >= 3;
        const uint64_t len = n >= f ? (n - f) / 3 : 0;        // lenTrans = (l - frame) / 3
        uint8_t aa = 0x0A;
        if (j < len) {
            int d0, d1, d2;
            if (!rev) {
                const uint64_t p = f + 3 * j;
                d0 = nt_digit(in[p], fold != 0); d1 = nt_digit(in[p + 1], fold != 0); d2 = nt_digit(in[p + 2], fold != 0);
            } else {                                           // reverse complement read backwards
                const uint64_t p = n - 1 - f - 3 * j;
                d0 = nt_digit(in[p], fold != 0); d1 = nt_digit(in[p - 1], fold != 0); d2 = nt_digit(in[p - 2], fold != 0);
                if (d0 >= 0) d0 = 3 - d0;
                if (d1 >= 0) d1 = 3 - d1;
                if (d2 >= 0) d2 = 3 - d2;
            }
            aa = (d0 | d1 | d2) < 0 ? (uint8_t)'*' : (uint8_t)kCodonTable[d0 * 16 + d1 * 4 + d2];
        }
        out[idx] = aa;
    }
}

hipError_t launch_translate6(const uint8_t *in, uint64_t n, uint8_t *out, uint64_t seg, bool fold, hipStream_t stream)
{
    uint64_t blocks = (6 * seg + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(translate6_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, in, n, out, seg, fold ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_screen_build(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                               unsigned long long *keys, uint64_t mask, hipStream_t stream)
{
    if (n * s == 0) return hipSuccess;
    uint64_t blocks = (n * s + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(screen_build_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, hashes, nhash, n, s, keys, mask);
    return hipGetLastError();
}

hipError_t launch_screen_gather(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                                const unsigned long long *keys, const uint32_t *obs, uint64_t mask,
                                uint32_t *counts_out, hipStream_t stream)
{
    if (n * s == 0) return hipSuccess;
    uint64_t blocks = (n * s + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(screen_gather_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, hashes, nhash, n, s, keys,
                       obs, mask, counts_out);
    return hipGetLastError();
}

}  // namespace mg
