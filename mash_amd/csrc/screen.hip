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
