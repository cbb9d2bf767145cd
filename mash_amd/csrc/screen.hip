// screen.hip — gfx950 kernels for the containment path of `mash screen`
// (CommandScreen.cpp:99-114 hash table build, :484-599 hashSequence, :338-355 shared counts).
//
//   build : every distinct hash of the query-sketch table -> open-addressing table in HBM
//           (u64 keys, linear probing, load <= 0.5) with one u32 observation counter per slot;
//   probe : stream the mixture (reads / contigs), hash every valid canonical k-mer exactly as
//           the sketch kernel does, and atomically count the ones present in the table
//           (hashCounts[key]++, CommandScreen.cpp:571-575);
//   gather: per (sketch, index) the observation count of that hash — `shared` and the
//           multiplicity medians follow on the host.
// The mixture's own bottom-s sketch (for estimateSetSize, CommandScreen.cpp:288-322) is
// produced by the ordinary sketch kernels.  All integer work; probe is ALU bound like sketching,
// the table probes are random 8-byte HBM/L2 reads (one 64-B line per k-mer at most).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmer_stream.h"
#include "screen_internal.h"

namespace mg {

constexpr unsigned long long SCR_EMPTY = 0xFFFFFFFFFFFFFFFFULL;

__device__ __forceinline__ uint64_t scr_slot(uint64_t key, uint64_t mask)
{
    uint64_t x = key * 0x9E3779B97F4A7C15ULL;              // keys are murmur outputs: one multiply suffices
    return (x >> 20) & mask;
}

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

__device__ __forceinline__ bool scr_find(const unsigned long long *keys, uint64_t mask, uint64_t key, uint64_t *slot_out)
{
    uint64_t slot = scr_slot(key, mask);
    for (;;) {
        const unsigned long long k = keys[slot];
        if (k == key) { *slot_out = slot; return true; }
        if (k == SCR_EMPTY) return false;
        slot = (slot + 1) & mask;
    }
}

template <int K, int MODE>
__global__ __launch_bounds__(256) void screen_probe_kernel(ScreenProbeArgs a)
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
    const unsigned long long *keys = a.keys;
    uint32_t *obs = a.obs;
    const uint64_t mask = a.mask;
    stream_chunk<K, MODE, NT>(a.bases, w, tile, alpha, a.fold_case != 0, a.seed, a.use64 != 0,
                              [&](uint64_t h, uint64_t) {
        uint64_t slot;
        if (scr_find(keys, mask, h, &slot)) atomicAdd(&obs[slot], 1u);
    });
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

template <int K, int MODE>
static hipError_t launch_probe_one(const ScreenProbeArgs &a, uint32_t nwork, hipStream_t stream)
{
    const size_t smem = (size_t)sk_tile_dw(256) * 4 + 256 + 64;
    hipLaunchKernelGGL((screen_probe_kernel<K, MODE>), dim3(nwork), dim3(256), smem, stream, a);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_probe_k(int k, const ScreenProbeArgs &a, uint32_t nwork, hipStream_t st)
{
    switch (k) {
#define MG_CASE(KK) case KK: return launch_probe_one<KK, MODE>(a, nwork, st);
        MG_CASE(1) MG_CASE(2) MG_CASE(3) MG_CASE(4) MG_CASE(5) MG_CASE(6) MG_CASE(7) MG_CASE(8)
        MG_CASE(9) MG_CASE(10) MG_CASE(11) MG_CASE(12) MG_CASE(13) MG_CASE(14) MG_CASE(15) MG_CASE(16)
        MG_CASE(17) MG_CASE(18) MG_CASE(19) MG_CASE(20) MG_CASE(21) MG_CASE(22) MG_CASE(23) MG_CASE(24)
        MG_CASE(25) MG_CASE(26) MG_CASE(27) MG_CASE(28) MG_CASE(29) MG_CASE(30) MG_CASE(31) MG_CASE(32)
#undef MG_CASE
    }
    return hipErrorInvalidValue;
}

hipError_t launch_screen_probe(int k, int mode, const ScreenProbeArgs &a, uint32_t nwork, hipStream_t stream)
{
    if (nwork == 0) return hipSuccess;
    if (mode == 0) return launch_probe_k<0>(k, a, nwork, stream);
    if (mode == 1) return launch_probe_k<1>(k, a, nwork, stream);
    return launch_probe_k<2>(k, a, nwork, stream);
}

}  // namespace mg
