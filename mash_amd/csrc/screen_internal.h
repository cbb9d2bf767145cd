// screen_internal.h — launch interface between mashgpu.cpp and screen.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sketch_internal.h"

namespace mg {

// open-addressing table shared by screen.hip (build, gather) and the probe fused into the
// sketch kernel: u64 keys, linear probing, ~0 = empty, load <= 0.5
constexpr unsigned long long SCR_EMPTY = 0xFFFFFFFFFFFFFFFFULL;

__device__ __forceinline__ uint64_t scr_slot(uint64_t key, uint64_t mask)
{
    uint64_t x = key * 0x9E3779B97F4A7C15ULL;              // keys are murmur outputs: one multiply suffices
    return (x >> 20) & mask;
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

// six translated segments of `seg` bytes each (seg > n/3), see translate6_kernel
hipError_t launch_translate6(const uint8_t *in, uint64_t n, uint8_t *out, uint64_t seg, bool fold, hipStream_t stream);
hipError_t launch_screen_build(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                               unsigned long long *keys, uint64_t mask, hipStream_t stream);
hipError_t launch_screen_gather(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                                const unsigned long long *keys, const uint32_t *obs, uint64_t mask,
                                uint32_t *counts_out, hipStream_t stream);

}  // namespace mg
