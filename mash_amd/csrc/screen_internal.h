// screen_internal.h — launch interface between host_screen.cpp and screen.hip.
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
struct ScreenHit {             // = mg_screen_hit (include/mashgpu.h)
    uint32_t row, count;
    uint64_t hash;
};

// (*distinct += keys inserted: the table's distinct hashes)
hipError_t launch_screen_build(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                               unsigned long long *keys, uint64_t mask, unsigned long long *distinct, hipStream_t stream);
size_t screen_index_temp_bytes(uint64_t slots);
hipError_t launch_screen_index(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s, const unsigned long long *keys,
                               uint64_t mask, uint32_t *slot_end, uint32_t *ent, void *temp, size_t temp_bytes, hipStream_t stream);
hipError_t launch_screen_hits(const uint32_t *touched, uint64_t nt, const unsigned long long *keys, const uint32_t *obs,
                              const uint32_t *slot_end, const uint32_t *ent, ScreenHit *hits, unsigned long long *cursor, uint64_t cap,
                              hipStream_t stream);
size_t screen_sort_temp_bytes(uint64_t n);
hipError_t launch_screen_sort_hits(const ScreenHit *hits, uint64_t n, ScreenHit *out, unsigned long long *k64a, unsigned long long *k64b,
                                   uint32_t *u32a, uint32_t *u32b, uint32_t *u32c, uint32_t *u32d, void *temp, size_t temp_bytes,
                                   hipStream_t stream);
hipError_t launch_screen_reset(const uint32_t *touched, uint64_t nt, uint32_t *obs, hipStream_t stream);
hipError_t launch_screen_bits(const unsigned long long *keys, uint64_t slots, uint64_t tier, uint64_t scale, uint32_t *bits, hipStream_t stream);
hipError_t launch_screen_count_below(const unsigned long long *keys, uint64_t slots, const uint64_t *bounds_dev, uint32_t nb,
                                     unsigned long long *below_dev, hipStream_t stream);
hipError_t launch_screen_gather(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                                const unsigned long long *keys, const uint32_t *obs, uint64_t mask,
                                uint32_t *counts_out, hipStream_t stream);

}  // namespace mg
