// screen_internal.h — launch interface between mashgpu.cpp and screen.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sketch_internal.h"

namespace mg {

struct ScreenProbeArgs {
    const uint8_t *bases;
    const SketchWork *work;
    const uint8_t *alphabet;
    const unsigned long long *keys;   // open-addressing table, ~0 = empty
    uint32_t *obs;                    // observation counter per slot
    uint64_t mask;                    // slots - 1
    uint32_t seed;
    uint32_t use64;
    uint32_t fold_case;
};

hipError_t launch_screen_build(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                               unsigned long long *keys, uint64_t mask, hipStream_t stream);
hipError_t launch_screen_probe(int k, int mode, const ScreenProbeArgs &a, uint32_t nwork, hipStream_t stream);
hipError_t launch_screen_gather(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                                const unsigned long long *keys, const uint32_t *obs, uint64_t mask,
                                uint32_t *counts_out, hipStream_t stream);

}  // namespace mg
