// compare_sparse_x.h -- launch interface of compare_sparse_x.hip (opt-in additions to the inverted-index engine).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "compare_internal.h"

namespace mg {

hipError_t launch_sparse_fill_value(uint2 *out, uint64_t pairs, uint32_t numer, uint32_t denom, uint32_t blocks_per_cu, uint32_t cus,
                                    hipStream_t stream);
hipError_t launch_sparse_merge_pack(const SparseArgs &a, uint64_t expect, uint32_t *chunks, void *temp, size_t temp_bytes, bool *used,
                                    hipStream_t stream);
// runs of the index that name the same rows: per row all but one of them are emptied in lohi
hipError_t launch_sparse_run_dedupe(const uint32_t *gstart, const uint32_t *sorted_rows, uint32_t G, const uint32_t *off,
                                    const uint32_t *rank_img, uint32_t rs, uint2 *lohi, uint32_t n, uint32_t max_cnt,
                                    unsigned long long *dig, unsigned long long *removed, hipStream_t stream);

}  // namespace mg
