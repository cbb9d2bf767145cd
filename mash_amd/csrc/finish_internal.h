// finish_internal.h — launch interface between host_compare.cpp and finish.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mg {

struct FinishPair {            // = mg_pair (include/mashgpu.h)
    uint32_t numer, denom;
    double distance, p_value;
    uint8_t pass;
    uint8_t _pad[7];
};

struct FinishEdge {            // = mg_result (include/mashgpu.h)
    uint32_t row, col, numer, denom;
    double distance, p_value;
};

struct FinishArgs {
    const uint2 *counts;           // {numer, denom} in the layout the compare kernels write
    uint64_t pairs;
    uint64_t first_row;            // triangle row / query index of counts[0]
    uint64_t ncols;                // rect: number of references
    const uint64_t *len_row;       // Reference::length by row (triangle row i / query)
    const uint64_t *len_col;       // ... by column (triangle column j / reference)
    const uint32_t *min_numer;     // [s + 1] smallest numer passing the distance filter per denom; nullptr = no filter
    const uint32_t *lut_start;     // [s + 1] first entry of denom's row in lut, 0xFFFFFFFF = not tabulated
    const double *lut;             // distances by (denom, numer), host libm
    double kmer_space;
    double max_p;                  // < 0: p-value filter off
    uint32_t s;
    uint32_t triangle;
    FinishPair *pairs_out;         // finish_pairs_kernel
    unsigned long long *masks;     // [finish_mask_words(pairs)] ballots of pass A
    uint32_t *seg_count;           // [finish_segments(pairs)]
    unsigned long long *seg_off;   // exclusive scan of seg_count
    uint32_t *denom_seen;          // [s + 1] denominators carried by survivors (pass A)
    FinishEdge *edges;             // pass B: survivors with rank in [win_lo, win_lo + win_n)
    uint64_t win_lo, win_n;
    const uint2 *list_rc;          // list mode: counts[idx] belongs to pair {row, col} = list_rc[idx] (reference order); nullptr: flat order
};

uint64_t finish_segments(uint64_t pairs);
uint64_t finish_mask_words(uint64_t pairs);
hipError_t launch_finish_pairs(const FinishArgs &a, hipStream_t stream);
hipError_t launch_finish_mark(const FinishArgs &a, unsigned long long *total, hipStream_t stream);
hipError_t launch_finish_write(const FinishArgs &a, hipStream_t stream);
hipError_t launch_denom_flags(const uint2 *counts, uint64_t pairs, uint32_t s, uint32_t *seen, hipStream_t stream);

}  // namespace mg
