// sort_bits.h -- how many leading bits of the values the inverted index's radix sort looks at (host_compare.cpp:
// table_sparse_index; compare_sparse.hip: sparse_sort_begin_bit is the rule for evenly spread values).  Plain C++, no
// device code, so that tests/test_sort_bits.py can compile it on its own.
#pragma once
#include <cmath>
#include <cstdint>

namespace mg {

// dens[b], b = 1..64: sum over the rows whose largest hash has bit length b of (hashes of the row) / (largest hash + 1)
// -- values per unit of the hash range: a row's s smallest hashes fill [0, its largest hash] evenly.  The values of
// the rows reaching bit length b and beyond lie at that density over [2^(b-1), 2^b), so the expected number of pairs of
// DIFFERENT values inside one bucket of 2^bb is 2^bb / 2 x sum_b 2^(b-1) x density_b^2.  Returns the begin bit of the
// sort: the largest bb that keeps that number below 2^13, rounded down to whole passes of 8 bits counted from
// end_bit, never above even_begin (the evenly-spread rule), 0 when no pass would be saved.
inline uint32_t sort_begin_bit_from_density(const double dens[65], uint32_t end_bit, uint32_t even_begin)
{
    if (even_begin == 0) return 0;
    double above = 0.0, integral = 0.0;
    for (int b = 64; b >= 1; b--) {
        above += dens[b];
        integral += std::ldexp(1.0, b - 1) * above * above;
    }
    uint32_t b_est = 0;
    if (integral > 0.0 && std::isfinite(integral)) {
        const double lg = std::log2(16384.0 / integral);      // 2^bb / 2 x integral <= 2^13
        b_est = lg <= 0.0 ? 0u : lg >= 63.0 ? 63u : (uint32_t)lg;
    }
    if (b_est >= even_begin) return even_begin;
    const uint32_t passes = (end_bit - b_est + 7u) / 8u;      // every bit of a pass that is paid for is used
    return (passes >= (end_bit + 7u) / 8u || passes * 8u >= end_bit) ? 0u : end_bit - passes * 8u;
}

}  // namespace mg
