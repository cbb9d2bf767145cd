// index_build.h -- the inverted index of a sketch table, built by kernels that use what the table already is:
// n ascending rows of (near-)uniformly spread hashes.  Interface between host_compare.cpp and index_build.hip.
//
// What the index is (unchanged since round 4, compare_sparse.hip): every (value, row) entry of the rows' first
// min(nhash, s) hashes sorted by value, rows ascending inside a value; per sorted position the value and the row, at a
// group's first position the group's end, and two images in the table's layout -- the code (2 x group start + shared
// bit) and the entry's own sorted position.
//
// How it is built here (round 5; rounds 3-4: rocPRIM's radix sort in 5-7 passes + 4-byte scattered write-back):
//   buckets   bucket(v) = v >> shift, about 1 300 - 2 500 entries each (room for 6 144: shared values make the fill uneven); a WINDOW is 2^k consecutive buckets;
//   K0        per row the position at which every window starts (rows ascend: a row's entries of one window are contiguous);
//   K1 + K2   entries per (block of 512 rows, bucket) -> where every bucket and every block's share of it starts;
//   K3        a tile = (block of rows, window): the rows' segments are read, sorted by bucket in LDS (stable: by row, then
//             position) and written as ONE contiguous piece per bucket -- a single pass replaces the radix passes, because a
//             tile's entries fall into a few hundred buckets only.  An entry travels as one 64-bit word: its value's bits
//             below the bucket | its row;
//   K4        a workgroup per bucket sorts its entries by (value, row) in LDS (counting sort on the next 13 bits, then
//             rank among the few entries that agree in them), finds the groups of equal values there -- no head flags, no
//             scan, no tie repair over the whole index -- and writes values, rows, group ends and, in the order the
//             entries ARRIVED in, {code, position};
//             K3 also leaves, in the position image, WHERE it put every entry;
//   K5        the tiles again (their order keeps the gathered lines in the L2): {code, position} are read back from there
//             and written into the images row segment by row segment (no 4-byte scattered stores).
//   K4b       a bucket beyond the LDS capacity (a value held by thousands of rows -- sketches of one species) is sorted in two
//             levels, the first through global memory; a value with more holders than the LDS takes is already in row order
//             and streamed out;
// A table that does not fit the assumptions (values clumped far from uniform: thousands of entries that agree in the 13
// bits below their bucket without being equal; too many buckets) raises a flag and the caller builds the index the old way.
#pragma once
#include <stdint.h>

namespace mg {

struct IxGeom {
    uint32_t n, nblk, E;
    uint32_t shift;       // bucket = value >> shift
    uint32_t bw_log, BW;  // buckets per window (a power of two, at most IX_BW_MAX)
    uint32_t NW, Bp;      // windows; buckets incl. the last window's padding (NW * BW)
    uint32_t rb;          // bits of a row index in the packed word: word = (value's low `shift` bits) << rb | row
    uint32_t npass;       // 4-bit passes of the tile's stable sort by bucket
    uint32_t wgrp;        // tile order: groups of this many windows, inside a group block-major
    uint32_t nseq;        // length of the tile sequence (incl. the windows past NW of the last group)
    uint32_t rs;          // row stride of the images
    uint32_t want_gs;     // 1: also write the group start of every sorted position (dense groups' leader search)
    uint64_t stride;      // row stride of the table (entries)
};

// The dense groups' leader search (compare_dense.hip), done where every group of equal values stands in LDS anyway: a value's
// first holder inside a group of rows is its LEADER if a second holder follows; {group, value's first position, own position}
// goes to one of `nsub` lists (by bucket).  cnt[nsub] counts what was ASKED for (beyond cap_sub nothing is written).
struct IxLeaders {
    const uint32_t *grp_of = nullptr;     // [4 n] per row {its group (0xFFFFFFFF: none), the group's first row, one past its last, 0} --
                                          // one 16-byte gather per entry instead of two dependent ones (nullptr: no leader search)
    unsigned long long *key = nullptr;    // [nsub * cap_sub] (group << 32) | first sorted position of the value
    uint32_t *val = nullptr;              // [nsub * cap_sub] the leader's own sorted position
    uint32_t *cnt = nullptr;              // [nsub], zeroed by the caller
    uint32_t cap_sub = 0, nsub = 0;       // nsub: a power of two
};

struct IxPlan {
    IxGeom g;
    bool ok = false;
    const char *why = "";
    size_t lb_bytes = 0, cnt_bytes = 0, start_bytes = 0, big_bytes = 0, pk_bytes = 0, tc_bytes = 0;     // scratch
};

// flags[] (device, 4 u32, zeroed by the caller): IXF_DEGENERATE != 0: the build stopped, the arrays are not an index;
// the fullest bucket; the buckets that went through the two-level sort (ix_big_bucket_kernel) and the values with more
// holders than the LDS takes that were streamed out there
enum { IXF_NSTREAMED = 0, IXF_DEGENERATE = 1, IXF_MAXBUCKET = 2, IXF_NBIG = 3 };

// dens0: entries per unit of the hash range where the table is densest (sum over rows of count / (largest hash + 1))
IxPlan index_plan(uint32_t n, uint32_t E, uint32_t s, uint32_t rs, uint64_t stride, uint64_t maxv, double dens0, bool want_gs);
size_t index_stat_scratch_bytes();
// All buffers are the caller's.  lb / cnt / start / big / pk / tc: scratch of the plan's sizes; the rest as sparse_build_index
// (compare_internal.h) -- on return (stream order) the index arrays are complete unless flags say otherwise.
// *incidences, *max_group, *groups, flags[4]: zeroed by the caller.
// test knob: two entries of one value swapped behind the partition (the bucket sorts' order check must fire); done: 1 u32, device
hipError_t index_debug_swap(const IxPlan &plan, void *pk, const void *start, uint32_t *done, hipStream_t stream);
hipError_t index_build(const IxPlan &plan, const uint64_t *hashes, const uint32_t *off, void *lb, void *cnt, void *start, void *big, void *pk, void *tc,
                       uint64_t *keys_sorted, uint32_t *sorted_rows, uint32_t *gend, uint32_t *gs_of, uint32_t *code_img, uint32_t *pos_img,
                       void *stat_scratch, unsigned long long *incidences, uint32_t *max_group, uint32_t *groups, uint32_t *flags,
                       const IxLeaders *leaders, hipStream_t stream, int stages = 7);
// The copy of the table in another order of its rows (out[a] = row inv[a], whole rows) that also leaves the window offsets of
// the copy's rows in lb (K0 inside the copy); cnt_table: entry counts in the table's order.  index_build on `out` then
// takes stage bit 8.
hipError_t index_gather_rows(const IxPlan &plan, const uint64_t *hashes, const uint32_t *inv, const uint32_t *cnt_table, uint64_t *out, void *lb,
                             hipStream_t stream);
// stages (bits): 8 = lb is made already (index_gather_rows), 1 = the partition (K0 - K3; needs neither the statistics' memory nor the leaders'), 2 = the bucket sorts (K4, K4b:
// values, rows, groups, statistics, leaders are final behind them), 4 = the images (K5).  A caller prepares the leader search
// while stage 1 runs, and queues its copy of the statistics between stages 2 and 4.

// MASHGPU_SPARSE_INDEX=verify: out2[0] += words of a and b that differ, out2[1] = min(out2[1], the first such word);
// mode 0: all words, 1: where cond[i] == i, 2: where cond[i] != 0xFFFFFFFF
hipError_t index_verify_words(const uint32_t *a, const uint32_t *b, const uint32_t *cond, uint32_t mode, uint64_t count, unsigned long long *out2,
                              hipStream_t stream);

}  // namespace mg
