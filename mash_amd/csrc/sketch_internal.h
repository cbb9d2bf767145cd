// sketch_internal.h — launch interface between the C-ABI layer (host_sketch.cpp) and
// the sketch kernels (sketch.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mg {

// One chunk of one sketch's byte range.  k-mer START positions [begin,end) belong
// to this chunk; bytes are readable up to `limit` (end of the sketch's range).
struct SketchWork {
    uint64_t begin, end, limit;
    uint32_t sketch;      // output row
    uint32_t slot;        // pool slot (multi-chunk sketches)
    uint32_t nchunks;     // chunks of this sketch (1 -> write the row directly)
    uint32_t _pad;
};

struct SketchArgs {
    const uint8_t *bases;
    const SketchWork *work;
    const uint8_t *alphabet;      // [256] device (MODE 2)
    uint64_t *hashes_out;         // [nsketch * s]
    uint32_t *nhash_out;          // [nsketch]
    uint64_t *pool;               // [nslots * s]
    uint32_t *pool_n;             // [nslots]
    uint64_t *g_T;                // [nsketch] shared thresholds, initialised to ~0
    uint32_t sketch_size;
    uint32_t cap;                 // LDS candidate buffer entries (power of two)
    uint32_t seed;
    uint32_t use64;
    uint32_t fold_case;
    // optional fused table probe (mash screen): every valid k-mer hash <= probe_max is looked
    // up in the open-addressing table and its observation counter incremented
    const unsigned long long *probe_keys;   // nullptr: plain sketching
    uint32_t *probe_obs;
    uint64_t probe_mask;
    uint64_t probe_max;
    // slots whose counter went 0 -> 1 are appended here (any order): what a job touched, so that its results
    // and the reset for the next job cost O(touched) instead of O(table) (resident database, mg_screen_reset)
    uint32_t *probe_touched;                 // nullptr: no list
    unsigned long long *probe_ntouched;
    uint64_t probe_touched_cap;
    // two-tier key bound (databases mixing small and large genomes): a hash in (probe_tier, probe_max] is
    // looked up only if its bit in probe_bits is set -- bit = mulhi(hash - probe_tier - 1, probe_bits_scale);
    // the bitmap is a filter, a set bit without a key costs one futile probe.  probe_tier == probe_max: one tier.
    uint64_t probe_tier;
    const uint32_t *probe_bits;
    uint64_t probe_bits_scale;
    const uint64_t *seed_T;       // [nsketch] seeded thresholds (HPAD = none), or nullptr
    // packed input read by the kernel itself (round 5; include/mashgpu.h: two bits per base + one invalid bit): base number
    // pskip + o of `packed` / bit pmskip + o of `pmask` stand for the byte at offset o of the batch; `bases` is unused then.
    // The tile in LDS is ASCII either way: the hash runs over the k-mer's characters (ingest.hip says why).
    const uint32_t *packed;       // nullptr: ASCII in `bases`
    const uint32_t *pmask;        // nullptr: every base valid
    uint32_t pskip, pmskip;       // < 16, < 32
};

// Merge of pool slots first_slot, first_slot+stride, ... (nchunks of them).
// to_pool = 1: the result replaces slot first_slot (first level of a two-level merge of a
// sketch with many chunks); 0: it becomes row `sketch` of the output.
struct MergeWork {
    uint32_t sketch, first_slot, nchunks;
    uint32_t stride : 31, to_pool : 1;
};

struct MergeArgs {
    const MergeWork *work;
    uint64_t *pool;
    uint32_t *pool_n;
    uint64_t *hashes_out;
    uint32_t *nhash_out;
    uint32_t sketch_size;
    uint32_t cap;
};

struct CountArgs {
    const uint8_t *bases;
    const SketchWork *work;
    const uint8_t *alphabet;
    const uint64_t *hashes;       // final sketch rows [nsketch * s]
    const uint32_t *nhash;
    uint32_t *counts;             // [nsketch * s]
    unsigned long long *firstpos; // [nsketch * s] phase 0: first occurrence (byte offset) of every kept hash;
                                  // phase 2: first occurrence AFTER prevpos (initialise to ~0)
    const unsigned long long *prevpos; // [nsketch * s] phase 2 only
    const unsigned long long *tstar;   // [nsketch] phase 1: only occurrences at positions <= tstar count
    uint32_t sketch_size;
    uint32_t seed;
    uint32_t use64;
    uint32_t fold_case;
    uint32_t phase;               // 0: multiplicities + first positions; 1: recount the largest kept hash;
                                  // 2: next occurrence after prevpos (minCov > 1: position of the m-th occurrence)
};

// exact multiplicities of every k-mer hash in [lo, hi] (minCov > 1, see range_count_kernel)
struct RangeCountArgs {
    const uint8_t *bases;
    const SketchWork *work;
    const uint8_t *alphabet;
    unsigned long long *keys;     // open-addressing table, ~0 = empty
    uint32_t *cnts;
    uint32_t *overflow;           // set when an insertion ran out of probes
    uint64_t mask;                // slots - 1
    uint64_t lo, hi;              // inclusive hash range
    uint32_t seed;
    uint32_t use64;
    uint32_t fold_case;
};
// every k-mer hash <= bound with its position, appended in arbitrary order (the sequential
// heap of `mash sketch -r -c` is replayed on the host over this thinned stream)
struct HashEvent {
    unsigned long long hash, pos;
};
struct EventArgs {
    const uint8_t *bases;
    const SketchWork *work;
    const uint8_t *alphabet;
    HashEvent *out;
    unsigned long long *count;    // events produced (may exceed capacity)
    uint64_t capacity;
    uint64_t bound;               // inclusive
    uint32_t seed;
    uint32_t use64;
    uint32_t fold_case;
};
hipError_t launch_hash_events(int k, int mode, const EventArgs &a, uint32_t nwork, hipStream_t stream);
hipError_t launch_range_count(int k, int mode, const RangeCountArgs &a, uint32_t nwork, hipStream_t stream);
hipError_t launch_range_extract(const unsigned long long *keys, const uint32_t *cnts, uint64_t slots,
                                uint32_t min_copies, unsigned long long *out, unsigned long long *out_n,
                                uint64_t out_cap, hipStream_t stream);

// geometry: threads per workgroup and LDS candidate capacity for sketch size s;
// false if s is too large for the LDS-resident selector.
bool sketch_geometry(uint64_t s, int *nt_out, uint32_t *cap_out);
uint32_t sketch_tile(int nt);                       // k-mer starts per tile
size_t sketch_smem_bytes(uint32_t cap, int nt);

// mode 0: DNA canonical, 1: DNA forward only, 2: table alphabet forward only
hipError_t launch_sketch_chunks(int k, int mode, int nt, const SketchArgs &a, uint32_t nwork,
                                hipStream_t stream);
hipError_t launch_merge_chunks(int nt, const MergeArgs &a, uint32_t nwork, hipStream_t stream);
// multiplicities (MinHashHeap counts, incl. the reference's order-dependent count of the
// largest kept hash): see count_chunks_kernel
bool count_supported(uint64_t s);
hipError_t launch_count_chunks(int k, int mode, const CountArgs &a, uint32_t nwork, hipStream_t stream);
hipError_t launch_count_tstar(const uint32_t *nhash, uint32_t *counts, const unsigned long long *firstpos,
                              unsigned long long *tstar, uint32_t *need_fix, uint32_t nsketch, uint32_t s,
                              hipStream_t stream);

// ingest.hip: a range of the packed transport format (two bits per base + one invalid bit per base) back to the bytes the
// ASCII path is handed: out[j], j < n = base skip + j of `packed` (skip < 16), invalid bit mskip + j of `mask` (mskip < 32;
// nullptr: none set).  packed / mask 4-byte aligned and readable 8 bytes past their last used byte; out 16-byte aligned,
// (n + 15) / 16 * 16 bytes.
hipError_t launch_unpack_bases(const uint8_t *packed, const uint8_t *mask, uint32_t skip, uint32_t mskip, uint64_t n, uint8_t *out,
                               hipStream_t stream);

}  // namespace mg
