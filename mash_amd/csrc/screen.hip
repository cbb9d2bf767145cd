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

#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "screen_internal.h"

namespace mg {

__global__ void screen_build_kernel(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                                    unsigned long long *keys, uint64_t mask, unsigned long long *distinct)
{
    const uint64_t total = n * s;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t fresh_total = 0;                              // (uniform per wave)
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const uint64_t i = e / s, j = e - i * s;
        uint32_t k = nhash[i];
        if (k > s) k = (uint32_t)s;
        const unsigned long long key = j < k ? hashes[e] : SCR_EMPTY;      // (~0 is reserved as the empty marker)
        bool fresh = false;
        if (key != SCR_EMPTY) {
            uint64_t slot = scr_slot(key, mask);
            for (;;) {
                const unsigned long long old = atomicCAS(&keys[slot], SCR_EMPTY, key);
                if (old == SCR_EMPTY) { fresh = true; break; }
                if (old == key) break;
                slot = (slot + 1) & mask;
            }
        }
        fresh_total += (uint32_t)__popcll(__ballot(fresh));
    }
    if (distinct && (threadIdx.x & 63) == 0 && fresh_total) atomicAdd(distinct, (unsigned long long)fresh_total);
}

// ---- index of the table by slot (resident database): which rows hold the key of a slot --------------------
// count -> exclusive scan -> fill; after the fill slot_end[slot] is the END of the slot's run in `ent`
// (its start: the end of the slot before it).  Rows inside a run come in no particular order.
__global__ void screen_count_kernel(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                                    const unsigned long long *keys, uint64_t mask, uint32_t *slot_cnt)
{
    const uint64_t total = n * s;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const uint64_t i = e / s, j = e - i * s;
        uint32_t k = nhash[i];
        if (k > s) k = (uint32_t)s;
        if (j >= k) continue;
        uint64_t slot;
        if (scr_find(keys, mask, hashes[e], &slot)) atomicAdd(&slot_cnt[slot], 1u);
    }
}

__global__ void screen_fill_kernel(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                                   const unsigned long long *keys, uint64_t mask, uint32_t *slot_end, uint32_t *ent)
{
    const uint64_t total = n * s;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const uint64_t i = e / s, j = e - i * s;
        uint32_t k = nhash[i];
        if (k > s) k = (uint32_t)s;
        if (j >= k) continue;
        uint64_t slot;
        if (scr_find(keys, mask, hashes[e], &slot)) ent[atomicAdd(&slot_end[slot], 1u)] = (uint32_t)i;
    }
}

// per touched slot: one hit {row, count, hash} for every row that holds its key
__global__ void screen_hits_kernel(const uint32_t *touched, uint64_t nt, const unsigned long long *keys, const uint32_t *obs,
                                   const uint32_t *slot_end, const uint32_t *ent, ScreenHit *hits, unsigned long long *cursor,
                                   uint64_t cap)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    const uint32_t slot = touched[t];
    const uint32_t b = slot ? slot_end[slot - 1] : 0, e = slot_end[slot];
    if (e == b) return;
    const unsigned long long at = atomicAdd(cursor, (unsigned long long)(e - b));
    if (!hits) return;                                     // counting pass
    const uint32_t c = obs[slot];
    const unsigned long long key = keys[slot];
    for (uint32_t q = b; q < e; q++) {
        const unsigned long long o = at + (q - b);
        if (o < cap) { ScreenHit h; h.row = ent[q]; h.count = c; h.hash = key; hits[o] = h; }
    }
}

// hits into (row, hash) order: two stable radix sorts of an index permutation (by hash, then by row)
__global__ void screen_hit_keys_kernel(const ScreenHit *hits, uint64_t n, const uint32_t *perm, unsigned long long *hash_out, uint32_t *row_out,
                                       uint32_t *iota_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t src = perm ? perm[i] : i;
    if (hash_out) hash_out[i] = hits[src].hash;
    if (row_out) row_out[i] = hits[src].row;
    if (iota_out) iota_out[i] = (uint32_t)i;
}

__global__ void screen_hit_permute_kernel(const ScreenHit *hits, uint64_t n, const uint32_t *perm, ScreenHit *out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = hits[perm[i]];
}

__global__ void screen_reset_kernel(const uint32_t *touched, uint64_t nt, uint32_t *obs)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nt) obs[touched[t]] = 0;
}

// second-tier filter: one bit per key above `tier` (SketchArgs::probe_bits)
__global__ void screen_bits_kernel(const unsigned long long *keys, uint64_t slots, uint64_t tier, uint64_t scale, uint32_t *bits)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += stride) {
        const unsigned long long k = keys[i];
        if (k == SCR_EMPTY || k <= tier) continue;
        const uint64_t b = __umul64hi(k - tier - 1, scale);
        atomicOr(&bits[b >> 5], 1u << (b & 31));
    }
}

// keys <= bound (second-tier planning: how many keys lie above a candidate tier)
__global__ void screen_count_below_kernel(const unsigned long long *keys, uint64_t slots, const uint64_t *bounds, uint32_t nb,
                                          unsigned long long *below)
{
    // nb <= 32 bounds, descending (bounds[q] = largest key >> (q + 1)): a key's count goes to every bound at or above
    // it; each lane keeps one bound's tally (lane q for bound q) and adds it once at the end
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long mine = 0;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x; i0 < slots; i0 += stride) {
        const uint64_t i = i0 + threadIdx.x;
        const unsigned long long k = i < slots ? keys[i] : SCR_EMPTY;
        for (uint32_t q = 0; q < nb; q++) {
            const uint32_t c = (uint32_t)__popcll(__ballot(k != SCR_EMPTY && k <= bounds[q]));
            if (lane == q) mine += c;
            if (c == 0) break;                             // (bounds descend: nothing below this one, nothing below the next)
        }
    }
    if (lane < nb && mine) atomicAdd(&below[lane], mine);
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

// 6-frame translation of a nucleotide batch (CommandScreen.cpp:516-531 + translate/aaFromCodon,
// :617-809) for amino-acid query sketches.  The reference translates each `*`-joined chunk of
// reads as one string in frames 0..2 of the chunk and of its reverse complement; a codon that
// holds anything but upper-case ACGT (after the optional case fold) becomes '*', which no k-mer
// may contain -- so codons never bridge records and the set of k-mers does not depend on how
// reads were chunked.  Output: six segments of `seg` bytes, amino acids then MG_RECORD_SEP
// padding (not in the protein alphabet), ready for the table-alphabet sketch/probe pass.
__constant__ char kCodonTable[65] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";

__device__ __forceinline__ int nt_digit(uint32_t c, bool fold)
{
    if (fold) c &= 0xDFu;                                  // only a/c/g/t fold onto A/C/G/T
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return -1;
    }
}

__global__ void translate6_kernel(const uint8_t *in, uint64_t n, uint8_t *out, uint64_t seg, uint32_t fold)
{
    const uint64_t total = 6 * seg;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const uint32_t fr = (uint32_t)(idx / seg);
        const uint64_t j = idx - (uint64_t)fr * seg;
        const uint32_t f = fr % 3;
        const bool rev = fr >= 3;
        const uint64_t len = n >= f ? (n - f) / 3 : 0;        // lenTrans = (l - frame) / 3
        uint8_t aa = 0x0A;
        if (j < len) {
            int d0, d1, d2;
            if (!rev) {
                const uint64_t p = f + 3 * j;
                d0 = nt_digit(in[p], fold != 0); d1 = nt_digit(in[p + 1], fold != 0); d2 = nt_digit(in[p + 2], fold != 0);
            } else {                                           // reverse complement read backwards
                const uint64_t p = n - 1 - f - 3 * j;
                d0 = nt_digit(in[p], fold != 0); d1 = nt_digit(in[p - 1], fold != 0); d2 = nt_digit(in[p - 2], fold != 0);
                if (d0 >= 0) d0 = 3 - d0;
                if (d1 >= 0) d1 = 3 - d1;
                if (d2 >= 0) d2 = 3 - d2;
            }
            aa = (d0 | d1 | d2) < 0 ? (uint8_t)'*' : (uint8_t)kCodonTable[d0 * 16 + d1 * 4 + d2];
        }
        out[idx] = aa;
    }
}

hipError_t launch_translate6(const uint8_t *in, uint64_t n, uint8_t *out, uint64_t seg, bool fold, hipStream_t stream)
{
    uint64_t blocks = (6 * seg + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(translate6_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, in, n, out, seg, fold ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_screen_build(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                               unsigned long long *keys, uint64_t mask, unsigned long long *distinct, hipStream_t stream)
{
    if (n * s == 0) return hipSuccess;
    uint64_t blocks = (n * s + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(screen_build_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, hashes, nhash, n, s, keys, mask, distinct);
    return hipGetLastError();
}

size_t screen_index_temp_bytes(uint64_t slots)
{
    size_t bytes = 0;
    uint32_t *p = nullptr;
    (void)rocprim::exclusive_scan(nullptr, bytes, p, p, 0u, (size_t)slots, rocprim::plus<uint32_t>(), (hipStream_t)0);
    return bytes;
}

// slot_end must be zeroed [slots]; ent [entries of the table]
hipError_t launch_screen_index(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s, const unsigned long long *keys,
                               uint64_t mask, uint32_t *slot_end, uint32_t *ent, void *temp, size_t temp_bytes, hipStream_t stream)
{
    if (n * s == 0) return hipSuccess;
    uint64_t blocks = (n * s + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(screen_count_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, hashes, nhash, n, s, keys, mask, slot_end);
    hipError_t e = rocprim::exclusive_scan(temp, temp_bytes, slot_end, slot_end, 0u, (size_t)(mask + 1), rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(screen_fill_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, hashes, nhash, n, s, keys, mask, slot_end, ent);
    return hipGetLastError();
}

hipError_t launch_screen_hits(const uint32_t *touched, uint64_t nt, const unsigned long long *keys, const uint32_t *obs,
                              const uint32_t *slot_end, const uint32_t *ent, ScreenHit *hits, unsigned long long *cursor, uint64_t cap,
                              hipStream_t stream)
{
    if (nt == 0) return hipSuccess;
    hipLaunchKernelGGL(screen_hits_kernel, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, stream, touched, nt, keys, obs, slot_end, ent,
                       hits, cursor, cap);
    return hipGetLastError();
}

size_t screen_sort_temp_bytes(uint64_t n)
{
    size_t a = 0, b = 0;
    unsigned long long *k64 = nullptr;
    uint32_t *k32 = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, a, k64, k64, k32, k32, (size_t)n, 0, 64, (hipStream_t)0);
    (void)rocprim::radix_sort_pairs(nullptr, b, k32, k32, k32, k32, (size_t)n, 0, 32, (hipStream_t)0);
    return a > b ? a : b;
}

// scratch: k64a, k64b [n] u64; u32a..u32d [n] u32.  out may not alias hits.
hipError_t launch_screen_sort_hits(const ScreenHit *hits, uint64_t n, ScreenHit *out, unsigned long long *k64a, unsigned long long *k64b,
                                   uint32_t *u32a, uint32_t *u32b, uint32_t *u32c, uint32_t *u32d, void *temp, size_t temp_bytes,
                                   hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const dim3 grid((uint32_t)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(screen_hit_keys_kernel, grid, block, 0, stream, hits, n, (const uint32_t *)nullptr, k64a, (uint32_t *)nullptr, u32a);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, k64a, k64b, u32a, u32b, (size_t)n, 0, 64, stream);       // by hash: perm u32b
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(screen_hit_keys_kernel, grid, block, 0, stream, hits, n, (const uint32_t *)u32b, (unsigned long long *)nullptr, u32c,
                       (uint32_t *)nullptr);
    e = rocprim::radix_sort_pairs(temp, temp_bytes, u32c, u32d, u32b, u32a, (size_t)n, 0, 32, stream);                  // stable by row: perm u32a
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(screen_hit_permute_kernel, grid, block, 0, stream, hits, n, (const uint32_t *)u32a, out);
    return hipGetLastError();
}

hipError_t launch_screen_reset(const uint32_t *touched, uint64_t nt, uint32_t *obs, hipStream_t stream)
{
    if (nt == 0) return hipSuccess;
    hipLaunchKernelGGL(screen_reset_kernel, dim3((uint32_t)((nt + 255) / 256)), dim3(256), 0, stream, touched, nt, obs);
    return hipGetLastError();
}

hipError_t launch_screen_bits(const unsigned long long *keys, uint64_t slots, uint64_t tier, uint64_t scale, uint32_t *bits, hipStream_t stream)
{
    hipLaunchKernelGGL(screen_bits_kernel, dim3(8192), dim3(256), 0, stream, keys, slots, tier, scale, bits);
    return hipGetLastError();
}

hipError_t launch_screen_count_below(const unsigned long long *keys, uint64_t slots, const uint64_t *bounds_dev, uint32_t nb,
                                     unsigned long long *below_dev, hipStream_t stream)
{
    hipLaunchKernelGGL(screen_count_below_kernel, dim3(8192), dim3(256), 0, stream, keys, slots, bounds_dev, nb, below_dev);
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
