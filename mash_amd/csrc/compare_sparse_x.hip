// compare_sparse_x.hip -- additions to the inverted-index engine (compare_sparse.hip) that were built and measured at
// the END of round 3, after the round's rocprofv3 / PMC passes.  They live in their own file, and are OPT-IN, so that
// the kernels a default run launches are exactly the ones profiles/compare_*_pmc.json and r03_kernel_stats_*.csv were
// read on (bench.py drops PMC figures when compare_sparse.hip changes).  Next round: profile, then make them the default.
//
//   MASHGPU_SPARSE_MERGE_PACK=1   several rows per merge work item (C3: merge 4.38 -> 3.91 ms, s = 400: 1.88 -> 1.31 ms;
//                                 parity: every compare test and 1 800 fuzz tables with the kernel forced)
//   MASHGPU_SPARSE_ONE_CLASS=1    a table that is nothing but copies of one sketch is answered by the fill alone
//                                 (all-identical bracket 14.5 -> 7.5 ms; tests/test_gpu_parity.py::test_compare_table_of_copies)
//   MASHGPU_SPARSE_RUN_DEDUP=1    runs of the index that name the same rows are read once per row (clades); NOT yet
//                                 validated on the GPU
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include "compare_internal.h"
#include "compare_sparse_x.h"

namespace mg {

constexpr uint32_t SPM_NT = 128;                          // as in compare_sparse.hip
constexpr uint32_t SPM_RING = 32;
constexpr uint32_t SPM_AWIN = 2048;

typedef uint32_t sp_u32x4 __attribute__((ext_vector_type(4)));

// ---- fill with any {numer, denom} (compare_sparse.hip's fill writes {0, denom}) ----------------------------------
__global__ __launch_bounds__(256) void sp_fill_value_kernel(uint2 *out, uint64_t pairs, uint32_t numer, uint32_t denom)
{
    const uint64_t head = ((reinterpret_cast<uintptr_t>(out) & 8u) != 0 && pairs > 0) ? 1u : 0u;
    const uint64_t nvec = (pairs - head) >> 1;
    sp_u32x4 *body = reinterpret_cast<sp_u32x4 *>(out + head);
    const sp_u32x4 v = {numer, denom, numer, denom};
    const uint64_t lane = threadIdx.x & 63u;
    const uint64_t gw = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6), tw = (uint64_t)gridDim.x * 4u;
    for (uint64_t base = gw * 256u; base < nvec; base += tw * 256u) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t i = base + (uint64_t)u * 64u + lane;
            if (i < nvec) __builtin_nontemporal_store(v, body + i);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (head) out[0] = make_uint2(numer, denom);
        if (((pairs - head) & 1u) != 0) out[pairs - 1] = make_uint2(numer, denom);
    }
}

hipError_t launch_sparse_fill_value(uint2 *out, uint64_t pairs, uint32_t numer, uint32_t denom, uint32_t blocks_per_cu, uint32_t cus,
                                    hipStream_t stream)
{
    if (pairs == 0) return hipSuccess;
    uint64_t blocks = (pairs / 2 + 1023) / 1024;
    const uint64_t most = (uint64_t)(cus ? cus : 256) * (blocks_per_cu ? blocks_per_cu : 16);
    if (blocks > most) blocks = most;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(sp_fill_value_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, out, pairs, numer, denom);
    return hipGetLastError();
}

// ---- several rows per merge work item ----------------------------------------------------------------------------
// The same merge with SEVERAL rows per work item.  A collection's rows have tens of candidates each (C3: 0 ... 99,
// 50 on average), so an item of one row fills half its lanes.  Here the candidates of all rows form one line of
// UNITS in visiting order -- a row with candidates takes max(candidates, SPM_PACK_MIN) units, so that at most
// SPM_PACK_ROWS rows meet in the 128 units of an item -- and item t takes units [128 t, 128 t + 128): every lane
// finds its row among the item's (at most one row boundary lies between two units 32 apart, so the rows of the
// units 0, 32, 64, 96 and 127 are all there are), all of them are staged in LDS, and the loop is the one above
// with a per-lane row base.  Lanes on a row's padding units idle (C3: 9 % against 43 %).
__global__ __launch_bounds__(256) void sp_pack_costs_kernel(const uint32_t *seg_cnt, uint32_t nrows, uint32_t pack_min, uint32_t *chunks)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r < nrows) {
        const uint32_t c = seg_cnt[r];
        chunks[r] = c ? (c < pack_min ? pack_min : c) : 0u;
    }
}

// SPM_PACK_MIN: units a row with candidates takes at least (32: up to 5 rows per item, 43: 4, 64: 3 -- fewer rows staged
// = more workgroups per CU, more padding units = more idle lanes)
template <bool RECT, uint32_t SPM_PACK_MIN>
__global__ __launch_bounds__(SPM_NT) void sp_merge_pack_kernel(SparseArgs a)
{
    constexpr uint32_t SPM_PACK_ROWS = (SPM_NT + SPM_PACK_MIN - 1u) / SPM_PACK_MIN + 1u;
    extern __shared__ __align__(16) uint32_t lds[];
    __shared__ uint32_t pslot[SPM_PACK_ROWS];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t nrows = a.row_end - a.row_begin;
    const uint32_t total = a.chunk_inc[nrows - 1];         // units in all (inclusive scan of the rows' costs)
    const uint32_t u0 = blockIdx.x * SPM_NT;
    if (u0 >= total) return;
    if (tid < SPM_PACK_ROWS) {                             // the rows of units 0, 32, 64, 96, 127 of this item
        uint32_t u = u0 + (tid == SPM_PACK_ROWS - 1u ? SPM_NT - 1u : tid * SPM_PACK_MIN);
        if (u >= total) u = total - 1u;
        uint32_t lo = 0, hi = nrows - 1;                   // first slot whose inclusive cost exceeds u
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.chunk_inc[mid] > u) hi = mid; else lo = mid + 1;
        }
        pslot[tid] = lo;
    }
    __syncthreads();
    const uint32_t s = a.s;
    // stage the item's rows (copies are compared through their representatives)
    {
        uint32_t nd = 0;
        for (uint32_t j = 0; j < SPM_PACK_ROWS; j++) {
            if (j && pslot[j] == pslot[j - 1]) continue;
            const uint32_t sl = pslot[j];
            const uint32_t row = a.order ? a.order[sl] : a.row_end - 1u - sl;
            const uint32_t ar = (a.rep && !RECT) ? a.rep[row] : row;
            const uint32_t n = a.off[ar + 1] - a.off[ar];
            const uint4 *src = reinterpret_cast<const uint4 *>(a.row_img + (uint64_t)ar * a.rs_row);
            uint4 *dst = reinterpret_cast<uint4 *>(lds + nd * a.rs_row);
            const uint32_t nvec = (n >> 2) + 1u;           // (and one chunk of the padding: A[nA] is read by a lane that has just finished)
            for (uint32_t v = tid; v < nvec; v += SPM_NT) dst[v] = src[v];
            nd++;
        }
    }
    // this lane's unit -> row, candidate
    const uint32_t u = u0 + tid;
    const uint32_t sA = pslot[tid / SPM_PACK_MIN], sB = pslot[tid / SPM_PACK_MIN + 1u];
    uint32_t slot = sA;
    if (sB != sA && u >= a.chunk_inc[sB - 1u]) slot = sB;  // (sB > sA: the units before sB's first end with the slot before it)
    const uint32_t q = u - (slot ? a.chunk_inc[slot - 1u] : 0u);
    const uint32_t cnt = a.seg_cnt[slot];
    const bool have = u < total && q < cnt;
    uint32_t ridx = 0;
#pragma unroll
    for (uint32_t j = 1; j < SPM_PACK_ROWS; j++) ridx += (pslot[j] != pslot[j - 1] && pslot[j] <= slot) ? 1u : 0u;
    const uint32_t row = a.order ? a.order[slot] : a.row_end - 1u - slot;
    const uint32_t arow = (a.rep && !RECT) ? a.rep[row] : row;
    const uint32_t nA = a.off[arow + 1] - a.off[arow];
    const uint32_t *A = lds + ridx * a.rs_row;
    const uint64_t at = a.seg_base[slot] + q;              // this lane's candidate (and result slot)
    uint32_t *myring = lds + SPM_PACK_ROWS * a.rs_row + (tid >> 6) * (SPM_RING * 64u) + lane;   // code e of this lane: myring[(e & 31) * 64]
    uint32_t j = have ? a.cand[at].y : 0u;
    if (a.rep) j = a.rep[j];
    const bool same = !RECT && a.rep != nullptr && j == arow;
    const uint32_t nB = have ? a.col_cnt_off[j + 1] - a.col_cnt_off[j] : 0u;
    const uint4 *B4 = reinterpret_cast<const uint4 *>(a.col_img + (uint64_t)j * a.rs_col);
    auto land = [&](uint32_t e, const uint4 &v) {         // codes e .. e + 3 (e a multiple of 4)
        uint32_t *p = myring + (e & (SPM_RING - 1u)) * 64u;
        p[0] = v.x; p[64] = v.y; p[128] = v.z; p[192] = v.w;
    };
    {
        const uint4 x0 = B4[0], x1 = B4[1], x2 = B4[2], x3 = B4[3];
        land(0, x0); land(4, x1); land(8, x2); land(12, x3);
    }
    uint4 p0 = B4[4], p1 = B4[5];
    uint32_t loaded = 16;
    bool pend = true;
    __syncthreads();                                     // rows staged
    uint32_t ia = 0, ib = 0, denom = 0;                  // (common = ia + ib - denom)
    bool active = have && !same && s > 0 && nA > 0 && nB > 0;
    while (__ballot(active) != 0) {
        uint32_t room = 0;
        if (active) {
            const uint32_t ra = nA - ia, rb = nB - ib, rd = s - denom;
            room = ra < rb ? ra : rb;
            room = room < rd ? room : rd;
        }
        if (__ballot(active && room < 8u) == 0) {
            if (active) {
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const uint32_t av = A[ia];
                    uint32_t bv = myring[(ib & (SPM_RING - 1u)) * 64u];
                    if (RECT) bv += 1u;
                    ia += av <= bv ? 1u : 0u;
                    ib += bv <= av ? 1u : 0u;
                }
                denom += 8;
                active = denom < s && ia < nA && ib < nB;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const uint32_t av = A[ia];
                uint32_t bv = myring[(ib & (SPM_RING - 1u)) * 64u];
                if (RECT) bv += 1u;
                const bool adva = active && av <= bv, advb = active && bv <= av;
                denom += active ? 1u : 0u;
                ia += adva ? 1u : 0u;
                ib += advb ? 1u : 0u;
                active = active && denom < s && ia < nA && ib < nB;
            }
        }
        if (pend) {
            land(loaded, p0);
            land(loaded + 4u, p1);
            loaded += 8;
        }
        pend = active && loaded + 8u - ib <= SPM_RING;
        if (pend) {
            p0 = B4[loaded >> 2];
            p1 = B4[(loaded >> 2) + 1u];
        }
    }
    if (have) {
        uint32_t common = ia + ib - denom;
        if (same) {
            common = denom = nA;
        } else if (denom < s) {                            // :367-385
            denom += (nA - ia) + (nB - ib);
            if (denom > s) denom = s;
        }
        a.res[at] = make_uint2(common, denom);
    }
}

// chunks / chunk_inc / temp as for launch_sparse_merge_rows; false in *used: the job is not one for this kernel (rows
// longer than the LDS window, too much LDS) and the caller takes launch_sparse_merge_rows
hipError_t launch_sparse_merge_pack(const SparseArgs &a, uint64_t expect, uint32_t *chunks, void *temp, size_t temp_bytes, bool *used,
                                    hipStream_t stream)
{
    *used = false;
    const uint32_t nrows = a.row_end - a.row_begin;
    if (expect == 0 || nrows == 0) return hipSuccess;
    uint32_t pack_min = 32;
    if (const char *ev = getenv("MASHGPU_SPARSE_PACK_MIN")) pack_min = atoi(ev) >= 64 ? 64u : atoi(ev) >= 43 ? 43u : 32u;
    const uint32_t pack_rows = (SPM_NT + pack_min - 1u) / pack_min + 1u;
    const size_t smem = ((size_t)pack_rows * a.rs_row + (SPM_NT / 64u) * SPM_RING * 64u) * 4;
    if (a.rs_row > SPM_AWIN + 8u || smem > 160 * 1024 - 256) return hipSuccess;
    const uint64_t items = (expect + (uint64_t)pack_min * nrows) / SPM_NT + 1;
    if (items >= (1ull << 31) || expect + (uint64_t)pack_min * nrows >= (1ull << 32)) return hipSuccess;
    hipLaunchKernelGGL(sp_pack_costs_kernel, dim3((nrows + 255u) / 256u), dim3(256), 0, stream, a.seg_cnt, nrows, pack_min, chunks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::inclusive_scan(temp, temp_bytes, (const uint32_t *)chunks, a.chunk_inc, (size_t)nrows, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e2 != hipSuccess) return e2;
        hipLaunchKernelGGL(kern, dim3((uint32_t)items), dim3(SPM_NT), smem, stream, a);
        return hipGetLastError();
    };
    *used = true;
    if (pack_min == 32) return a.triangle ? go(sp_merge_pack_kernel<false, 32>) : go(sp_merge_pack_kernel<true, 32>);
    if (pack_min == 43) return a.triangle ? go(sp_merge_pack_kernel<false, 43>) : go(sp_merge_pack_kernel<true, 43>);
    return a.triangle ? go(sp_merge_pack_kernel<false, 64>) : go(sp_merge_pack_kernel<true, 64>);
}

// ---- runs that name the same rows -----------------------------------------------------------------------------
// In a clade of near-identical sketches most values are held by the same set of rows: their runs are copies of each
// other, and a row that ORs one of them into its bitmap learns nothing from the next.  Once per index: a 128-bit
// digest of every run (which rows, in which order), then per row the entries whose run equals the run of another entry
// of the row -- same digest, same number of rows below this one, same first row -- lose their share of the work:
// their {lo, hi} becomes empty, discovery skips them like values nobody else holds.  Candidates are the rows named by
// ANY entry of a row, so dropping a copy of a run changes nothing (and two different runs are taken for copies only
// if 128 bits of digest, the length and the first row agree).
__device__ __forceinline__ uint64_t sp_mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void sp_run_digest_kernel(const uint32_t *gstart, const uint32_t *sorted_rows, uint32_t G,
                                                            unsigned long long *dig)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t lo = 0, hi = 0;
    if (g < G) { lo = gstart[g]; hi = gstart[g + 1]; }
    const uint32_t len = hi - lo;
    uint64_t d1 = 0, d2 = 0;
    if (len >= 2u && len <= 32u) {                       // (a run of one row is that row alone: never a candidate)
        for (uint32_t q = lo; q < hi; q++) {
            const uint64_t x = (uint64_t)sorted_rows[q] | ((uint64_t)(q - lo) << 32);
            d1 += sp_mix64(x);
            d2 += sp_mix64(x * 0x9E3779B97F4A7C15ULL + 0xD1B54A32D192ED03ULL);
        }
    }
    uint64_t m = __ballot(len > 32u);                    // long runs: the wave together
    while (m != 0) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        const uint32_t LO = (uint32_t)__builtin_amdgcn_readlane((int)lo, l), HI = (uint32_t)__builtin_amdgcn_readlane((int)hi, l);
        uint64_t p1 = 0, p2 = 0;
        for (uint32_t q = LO + lane; q < HI; q += 64u) {
            const uint64_t x = (uint64_t)sorted_rows[q] | ((uint64_t)(q - LO) << 32);
            p1 += sp_mix64(x);
            p2 += sp_mix64(x * 0x9E3779B97F4A7C15ULL + 0xD1B54A32D192ED03ULL);
        }
        for (int d = 32; d > 0; d >>= 1) { p1 += __shfl_xor(p1, d); p2 += __shfl_xor(p2, d); }
        if ((int)lane == l) { d1 = p1; d2 = p2; }
    }
    if (g < G) {
        dig[2ull * g] = d1 + (uint64_t)len * 0xC2B2AE3D27D4EB4FULL;
        dig[2ull * g + 1] = d2 ^ ((uint64_t)len << 40);
    }
}

// one workgroup per row; LDS: an open-addressing table of `tsize` {key, check} pairs (tsize a power of two >= 2 x the row's
// entries, so a thread handles at most 16 entries).  Two phases with a barrier between them and NO waiting inside a
// wave (lanes of a wave cannot wait for each other): (1) every entry claims the slot of its key or finds it claimed;
// claimers publish their check word; (2) the others compare theirs with the published one.
__global__ __launch_bounds__(256) void sp_run_dedupe_kernel(const uint32_t *off, const uint32_t *rank_img, uint32_t rs,
                                                            const unsigned long long *dig, const uint32_t *sorted_rows, uint2 *lohi,
                                                            uint32_t tsize, unsigned long long *removed)
{
    extern __shared__ unsigned long long tab[];          // [tsize] keys, [tsize] checks
    const uint32_t row = blockIdx.x, tid = threadIdx.x;
    const uint32_t b = off[row], cnt = off[row + 1] - b;
    if (cnt == 0 || 2u * cnt > tsize || cnt > 16u * 256u) return;      // (uniform; rows too long for the table keep all their runs)
    unsigned long long *keys = tab, *chk = tab + tsize;
    for (uint32_t i = tid; i < 2u * tsize; i += 256u) tab[i] = 0;
    __syncthreads();
    uint32_t found[16];                                  // slot + 1 where an entry found its key claimed by another; 0: it stays
    unsigned long long mycheck[16];
#pragma unroll
    for (int it = 0; it < 16; it++) {
        found[it] = 0;
        mycheck[it] = 0;
        const uint32_t p = tid + (uint32_t)it * 256u;
        if (p >= cnt) continue;
        const uint2 lh = lohi[b + p];
        const uint32_t len = lh.y - lh.x;
        if (len == 0) continue;
        const uint32_t g = rank_img[(uint64_t)row * rs + p] >> 1;
        // (equal runs put this row at the same place: equal prefix length, equal first row)
        const uint64_t salt = sp_mix64(((uint64_t)len << 32) | sorted_rows[lh.x]);
        uint64_t key = dig[2ull * g] ^ salt, check = dig[2ull * g + 1] + salt;
        if (key == 0) key = 1;
        if (check == 0) check = 1;
        uint32_t slot = (uint32_t)(key >> 17) & (tsize - 1u);
        for (uint32_t tries = 0; tries < tsize; tries++) {
            const unsigned long long old = atomicCAS(&keys[slot], 0ULL, (unsigned long long)key);
            if (old == 0) { chk[slot] = check; break; }                          // first of its kind: stays
            if (old == key) { found[it] = slot + 1u; mycheck[it] = check; break; }
            slot = (slot + 1u) & (tsize - 1u);
        }
    }
    __syncthreads();                                     // every claimed slot has its check word
    uint32_t dropped = 0;
#pragma unroll
    for (int it = 0; it < 16; it++) {
        if (found[it] == 0) continue;
        if (chk[found[it] - 1u] != mycheck[it]) continue;                        // (same key, another run: both stay)
        const uint32_t p = tid + (uint32_t)it * 256u;
        const uint2 lh = lohi[b + p];
        lohi[b + p] = make_uint2(lh.x, lh.x);
        dropped++;
    }
    if (removed && dropped) atomicAdd(removed, (unsigned long long)dropped);
}

// dig: scratch of 2 G u64.  max_cnt: the most entries a row has.  *removed += entries whose run was a copy.
hipError_t launch_sparse_run_dedupe(const uint32_t *gstart, const uint32_t *sorted_rows, uint32_t G, const uint32_t *off,
                                    const uint32_t *rank_img, uint32_t rs, uint2 *lohi, uint32_t n, uint32_t max_cnt,
                                    unsigned long long *dig, unsigned long long *removed, hipStream_t stream)
{
    if (n == 0 || G == 0) return hipSuccess;
    uint32_t tsize = 256;
    while (tsize < 2u * max_cnt && tsize < 8192u) tsize <<= 1;
    hipLaunchKernelGGL(sp_run_digest_kernel, dim3((G + 255u) / 256u), dim3(256), 0, stream, gstart, sorted_rows, G, dig);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const size_t smem = (size_t)tsize * 16;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(sp_run_dedupe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sp_run_dedupe_kernel, dim3(n), dim3(256), smem, stream, off, rank_img, rs, dig, sorted_rows, lohi, tsize, removed);
    return hipGetLastError();
}

}  // namespace mg
