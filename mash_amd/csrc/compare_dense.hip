// compare_dense.hip — gfx950: the pairs INSIDE a group of near-identical sketches, as bit-mask arithmetic.
//
// The inverted-index engine (compare_sparse.hip) pays per shared hash (discovery reads every run) and per candidate
// (a merge of ~s steps).  Inside a clade -- thousands of isolates of one species, rows that are near-copies of each
// other -- every pair is a candidate sharing ~s hashes, and both costs explode (one clade of 32 768 rows: 4.8e11
// shared hashes to read, 5.4e8 merges).  But such a group has a tiny UNIVERSE: the values held by at least two of
// its rows (about s of them for near-copies).  A row is then a bit mask over the universe plus its EXTRAS -- values no
// other row of the group holds, which can never be common and only count as union elements -- recorded by their gap
// (the number of universe values below them).  The loop of compareSketches (CommandDistance.cpp:347-385) counts the
// common values among the first s values of the union; in universe order:
//
//     common = popcount(A & B) over the universe positions e with f(e) < s,
//     f(e)   = union bits before e + extras of either row with gap <= e,        denom = min(s, |A u B|).
//
// A word of 64 universe positions is taken whole while the count at its end stays <= s (a popcount, two table
// reads); the word in which s is reached is resolved by bisection over its bit positions.  ~17 word steps per pair at
// s = 1000 instead of ~1100 merge steps.  The arithmetic is pinned against the reference's loop by a numpy model
// (tests/test_dense_model.py) and the kernels against the oracle (tests/test_gpu_parity.py).
//
// Groups are runs of CONSECUTIVE rows whose neighbours are related (collections are listed in taxonomic order); any
// set of rows would be correct, so the relatedness test is a sample.  Layout per group, in blocks of 128 rows:
// mask word w of the block's rows side by side (w * 128 + lane, u64), then the cumulative extra counts
// cx[w] = extras with gap < 64 w (w = 0 .. W, u16, the same way); extras as u16 gaps per row.
// mashgpu.cpp::table_sparse_index builds this next to the inverted index, whose runs it then clips so that discovery
// sees only the partners OUTSIDE a row's group; run_compare_sparse launches dn_pairs_kernel after the fill.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "compare_internal.h"

namespace mg {

// ------------------------------------------------------------------------------------------------
// which neighbours are related: link[i] = 1 if at least half of the first 64 values of row i occur among the first
// 256 of row i - 1 (the smallest hashes of a sketch are as good a sample as any)
__global__ __launch_bounds__(256) void dn_neighbor_kernel(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt, uint32_t n,
                                                          uint8_t *link)
{
    const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (row >= n) return;
    uint32_t hit = 0;
    uint32_t ci = 0, cj = 0;
    if (row > 0) {
        ci = cnt[row];
        cj = cnt[row - 1];
        const uint32_t take = ci < 64u ? ci : 64u, span = cj < 256u ? cj : 256u;
        if (lane < take && span > 0) {
            const uint64_t v = hashes[(uint64_t)row * stride + lane];
            const uint64_t *q = hashes + (uint64_t)(row - 1) * stride;
            uint32_t lo = 0, hi = span;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (q[mid] < v) lo = mid + 1; else hi = mid;
            }
            hit = (lo < span && q[lo] == v) ? 1u : 0u;
        }
    }
    const uint32_t hits = (uint32_t)__popcll(__ballot(hit != 0));
    const uint32_t take = ci < 64u ? ci : 64u;
    if (lane == 0) link[row] = (row > 0 && take >= 16u && cj >= 16u && 2u * hits >= take) ? 1 : 0;
}

hipError_t launch_dense_neighbors(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt, uint32_t n, uint8_t *link,
                                  hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(dn_neighbor_kernel, dim3((n + 3u) / 4u), dim3(256), 0, stream, hashes, stride, cnt, n, link);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// universe of every group: the values held by at least two of its rows.  In the sorted index the rows of a value
// ascend, a group is a row interval, so a group's holders of a value sit side by side: the first of them is the
// value's LEADER if a second one follows.  Leaders, in sorted (= value) order, are selected and then sorted by group
// (stable): every group's universe, ascending, as the sorted positions gs that are the values' codes.
__global__ __launch_bounds__(256) void dn_leader_flags_kernel(const uint32_t *sorted_rows, const uint32_t *gs_of, const uint32_t *gend,
                                                              const uint32_t *grp_of, const DenseGroup *groups, uint32_t E,
                                                              uint8_t *flag)
{
    const uint32_t pos = blockIdx.x * 256u + threadIdx.x;
    if (pos >= E) return;
    const uint32_t row = sorted_rows[pos];
    const uint32_t g = grp_of[row];
    uint8_t f = 0;
    if (g != 0xFFFFFFFFu) {
        const uint32_t g0 = groups[g].g0, g1 = groups[g].g1;
        const uint32_t gs = gs_of[pos];
        const bool first = pos == gs || sorted_rows[pos - 1] < g0;
        const bool more = pos + 1u < gend[gs] && sorted_rows[pos + 1] < g1;
        f = (first && more) ? 1 : 0;
    }
    flag[pos] = f;
}

struct dn_is_set {
    const uint8_t *flag;
    __device__ bool operator()(const uint32_t &pos) const { return flag[pos] != 0; }
};

// per leader: its group (the sort key) and the VALUE it stands for -- as the start of the value's run in the sorted index,
// which is what a row's code says (the leader itself is the first holder inside the group, not the first of the run)
__global__ __launch_bounds__(256) void dn_leader_keys_kernel(const uint32_t *lead_pos, const uint32_t *nlead, const uint32_t *sorted_rows,
                                                             const uint32_t *gs_of, const uint32_t *grp_of, uint32_t *key, uint32_t *val)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < *nlead) {
        const uint32_t pos = lead_pos[i];
        key[i] = grp_of[sorted_rows[pos]];
        val[i] = gs_of[pos];
    }
}

// after the stable sort by group: where every group's universe starts and how long it is
__global__ __launch_bounds__(256) void dn_universe_bounds_kernel(const uint32_t *key_sorted, const uint32_t *nlead, uint32_t *ustart, uint32_t *ucount)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, m = *nlead;
    if (i >= m) return;
    const uint32_t g = key_sorted[i];
    if (i == 0 || key_sorted[i - 1] != g) ustart[g] = i;
    if (i + 1u == m || key_sorted[i + 1] != g) ucount[g] = i + 1u;      // (one past the last: the host subtracts ustart)
}

size_t dense_universe_temp_bytes(uint32_t E)
{
    size_t a = 0, b = 0;
    rocprim::select(nullptr, a, rocprim::counting_iterator<uint32_t>(0u), (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)E,
                    dn_is_set{nullptr}, (hipStream_t) nullptr);
    rocprim::radix_sort_pairs(nullptr, b, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                              (size_t)E, 0u, 32u, (hipStream_t) nullptr);
    return a > b ? a : b;
}

// step 1: flag: scratch of E bytes; lead_pos: out, the leaders' sorted positions in value order; *nlead (device) their number
hipError_t dense_select_leaders(const uint32_t *sorted_rows, const uint32_t *gs_of, const uint32_t *gend, const uint32_t *grp_of,
                                const DenseGroup *groups, uint32_t E, void *temp, size_t temp_bytes, uint8_t *flag, uint32_t *lead_pos,
                                uint32_t *nlead, hipStream_t stream)
{
    if (E == 0) return hipSuccess;
    hipLaunchKernelGGL(dn_leader_flags_kernel, dim3((E + 255u) / 256u), dim3(256), 0, stream, sorted_rows, gs_of, gend, grp_of, groups, E, flag);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return rocprim::select(temp, temp_bytes, rocprim::counting_iterator<uint32_t>(0u), lead_pos, nlead, (size_t)E, dn_is_set{flag}, stream);
}

// step 2 (the host knows m = *nlead): key / key_sorted / val: scratch of m u32; ulist: out, the universes' values (run starts) grouped
// by group, ascending inside a group; ustart / uend [ngroups]: zeroed by the caller
hipError_t dense_sort_universes(const uint32_t *lead_pos, const uint32_t *nlead, uint32_t m, const uint32_t *sorted_rows, const uint32_t *gs_of,
                                const uint32_t *grp_of, void *temp, size_t temp_bytes, uint32_t *key, uint32_t *key_sorted, uint32_t *val,
                                uint32_t *ulist, uint32_t *ustart, uint32_t *uend, uint32_t group_bits, hipStream_t stream)
{
    if (m == 0) return hipSuccess;
    const uint32_t blocks = (m + 255u) / 256u;
    hipLaunchKernelGGL(dn_leader_keys_kernel, dim3(blocks), dim3(256), 0, stream, lead_pos, nlead, sorted_rows, gs_of, grp_of, key, val);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(temp, temp_bytes, (const uint32_t *)key, key_sorted, (const uint32_t *)val, ulist, (size_t)m, 0u, group_bits, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dn_universe_bounds_kernel, dim3(blocks), dim3(256), 0, stream, (const uint32_t *)key_sorted, nlead, ustart, uend);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// a grouped row -> its mask words, cumulative extra counts and extras.  One workgroup per row.
__global__ __launch_bounds__(256) void dn_encode_kernel(const uint32_t *off, const uint32_t *code_img, uint32_t rs, const uint32_t *grp_of,
                                                        const DenseGroup *groups, const uint32_t *ulist, unsigned long long *gdata,
                                                        uint16_t *ext, uint32_t xs, uint32_t n)
{
    extern __shared__ uint32_t lds[];                    // [2 * W] mask halves, [W + 1] extras per word, [8] scan scratch
    const uint32_t row = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    if (row >= n) return;
    const uint32_t g = grp_of[row];
    if (g == 0xFFFFFFFFu) return;                        // uniform
    const DenseGroup G = groups[g];
    const uint32_t W = G.W, u = G.u;
    uint32_t *mask32 = lds, *hist = lds + 2u * W, *scr = hist + W + 1u;
    for (uint32_t i = tid; i < 3u * W + 1u; i += 256u) lds[i] = 0;
    __syncthreads();
    const uint32_t cnt = off[row + 1] - off[row];
    const uint32_t *ul = ulist + G.ustart;
    uint16_t *xrow = ext + (uint64_t)(G.xrow0 + (row - G.g0)) * xs;
    uint32_t nx = 0;                                     // extras written so far (uniform)
    for (uint32_t base = 0; base < cnt; base += 256u) {
        const uint32_t p = base + tid;
        uint32_t idx = 0;
        bool extra = false;
        if (p < cnt) {
            const uint32_t gs = code_img[(uint64_t)row * rs + p] >> 1;
            uint32_t lo = 0, hi = u;                      // lower bound of gs in the universe
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (ul[mid] < gs) lo = mid + 1; else hi = mid;
            }
            idx = lo;
            if (lo < u && ul[lo] == gs) atomicOr(&mask32[idx >> 5], 1u << (idx & 31u));
            else extra = true;
        }
        // extras keep their order (entries ascend, so gaps ascend): block-wide exclusive scan of the flags
        const uint64_t bal = __ballot(extra);
        const uint32_t before = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) scr[wid] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t k = 0; k < wid; k++) woff += scr[k];
        const uint32_t total = scr[0] + scr[1] + scr[2] + scr[3];
        if (extra) {
            xrow[nx + woff + before] = (uint16_t)idx;
            atomicAdd(&hist[idx >> 6], 1u);
        }
        nx += total;
        __syncthreads();                                 // scr is reused
    }
    __syncthreads();
    // cumulative counts: cx[w] = extras with gap < 64 w (w = 0 .. W), by one wave (W + 1 <= a few dozen)
    const uint32_t j = (row - G.g0) >> 7, bl = (row - G.g0) & 127u;
    const uint64_t bw = 128ull * W + 32ull * (W + 1u);
    unsigned long long *blk = gdata + G.data_off + (uint64_t)j * bw;
    uint16_t *cxp = reinterpret_cast<uint16_t *>(blk + 128ull * W);
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t w = 0; w <= W; w++) {
            cxp[w * 128u + bl] = (uint16_t)run;
            if (w < W) run += hist[w];
        }
    }
    for (uint32_t w = tid; w < W; w += 256u)
        blk[w * 128u + bl] = (unsigned long long)mask32[2u * w] | ((unsigned long long)mask32[2u * w + 1u] << 32);
}

hipError_t launch_dense_encode(const uint32_t *off, const uint32_t *code_img, uint32_t rs, const uint32_t *grp_of, const DenseGroup *groups,
                               const uint32_t *ulist, unsigned long long *gdata, uint16_t *ext, uint32_t xs, uint32_t n, uint32_t wmax,
                               hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const size_t smem = ((size_t)3 * wmax + 1 + 8) * 4;
    hipLaunchKernelGGL(dn_encode_kernel, dim3(n), dim3(256), smem, stream, off, code_img, rs, grp_of, groups, ulist, gdata, ext, xs, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// the inverted index's runs, clipped for the rows of a group: of the rows below a row that hold one of its values,
// discovery is to see those OUTSIDE the group only (the pairs inside are this file's).  Rows ascend inside a run and
// the group is a row interval, so the partners inside are the run's tail: the new end is the first position whose
// row is not below the group's first row.
__global__ __launch_bounds__(256) void dn_clip_kernel(const uint32_t *off, const uint32_t *code_img, uint32_t *pos_img, uint32_t rs,
                                                      const uint32_t *grp_of, const DenseGroup *groups, const uint32_t *sorted_rows,
                                                      uint32_t n)
{
    const uint32_t row = blockIdx.x;
    if (row >= n) return;
    const uint32_t g = grp_of[row];
    if (g == 0xFFFFFFFFu) return;                        // uniform
    const uint32_t g0 = groups[g].g0;
    const uint32_t cnt = off[row + 1] - off[row];
    for (uint32_t p = threadIdx.x; p < cnt; p += 256u) {
        const uint64_t at = (uint64_t)row * rs + p;
        uint32_t lo = code_img[at] >> 1, hi = pos_img[at];
        if (lo == hi) continue;
        while (lo < hi) {                                  // first position in [lo, hi) with row >= g0
            const uint32_t mid = (lo + hi) >> 1;
            if (sorted_rows[mid] < g0) lo = mid + 1; else hi = mid;
        }
        pos_img[at] = lo;
    }
}

hipError_t launch_dense_clip(const uint32_t *off, const uint32_t *code_img, uint32_t *pos_img, uint32_t rs, const uint32_t *grp_of,
                             const DenseGroup *groups, const uint32_t *sorted_rows, uint32_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(dn_clip_kernel, dim3(n), dim3(256), 0, stream, off, code_img, pos_img, rs, grp_of, groups, sorted_rows, n);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// the pairs of a tile: DN_ROWS rows x the 128 columns of one block of the group, lane = column (so that a row's
// results leave as one contiguous store per wave).  The column block's words are staged in LDS as they lie in
// memory (word w of lane l at w * 128 + l: conflict free), the rows' words beside them (read by all lanes at once).
constexpr uint32_t DN_ROWS = 32;

// where a pair of index rows lands in the output: the index may have been built on a PERMUTED table (rows that belong
// together next to each other, see dense_cluster_rows); inv maps an index row back to the table's row
__device__ __forceinline__ uint64_t dn_out_index(uint32_t a, uint32_t b, const uint32_t *inv, uint64_t out_base)
{
    uint32_t i = a, j = b;
    if (inv) {
        i = inv[a];
        j = inv[b];
        if (i < j) { const uint32_t t = i; i = j; j = t; }
    }
    return (uint64_t)i * (i - 1u) / 2u - out_base + j;
}

// STREAM: the column block's words are read from global memory (L2) word by word instead of being staged -- universes of
// hundreds of words (s = 10 000) do not fit the LDS; the rows' words still sit in LDS
template <bool STREAM>
__global__ __launch_bounds__(128) void dn_pairs_kernel(const DenseTile *tiles, const DenseGroup *groups, const unsigned long long *gdata,
                                                       const uint16_t *ext, uint32_t xs, uint32_t s, uint32_t row_begin, uint32_t row_end,
                                                       uint64_t out_base, const uint32_t *inv, uint2 *out)
{
    extern __shared__ __align__(16) unsigned long long dl[];
    const DenseTile T = tiles[blockIdx.x];
    const DenseGroup G = groups[T.group];
    const uint32_t W = G.W, tid = threadIdx.x;
    const uint64_t bw = 128ull * W + 32ull * (W + 1u);
    const unsigned long long *bsrc = gdata + G.data_off + (uint64_t)T.cblk * bw;
    const unsigned long long *Bm = STREAM ? bsrc : dl;                                   // [W][128]
    const uint16_t *Bcx = reinterpret_cast<const uint16_t *>(Bm + 128ull * W);             // [W + 1][128]
    unsigned long long *Am = dl + (STREAM ? 0ull : bw);                                   // [W][DN_ROWS]
    uint16_t *Acx = reinterpret_cast<uint16_t *>(Am + (uint64_t)DN_ROWS * W);             // [W + 1][DN_ROWS]
    if (!STREAM)
        for (uint32_t i = tid; i < (uint32_t)bw; i += 128u) dl[i] = bsrc[i];
    const uint32_t ra = T.row0 - G.g0;                                    // (a multiple of DN_ROWS: the tile's rows share a block)
    const unsigned long long *asrc = gdata + G.data_off + (uint64_t)(ra >> 7) * bw;
    const uint32_t la0 = ra & 127u;
    for (uint32_t i = tid; i < W * DN_ROWS; i += 128u) Am[i] = asrc[(i / DN_ROWS) * 128u + la0 + (i % DN_ROWS)];
    {
        const uint16_t *acs = reinterpret_cast<const uint16_t *>(asrc + 128ull * W);
        for (uint32_t i = tid; i < (W + 1u) * DN_ROWS; i += 128u) Acx[i] = acs[(i / DN_ROWS) * 128u + la0 + (i % DN_ROWS)];
    }
    __syncthreads();
    const uint32_t b = G.g0 + T.cblk * 128u + tid;                         // this lane's column
    const uint16_t *xb = ext + (uint64_t)(G.xrow0 + (b < G.g1 ? b - G.g0 : 0u)) * xs;
    for (uint32_t ai = 0; ai < DN_ROWS; ai++) {
        const uint32_t a = T.row0 + ai;
        if (a >= G.g1 || a >= row_end) break;                             // uniform
        if (a < row_begin) continue;                                      // uniform
        if (G.g0 + T.cblk * 128u >= a) continue;                          // uniform: for this row the whole block is at or above the diagonal
        const bool valid = b < a;
        const uint16_t *xa = ext + (uint64_t)(G.xrow0 + (a - G.g0)) * xs;
        uint32_t pu = 0, common = 0, denom = 0;
        bool done = !valid;
        for (uint32_t w = 0; w < W; w++) {
            if (__ballot(!done) == 0) break;                              // uniform
            const unsigned long long ma = Am[w * DN_ROWS + ai], mb = Bm[w * 128u + tid];
            const unsigned long long un = ma | mb, an = ma & mb;
            const uint32_t pun = (uint32_t)__popcll(un);
            const uint32_t ca1 = Acx[(w + 1u) * DN_ROWS + ai], cb1 = Bcx[(w + 1u) * 128u + tid];
            const uint32_t F = pu + pun + ca1 + cb1;
            if (!done) {
                if (F <= s) {                                             // the whole word lies before the s-th union element
                    common += (uint32_t)__popcll(an);
                    pu += pun;
                    if (F == s) { done = true; denom = s; }
                } else {
                    // s is reached inside this word: the smallest bit position t with f(t) >= s, f(t) = what lies before the
                    // word + union bits below t + extras of either row with offset <= t; bits below it are counted
                    const uint32_t ca0 = Acx[w * DN_ROWS + ai], cb0 = Bcx[w * 128u + tid];
                    const uint32_t Fprev = pu + ca0 + cb0;
                    const uint32_t na = ca1 - ca0, nb = cb1 - cb0, wbase = w << 6;
                    uint32_t lo = 0, hi = 63;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        uint32_t c = Fprev + (uint32_t)__popcll(un & ((1ull << mid) - 1ull));
                        for (uint32_t k = 0; k < na; k++) c += ((uint32_t)xa[ca0 + k] - wbase <= mid) ? 1u : 0u;
                        for (uint32_t k = 0; k < nb; k++) c += ((uint32_t)xb[cb0 + k] - wbase <= mid) ? 1u : 0u;
                        if (c >= s) hi = mid; else lo = mid + 1u;
                    }
                    common += (uint32_t)__popcll(an & ((1ull << lo) - 1ull));
                    done = true;
                    denom = s;
                }
            }
        }
        if (valid) {
            if (!done) {                                                  // the union ends before s (short sketches)
                const uint32_t total = pu + Acx[W * DN_ROWS + ai] + Bcx[W * 128u + tid];
                denom = total < s ? total : s;
            }
            out[dn_out_index(a, b, inv, out_base)] = make_uint2(common, denom);
        }
    }
}

// LDS of a tile: staged column block + rows (W up to dense_max_words_staged()), or the rows alone (streamed column block)
constexpr uint32_t DN_STAGE_WORDS = 48;
uint32_t dense_max_words_staged() { return DN_STAGE_WORDS; }
uint32_t dense_max_words() { return 400; }                 // rows of a tile in LDS: 32 x (8 W + 2 W + 2) bytes

size_t dense_pairs_lds(uint32_t W)
{
    const size_t bw = W <= DN_STAGE_WORDS ? 128ull * W + 32ull * (W + 1u) : 0;
    return (bw + (size_t)DN_ROWS * W) * 8 + (size_t)(W + 1u) * DN_ROWS * 2 + 16;
}

uint32_t dense_rows_per_tile() { return DN_ROWS; }

hipError_t launch_dense_pairs(const DenseTile *tiles, uint32_t ntiles, const DenseGroup *groups, const unsigned long long *gdata,
                              const uint16_t *ext, uint32_t xs, uint32_t s, uint32_t wmax, uint32_t row_begin, uint32_t row_end,
                              uint64_t out_base, const uint32_t *inv, uint2 *out, hipStream_t stream)
{
    if (ntiles == 0) return hipSuccess;
    const size_t smem = dense_pairs_lds(wmax);
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(ntiles), dim3(128), smem, stream, tiles, groups, gdata, ext, xs, s, row_begin, row_end, out_base, inv, out);
        return hipGetLastError();
    };
    return wmax <= DN_STAGE_WORDS ? go(dn_pairs_kernel<false>) : go(dn_pairs_kernel<true>);
}

// ------------------------------------------------------------------------------------------------
// which rows belong together, whatever their order in the table (collections are not always listed by species): the
// LABEL of a row is the smallest row that holds one of its first four hashes -- rows of a clade agree on it with high
// probability (a member lacks all four of the clade's smallest values only rarely), two jumps label -> label of the
// label close the chains -- and the rows sorted by (label, row) are the order in which the index is built.  Any
// order would be correct; this one puts near-copies next to each other so that they form dense groups.
constexpr uint32_t CL_FIRST = 4;

__global__ __launch_bounds__(256) void cl_emit_kernel(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt, uint32_t n,
                                                      unsigned long long *key, uint32_t *row_out, uint32_t *label)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n * CL_FIRST) return;
    const uint32_t row = i / CL_FIRST, f = i % CL_FIRST;
    key[i] = f < cnt[row] ? (unsigned long long)hashes[(uint64_t)row * stride + f] : ~0ull;
    row_out[i] = row;
    if (f == 0) label[row] = row;
}

__global__ __launch_bounds__(256) void cl_minrow_kernel(const unsigned long long *key_sorted, const uint32_t *row_sorted, uint32_t m,
                                                        uint32_t *label)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const unsigned long long k = key_sorted[i];
    if (k == ~0ull) return;
    uint32_t lo = 0, hi = i;                               // first position holding k (the sort is stable: its row is the smallest)
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (key_sorted[mid] < k) lo = mid + 1; else hi = mid;
    }
    atomicMin(&label[row_sorted[i]], row_sorted[lo]);
}

__global__ __launch_bounds__(256) void cl_jump_kernel(const uint32_t *in, uint32_t *out, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = in[in[i]];
}

__global__ __launch_bounds__(256) void cl_order_keys_kernel(const uint32_t *label, uint32_t n, unsigned long long *key)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) key[i] = ((unsigned long long)label[i] << 32) | i;
}

__global__ __launch_bounds__(256) void cl_split_keys_kernel(const unsigned long long *key_sorted, uint32_t n, uint32_t *inv, uint32_t *label_sorted)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) {
        inv[i] = (uint32_t)(key_sorted[i] & 0xFFFFFFFFull);
        label_sorted[i] = (uint32_t)(key_sorted[i] >> 32);
    }
}

size_t dense_cluster_temp_bytes(uint32_t n)
{
    size_t a = 0, b = 0;
    rocprim::radix_sort_pairs(nullptr, a, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (const uint32_t *)nullptr,
                              (uint32_t *)nullptr, (size_t)n * CL_FIRST, 0u, 64u, (hipStream_t) nullptr);
    rocprim::radix_sort_keys(nullptr, b, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (size_t)n, 0u, 64u,
                             (hipStream_t) nullptr);
    return a > b ? a : b;
}

// scratch: key_a / key_b [4 n] u64, row_a / row_b [4 n] u32, lab_a / lab_b [n] u32.  Out: inv[n] (index row -> table row),
// label_sorted[n] (the label of every index row: equal labels = one cluster).
hipError_t dense_cluster_rows(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt, uint32_t n, void *temp, size_t temp_bytes,
                              unsigned long long *key_a, unsigned long long *key_b, uint32_t *row_a, uint32_t *row_b, uint32_t *lab_a,
                              uint32_t *lab_b, uint32_t *inv, uint32_t *label_sorted, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint32_t m = n * CL_FIRST;
    hipLaunchKernelGGL(cl_emit_kernel, dim3((m + 255u) / 256u), dim3(256), 0, stream, hashes, stride, cnt, n, key_a, row_a, lab_a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(temp, temp_bytes, (const unsigned long long *)key_a, key_b, (const uint32_t *)row_a, row_b, (size_t)m, 0u, 64u,
                                  stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cl_minrow_kernel, dim3((m + 255u) / 256u), dim3(256), 0, stream, (const unsigned long long *)key_b, (const uint32_t *)row_b, m,
                       lab_a);
    hipLaunchKernelGGL(cl_jump_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, (const uint32_t *)lab_a, lab_b, n);
    hipLaunchKernelGGL(cl_jump_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, (const uint32_t *)lab_b, lab_a, n);
    hipLaunchKernelGGL(cl_order_keys_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, (const uint32_t *)lab_a, n, key_a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_keys(temp, temp_bytes, (const unsigned long long *)key_a, key_b, (size_t)n, 0u, 64u, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cl_split_keys_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, (const unsigned long long *)key_b, n, inv, label_sorted);
    return hipGetLastError();
}

// the table in index order: out[a] = row inv[a] (whole rows, padding included)
__global__ __launch_bounds__(256) void cl_gather_rows_kernel(const uint64_t *hashes, uint64_t stride, const uint32_t *inv, uint64_t *out)
{
    const uint32_t a = blockIdx.x;
    const uint64_t *src = hashes + (uint64_t)inv[a] * stride;
    uint64_t *dst = out + (uint64_t)a * stride;
    for (uint64_t p = threadIdx.x; p < stride; p += 256u) dst[p] = src[p];
}

hipError_t launch_dense_gather_rows(const uint64_t *hashes, uint64_t stride, const uint32_t *inv, uint32_t n, uint64_t *out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(cl_gather_rows_kernel, dim3(n), dim3(256), 0, stream, hashes, stride, inv, out);
    return hipGetLastError();
}

}  // namespace mg
