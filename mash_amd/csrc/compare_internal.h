// compare_internal.h — launch interface between host_compare.cpp and the compare kernels (compare_sparse.hip,
// compare_merged.hip, compare.hip).
#pragma once
#ifdef MG_HIP_EMU                    // tools/hipemu: the kernels on host threads (tests/test_dense_emu.py)
#include "hipemu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace mg {

// Tile of the merged-rows kernel: up to 16 rows given EXPLICITLY (ascending row indices,
// 0xFFFFFFFF = unused slot) and a column range.  Rows of one tile need not be adjacent: the
// host groups rows of similar hash density (see run_compare), because one linear
// value -> bucket map per tile only spreads entries evenly when its rows are equally dense.
struct MergedTile {
    uint32_t rows[32];      // plain tiles list up to 16 rows, window tiles up to 32
    uint32_t col0, col1;
};

struct CompareArgs {
    const uint64_t *row_hashes;   // table whose rows sit in LDS (triangle: the table; rect: queries)
    const uint32_t *row_nhash;
    const uint64_t *col_hashes;   // table streamed through registers (triangle: same; rect: refs)
    const uint32_t *col_nhash;
    const uint32_t *row_pfx;      // u32 prefix images (value >> pfx_shr, saturated), same strides
    const uint32_t *col_pfx;
    const MergedTile *mtiles;     // merged kernel
    uint2 *out;                   // {numer, denom}
    uint64_t row_stride, col_stride;
    uint64_t row_pfx_stride, col_pfx_stride;   // strides of the padded prefix images
    uint64_t row_begin, row_end;  // rows handled by this launch
    uint64_t ncols;               // rect: number of refs
    uint64_t out_base;            // triangle: row_begin*(row_begin-1)/2
    uint32_t s;                   // sketch size used for the comparison
    uint32_t rows_per_tile;       // R
    uint32_t triangle;            // 1: only j < i, triangular output; 0: rect
    uint32_t unroll;              // probes in flight per wave (tuning knob; 0 = default)
    uint32_t pfx_shr;             // prefix shift shared by both tables
    uint32_t nbuckets;            // merged kernel: buckets of the tile table (set by its launcher)
    unsigned long long *dbg;      // tuning hook: per-tile {start, built, end} clocks (nullptr = off)
    // Value-window mode of the merged kernel (large sketches, see compare_merged.hip): one launch
    // handles the hashes whose prefix lies in [win_lo, win_hi); row_win / col_win give, per row and
    // window boundary, the index of the first hash at or above the boundary (nwin + 1 per row).
    const uint32_t *row_win;
    const uint32_t *col_win;
    uint32_t win, nwin;           // this launch's window, number of windows (0 = not windowed)
    uint32_t win_lo, win_hi;      // prefix range of the window (win_hi: exclusive, for the bucket scale)
    uint32_t win_ecap;            // table entries a windowed tile may hold
    uint8_t *win_mask;            // [tiles][16 waves][win_kmax]: live columns per batch, carried between launches
    uint32_t win_kmax;            // batches of 8 columns per wave and tile
    uint32_t xcd_remap;           // 1: XCD x takes the x-th contiguous eighth of the tile list
    uint32_t stage_pack;          // window mode: results staged as u16 pairs (s < 32768; set by the launcher)
};

// Tile engine ("merged rows", compare_merged.hip): one bucketed table for all rows of a tile.
bool compare_merged_supported(uint32_t s);
uint32_t compare_merged_rows(uint32_t s);
hipError_t launch_compare_merged(const CompareArgs &a, uint32_t ntiles, hipStream_t stream);
// windowed mode: rows per tile, entries per tile, and the per-row window offsets
uint32_t compare_window_rows(uint32_t s);     // rows a window tile may list (32, or 16 for s >= 32768)
uint32_t compare_window_entries();            // entries a window tile may hold
uint32_t compare_window_row_entries();        // entries of ONE row a window tile may hold (tag index field)
hipError_t launch_window_offsets(const uint32_t *pfx, uint64_t pfx_stride, const uint32_t *nhash, uint64_t n, uint32_t s,
                                 uint32_t nwin, uint32_t delta, uint32_t *out, hipStream_t stream);
// table max (u64 atomicMax over the last valid entry of every row) and prefix image
hipError_t launch_table_max(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t stride,
                            unsigned long long *out_max, hipStream_t stream);
// density class of every row: bit length of (largest hash / number of hashes), 0 for empty rows
hipError_t launch_row_classes(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t stride,
                              uint8_t *out, unsigned long long *last_out, hipStream_t stream);
uint64_t compare_pfx_stride(uint64_t s);          // row stride (u32 entries) of the padded prefix image
hipError_t launch_make_prefix(const uint64_t *hashes, const uint32_t *nhash, uint64_t n, uint64_t s,
                              uint64_t pfx_stride, uint32_t shr, uint32_t *out, hipStream_t stream);
// Generic kernel (any s): one wave per pair, binary search in global memory.
hipError_t launch_compare_generic(const CompareArgs &a, hipStream_t stream);

struct DenseGroup;
// Inverted-index ("sparse") engine (compare_sparse.hip): fill + discover + merge over an index of
// the column table built once per table (host_compare.cpp::table_sparse_index).
struct SparseArgs {
    const uint32_t *sorted_rows;   // column table: row of the entry at every sorted position (value major, rows ascending)
    // per entry of the ROW side (image layout, stride rs_row): [lo_img >> lo_shift, hi_img) = run of partner rows in
    // sorted_rows.  Triangle: lo_img = the code image (2 x group start, lo_shift 1), hi_img = the position image (the
    // entry's own sorted position: the rows BELOW it).  Rect: the located run of every query value (lo_shift 0).
    const uint32_t *lo_img;
    const uint32_t *hi_img;
    uint32_t lo_shift;
    const uint32_t *off;           // row side: compact entry offsets (triangle: the table's; rect: the queries' from q_begin)
    const uint32_t *row_img;       // row side code image (triangle: the table's; rect: query codes)
    const uint32_t *col_img;       // column side code image (codes 2 * start position of the value's group in the sorted index)
    const uint32_t *col_cnt_off;   // column side compact offsets (hash counts = differences)
    uint32_t rs_row, rs_col;       // row strides of the two images (multiples of 4, padded)
    uint32_t row_begin, row_end;   // rows handled (rect: 0 .. number of queries, relative to q_begin)
    uint32_t ncols;                // rect: rows of the reference table
    uint32_t triangle;
    uint32_t s;
    uint64_t out_base;             // triangle: row_begin (row_begin - 1) / 2
    uint2 *cand;                   // candidate pairs {row, col}
    uint64_t cand_cap;
    unsigned long long *counters;  // [0] candidates, [1] shared hashes of the rows handled, [2] candidate list overflowed
    uint2 *out;
    uint2 *res;                    // {common, denom} per candidate, in list order (scattered to `out` after the fill)
    unsigned long long *seg_base;  // per discover slot (slot = row_end - 1 - row): the row's segment of the list
    uint32_t *seg_cnt;
    uint32_t *chunk_inc;           // inclusive scan of the rows' merge work items
    // identical rows of the column table (nullptr: none): rep[row] = first row of its class; classes of two
    // rows and more: cls_of[representative] = class id (else 0xFFFFFFFF), rows cls_rows[cls_off[id] .. cls_off[id + 1]) ascending
    const uint32_t *rep;
    const uint32_t *cls_of;
    const uint32_t *cls_off;
    const uint32_t *cls_rows;
    const uint32_t *gend;          // at a group's start position: one past its last (a copy walks the whole run of a value)
    const uint32_t *order;         // rows of the launch in visiting order (nullptr: row_end - 1 - slot)
    const uint32_t *inv;           // index built on a permuted table: index row -> table row (nullptr: the same)
    // list jobs on a table with dense groups (compare_dense.hip): a grouped row's list ends with its pairs inside the group
    const uint32_t *dn_grp_of;     // [n] group of a row, 0xFFFFFFFF: none (nullptr: no groups)
    const DenseGroup *dn_groups;
};
size_t sparse_dup_temp_bytes(uint32_t n);
hipError_t launch_sparse_dup_suspects(const unsigned long long *dig, const uint32_t *cnt, uint32_t n, void *temp, size_t temp_bytes,
                                      unsigned long long *dig_sorted, uint32_t *rows_sorted, uint32_t *flags, uint32_t *nflag,
                                      hipStream_t stream);
size_t sparse_order_temp_bytes(uint32_t n);
hipError_t launch_sparse_row_order(const uint32_t *off, const uint32_t *code_img, const uint32_t *gend, const uint32_t *rep, uint32_t n,
                                   uint32_t rs, void *temp, size_t temp_bytes, unsigned long long *key_a, unsigned long long *key_b,
                                   uint32_t *order, hipStream_t stream);
size_t sparse_order_slice_temp_bytes(uint32_t n);
hipError_t launch_sparse_order_slice(const uint32_t *order, uint32_t n, uint32_t rb, uint32_t re, void *temp, size_t temp_bytes,
                                     uint32_t *out, uint32_t *count_out, hipStream_t stream);
hipError_t launch_sparse_class_pairs(uint2 *out, const uint32_t *cls_rows, const uint32_t *cls_first, const uint32_t *off,
                                     const uint32_t *rep, uint32_t members, uint32_t row_begin, uint32_t row_end, uint64_t out_base,
                                     const uint32_t *inv, hipStream_t stream);
hipError_t launch_sparse_row_digest(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt, uint32_t n,
                                    unsigned long long *digest, hipStream_t stream);
hipError_t launch_sparse_row_equal(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt, const uint2 *pairs, uint32_t npairs,
                                   uint32_t *equal, hipStream_t stream);
hipError_t launch_sparse_row_counts(const uint32_t *nhash, uint32_t n, uint32_t cap, uint32_t *cnt, hipStream_t stream);
// off[0 .. n]: exclusive prefix of cnt[inv[a]] (inv == nullptr: cnt[a]) -- no row kept out of the index
size_t sparse_offsets_temp_bytes(uint32_t n);
hipError_t launch_sparse_offsets(const uint32_t *cnt, const uint32_t *inv, uint32_t n, void *temp, uint32_t *off, hipStream_t stream);
size_t sparse_sort_temp_bytes(uint32_t E, uint32_t end_bit, uint32_t begin_bit);
uint32_t sparse_img_stride(uint32_t s);          // row stride of a code image
hipError_t sparse_build_index(const uint64_t *hashes, uint64_t stride, const uint32_t *off, uint32_t n, uint32_t E,
                              uint32_t rs, uint32_t end_bit, void *temp, size_t temp_bytes, uint64_t *keys_a,
                              uint32_t *idx_a, uint64_t *keys_sorted, uint32_t *idx_sorted, uint32_t *head, uint32_t *gs_of,
                              uint32_t *sorted_rows, uint32_t *gend, uint32_t *code_img, uint32_t *pos_img, void *stat_scratch,
                              uint32_t begin_bit, void *tie_scratch, unsigned long long *incidences, uint32_t *max_group, uint32_t *groups,
                              uint32_t *bad, uint32_t *tie_overflow, hipStream_t stream);
size_t sparse_stat_scratch_bytes();
uint32_t sparse_sort_begin_bit(uint32_t E, uint32_t end_bit, const char *forced_bits, bool all_bits);
size_t sparse_tie_scratch_bytes();
hipError_t launch_sparse_locate(const uint64_t *qhashes, uint64_t qstride, const uint32_t *qoff, uint32_t q_begin, uint32_t nq,
                                const uint64_t *keys_sorted, const uint32_t *gend, uint32_t E, uint32_t rs, uint32_t *qlo_img,
                                uint32_t *qhi_img, uint32_t *qcode_img, hipStream_t stream);
bool sparse_discover_supported(uint32_t ncols_max);
hipError_t launch_sparse_discover(const SparseArgs &a, bool count_only, hipStream_t stream);
hipError_t launch_sparse_merge(const SparseArgs &a, uint64_t expect, uint32_t cus, hipStream_t stream);
bool sparse_merge_rows_supported(uint32_t rs_row);
size_t sparse_scan_temp_bytes(uint32_t nrows);
hipError_t launch_sparse_merge_rows(const SparseArgs &a, uint64_t expect, uint32_t *chunks, void *temp, size_t temp_bytes,
                                    hipStream_t stream);
hipError_t launch_sparse_scatter(const SparseArgs &a, uint64_t expect, uint32_t cus, hipStream_t stream);
size_t sparse_gather_temp_bytes(uint32_t nrows);
hipError_t launch_sparse_gather_rows(const SparseArgs &a, uint32_t *cnt_by_row, uint32_t *row_base, void *temp, size_t temp_bytes, uint32_t row_add,
                                     uint2 *rc_out, uint2 *counts_out, hipStream_t stream);
size_t sparse_edges_temp_bytes(uint64_t K);
uint64_t sparse_edges_blocks(uint64_t K);
hipError_t launch_sparse_list_edges(const uint2 *rc, const uint2 *counts, uint64_t K, uint32_t *blk_cnt, uint32_t *blk_off, void *temp, size_t temp_bytes,
                                    uint4 *edges, unsigned long long *total, hipStream_t stream);
// the fill in chunks handed out by a counter: any number of launches (a slow one beside other work, a fast one to end it) share one job
uint64_t sparse_fill_chunks(uint64_t pairs);
constexpr uint32_t kFillStop = 0x80000000u;                 // the counter's value that ends every launch at its next chunk
hipError_t launch_sparse_fill_chunks(uint2 *out, uint64_t pairs, uint32_t numer, uint32_t denom, uint32_t blocks, uint32_t naps, uint32_t *ctr,
                                     hipStream_t stream, uint32_t threads = 256);
hipError_t launch_sparse_fill_value(uint2 *out, uint64_t pairs, uint32_t numer, uint32_t denom, uint32_t blocks_per_cu, uint32_t cus,
                                    hipStream_t stream);
hipError_t launch_sparse_merge_pack(const SparseArgs &a, uint64_t expect, uint32_t *chunks, void *temp, size_t temp_bytes, bool *used,
                                    hipStream_t stream);
hipError_t launch_sparse_fill_short(uint2 *out, const uint32_t *short_rows, const uint32_t *short_rcnt, uint32_t nshort_rows,
                                    const uint32_t *short_cols, const uint32_t *short_ccnt, uint32_t nshort_cols,
                                    uint32_t row_begin, uint32_t ncols, uint32_t triangle, uint64_t out_base, uint32_t s,
                                    const uint32_t *inv, hipStream_t stream);

// Dense group engine (compare_dense.hip): the pairs inside a group of near-identical consecutive rows as bit-mask
// arithmetic over the group's universe (the values at least two of its rows hold).
struct DenseGroup {
    uint32_t g0, g1;               // rows [g0, g1)
    uint32_t ustart, u;            // the group's universe: ulist[ustart .. ustart + u), ascending sorted positions (= codes / 2)
    uint32_t W;                    // (u >> 6) + 1 mask words per row
    uint32_t xrow0;                // index of row g0 among the grouped rows (extras of row r: ext + (xrow0 + r - g0) * xs)
    uint64_t data_off;             // first block of the group in gdata (u64 words), see dense_block_words
};
struct DenseTile { uint32_t group, row0, cblk; };
// u64 words of a block of 128 rows of a group with W mask words per row: the masks [W][128], then two tables of u16
// [W + 1][128] -- cx (extras before every word boundary | flag) and tot (ALL the row's values before it: mask bits + extras)
// -- then the extras' bit planes [W][4][128] (bit o of plane j of word w: bit j of the number of extras at offset o of the
// word), lane = row of the block everywhere: the 128 columns of a tile read whole lines
#ifdef __HIPCC__
__host__ __device__
#endif
static inline uint64_t dense_block_planes(uint32_t W) { return 128ull * W + 64ull * (W + 1u); }      // where the planes start
#ifdef __HIPCC__
__host__ __device__
#endif
static inline uint64_t dense_block_words(uint32_t W) { return dense_block_planes(W) + 512ull * W; }
// where a list job wants the pairs inside the groups: the list of row a (reference order: rows ascending, a row's pairs by
// column) ends with its partners inside its group, so pair (a, b) stands at row_base[a - row_first] + row_cnt[a - row_first] - (a - b)
struct DenseList {
    const uint32_t *row_base = nullptr, *row_cnt = nullptr;
    uint2 *rc = nullptr, *counts = nullptr;               // {row, col} and {common, denom} per list entry (rc == nullptr: the matrix)
    uint32_t row_first = 0;
};
hipError_t launch_dense_neighbors(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt, uint32_t n, uint8_t *link,
                                  hipStream_t stream);
size_t dense_universe_temp_bytes(uint32_t cap);
uint32_t dense_sublists();
hipError_t dense_find_leaders(const uint32_t *sorted_rows, const uint32_t *gs_of, const uint32_t *gend, const uint32_t *grp_of,
                              const DenseGroup *groups, uint32_t E, unsigned long long *key, uint32_t *val, uint32_t cap_sub,
                              unsigned long long *key_out, uint32_t *val_out, uint32_t *cnt, uint32_t *off, uint32_t *total,
                              hipStream_t stream);
hipError_t dense_join_leaders(const unsigned long long *key, const uint32_t *val, uint32_t cap_sub, unsigned long long *key_out, uint32_t *val_out,
                              const uint32_t *cnt, uint32_t *off, uint32_t *total, hipStream_t stream);
hipError_t dense_sort_universes(const unsigned long long *key, const uint32_t *val, uint32_t m, void *temp, size_t temp_bytes,
                                unsigned long long *key_sorted, uint32_t *ulist, uint32_t *upos, uint32_t *ustart, uint32_t *uend,
                                uint32_t group_bits, hipStream_t stream);
hipError_t launch_dense_encode(const uint32_t *off, const uint32_t *code_img, uint32_t *pos_img, uint32_t rs, const uint32_t *grp_of,
                               const DenseGroup *groups, const uint32_t *ulist, const uint32_t *upos, unsigned long long *gdata,
                               uint16_t *ext, uint32_t xs, uint32_t n, uint32_t wmax, hipStream_t stream, int ul_mode = -1);
size_t dense_pairs_lds(uint32_t W, uint32_t rows);
uint32_t dense_rows_per_tile(uint64_t wave_rows);
uint32_t dense_max_words();
hipError_t launch_dense_pairs(const DenseTile *tiles, uint32_t ntiles, uint32_t rows_per_tile, const DenseGroup *groups,
                              const unsigned long long *gdata, bool use_lists, const uint16_t *ext, uint32_t xs,
                              uint32_t s, uint32_t wmax, uint32_t row_begin, uint32_t row_end, uint64_t out_base, const uint32_t *inv, uint2 *out,
                              hipStream_t stream, const DenseList *list = nullptr);
size_t dense_cluster_temp_bytes(uint32_t n);
hipError_t dense_cluster_rows(const uint64_t *hashes, uint64_t stride, const uint32_t *cnt, uint32_t n, void *temp, size_t temp_bytes,
                              unsigned long long *key_a, unsigned long long *key_b, uint32_t *row_a, uint32_t *row_b, uint32_t *lab_a,
                              uint32_t *lab_b, uint32_t *inv, uint32_t *label_sorted, hipStream_t stream, uint32_t split = 0);
hipError_t launch_dense_gather_rows(const uint64_t *hashes, uint64_t stride, const uint32_t *inv, uint32_t n, uint64_t *out, hipStream_t stream);
// grp_of[n], lead_rows[4 n] (16-byte aligned) from the candidate groups (disjoint, ascending)
hipError_t launch_dense_group_rows(const DenseGroup *groups, uint32_t ng, uint32_t n, uint32_t *grp_of, uint32_t *lead_rows, hipStream_t stream);

// Join engine (compare_join.hip): collections in the middle of the similarity range -- every pair shares a tenth to a half
// of its values, none is a near-copy.  Rows in blocks of 64; per block the list of its entries in value order, as groups of
// equal values; a tile (block of rows, block of columns) intersects two lists and counts, per pair and in ascending order,
// the common values whose rank in the pair's union is below s.  It writes every pair of its tiles: no fill, no discovery.
struct JoinSide {
    const uint2 *grp = nullptr;        // per group {value id (code >> 1), first entry}; one record more at the end
    const uint32_t *ent = nullptr;     // per entry (position in its row << 8) | row in its block; rows ascend inside a group
    const uint32_t *goff = nullptr;    // [blocks + 1] first group of a block
    const uint32_t *gend = nullptr;    // [blocks] one past the block's last group that takes part (behind it: the entries left out)
    const uint32_t *thr = nullptr;     // [blocks * 16] largest value id at position k s / 16 over the block's rows (nullptr: no early stop)
};
struct JoinArgs {
    JoinSide rows, cols;               // triangle: the same lists
    const uint32_t *row_cnt_off, *col_cnt_off;   // compact entry offsets: hash counts are differences
    const uint32_t *rep, *col_rep;     // the index row whose entries speak for a row / a column of the lists (nullptr: itself) -- a
                                       // copy's representative, and the lists may stand on the rows in an order of their own
    const uint32_t *inv;               // triangle: row of the lists -> row of the TABLE (nullptr: the same)
    uint2 *out;
    uint64_t out_base;                 // triangle: row_begin (row_begin - 1) / 2
    uint64_t ntiles;
    uint32_t ncols;                    // columns (triangle: rows of the table)
    uint32_t row_begin, row_end;       // rows of the job
    uint32_t bi0;                      // first block of rows of the job
    uint32_t ncb;                      // rect: blocks of columns
    uint32_t triangle, s;
    uint32_t tiles_per_wg = 1;         // 1 or 4 waves per workgroup (a wave takes a tile)
};
size_t join_order_temp_bytes(uint32_t nrows);
hipError_t join_order_rows(const uint32_t *img, uint32_t rs, const uint32_t *cnt_off, const uint32_t *rep, const uint32_t *inv, const uint32_t *gend,
                           const uint32_t *sorted_rows, uint32_t nrows, void *temp, size_t temp_bytes, uint32_t *lab, unsigned long long *key_a,
                           unsigned long long *key_b, uint32_t *val_a, uint32_t *perm, uint32_t *src, uint32_t *map, hipStream_t stream, uint32_t split = 0);
uint32_t join_block_rows();
uint32_t join_levels();
size_t join_build_temp_bytes(uint64_t slots);
hipError_t join_build_lists(const uint32_t *img, uint32_t rs, const uint32_t *cnt_off, const uint32_t *rep, uint32_t nrows, uint32_t s, uint32_t E,
                            bool only_shared, void *temp, size_t temp_bytes, unsigned long long *key_a, unsigned long long *key_b, uint32_t *val_a,
                            uint32_t *val_b, uint2 *grp, uint32_t *goff, uint32_t *gend, uint32_t *thr, const uint32_t **ent_out, hipStream_t stream);
hipError_t launch_join_levels(const uint32_t *img, uint32_t rs, const uint32_t *cnt_off, const uint32_t *rep, uint32_t nrows, uint32_t s,
                              uint32_t *thr, hipStream_t stream);
hipError_t launch_join_shared(const uint32_t *lo_img, const uint32_t *hi_img, uint32_t lo_shift, uint32_t rs, const uint32_t *cnt_off,
                              uint32_t row_begin, uint32_t row_end, unsigned long long *sum, hipStream_t stream);
hipError_t launch_join_tiles(const JoinArgs &a, hipStream_t stream);

// Distance filter + ordered compaction (see filter_pass_kernel).  `counts` holds
// `pairs` entries in the layout the compare kernels write, starting at row
// `first_row` (triangle row / query index).  Survivors with rank in
// [win_lo, win_lo + win_n) are written to edges[rank - win_lo].
struct FilterArgs {
    const uint2 *counts;
    const uint32_t *min_numer;    // [s+1] smallest numer passing the distance filter, per denom
    uint32_t *seg_count;          // [filter_segments(pairs)]
    unsigned long long *seg_off;  // [filter_segments(pairs)] exclusive scan of seg_count
    uint4 *edges;                 // {row, col, numer, denom}
    uint64_t pairs;
    uint64_t first_row;
    uint64_t ncols;               // rect: number of refs
    uint64_t win_lo, win_n;
    uint32_t s;
    uint32_t triangle;
};
uint64_t filter_segments(uint64_t pairs);
hipError_t launch_filter_count(const FilterArgs &a, unsigned long long *total, hipStream_t stream);
hipError_t launch_filter_write(const FilterArgs &a, hipStream_t stream);

}  // namespace mg
