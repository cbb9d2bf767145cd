/*
 * mashgpu.h — C ABI of the MI355X-native Mash hot path (libmashgpu.so).
 *
 * This is the drop-in boundary for the ONE data-parallel path of marbl/Mash:
 * k-mer hashing + bottom-s selection (sketching) and the early-terminating
 * sorted-merge Jaccard / Mash distance (comparing).  The reference has no
 * FFI/plugin API; its seams are the worker function pointers it hands to
 * ThreadPool<In,Out>.  Each entry point below names the
 * reference interface it replaces (file:line under /root/reference/src/mash).
 * Per-record / per-pair callbacks are far too fine for a GPU, so the ABI works
 * at batch granularity: plain pointers and sizes, caller-owned buffers, opaque
 * handles for device-resident state, `int` status returns (0 = ok, <0 = error;
 * text via mg_last_error).  No torch types, no exit(), no global state.
 *
 * Pointer conventions: `*_host` entry points take host pointers and stage
 * through HBM themselves; `*_dev` entry points take DEVICE pointers (e.g.
 * torch tensors' data_ptr()) and run on the context's stream without copies.
 */
#ifndef MASHGPU_H
#define MASHGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_OK                 0
#define MG_ERR_INVALID       -1   /* bad argument */
#define MG_ERR_UNSUPPORTED   -2   /* parameter combination outside the device path */
#define MG_ERR_HIP           -3   /* HIP runtime error (see mg_last_error) */
#define MG_ERR_NOMEM         -4

/* Byte that separates records inside one sketch's byte range: k-mers never
 * span records (Sketch.cpp:1200-1270 calls addMinHashes once per record).
 * Any byte outside the alphabet works; this one is the convention. */
#define MG_RECORD_SEP  0x0A
/* Padding value of unused hash slots in dense sketch tables. */
#define MG_HASH_PAD    0xFFFFFFFFFFFFFFFFull

typedef struct mg_ctx   mg_ctx;     /* one per process+GPU: device, stream, scratch */
typedef struct mg_table mg_table;   /* device-resident dense sketch table */

/* Mirrors the fields of Sketch::Parameters (Sketch.h:86-105) that change results. */
typedef struct mg_params {
    int32_t  kmer_size;        /* kmerSize, 1..32                     (Command.cpp:168) */
    uint32_t seed;             /* seed, default 42                    (Command.cpp:178) */
    uint64_t sketch_size;      /* minHashesPerWindow, default 1000               */
    uint32_t alphabet_size;    /* alphabetSize                                   */
    uint8_t  alphabet[256];    /* alphabet[] (1 = member)                        */
    uint8_t  preserve_case;    /* preserveCase (-Z)                              */
    uint8_t  use64;            /* use64 = alphabetSize^k > 2^32  (Sketch.cpp:1136) */
    uint8_t  noncanonical;     /* noncanonical (-n, forced by -a / -z)           */
    uint8_t  counts;           /* counts (multiplicities requested)              */
    uint32_t min_copies;       /* minCov (-m, reads mode; Sketch.cpp:1156): a hash enters the sketch at
                                * its m-th occurrence.  0 / 1 = every k-mer (mg_params_init sets 1). */
    double   target_cov;       /* targetCov (-c, reads mode; Sketch.cpp:1258), 0 = off: only
                                * mg_sketch_reads_host honours it (mg_params_init sets 0). */
    uint64_t bloom_bytes;      /* memoryBound (-b, reads mode; Sketch.h:101, MinHashHeap.cpp:19-41,78-94),
                                * 0 = off (mg_params_init): a Bloom filter of this many bytes stands in
                                * front of the heap -- a hash enters the sketch, with count 2, when its
                                * bit is found set, else it sets the bit.  Order-dependent by design
                                * (aliases); reproduced exactly by mg_sketch_reads_host / mg_reads_*,
                                * refused elsewhere.  Geometry: one hash function over bloom_bytes * 8
                                * bits, what x86-64 builds of the reference use. */
} mg_params;

/* {numer, denom} of one pair: what the merge loop of compareSketches produces
 * (CommandDistance.cpp:347-385).  8 bytes per pair. */
typedef struct mg_counts { uint32_t numer, denom; } mg_counts;

/* Full PairOutput (CommandDistance.h:63-70). `pass` semantics as the reference:
 * when the distance filter rejects, only `pass` (=0) is meaningful. */
typedef struct mg_pair {
    uint32_t numer, denom;
    double   distance;
    double   p_value;
    uint8_t  pass;
    uint8_t  _pad[7];
} mg_pair;

/* ---- context --------------------------------------------------------------
 * (Tuning knobs are environment variables read per call -- about a microsecond -- so that tests
 *  can switch engines on a live context; none is needed in normal use) */
int         mg_device_count(void);               /* visible GPUs (0 when there is none) */
int         mg_ctx_create(int device, mg_ctx **out);
void        mg_ctx_destroy(mg_ctx *ctx);
const char *mg_last_error(mg_ctx *ctx);      /* ctx may be NULL: last create error */
/* Run on an existing hipStream_t (e.g. torch's current stream).  NULL (which is also what the legacy
 * default stream's handle is) = the context's own stream: a blocking stream, i.e. implicitly ordered
 * with work on the legacy default stream, as every hipStreamDefault stream is. */
int         mg_ctx_set_stream(mg_ctx *ctx, void *hip_stream);
int         mg_ctx_synchronize(mg_ctx *ctx);
/* Entry points lock their context: any number of host threads may drive one context (their
 * calls run one at a time, in lock order, on the context's stream).  With async on, the compare
 * *_dev entry points return as soon as their work is queued on that stream -- tile lists live in
 * a ring of pinned slots, scratch goes back to the context in stream order, no call ends in a
 * stream synchronisation once the table's derived data is cached; mg_ctx_synchronize (or work
 * queued behind on the same stream) completes them, and an error of a queued kernel is reported
 * by the next synchronising call.  Default off: *_dev calls return with their output complete. */
int         mg_ctx_set_async(mg_ctx *ctx, int on);
/* Tuning and test knobs of ONE context (round 3 review: the library read some thirty MASHGPU_* environment
 * variables).  name = the knob's name ("MASHGPU_COMPARE_KERNEL", "MASHGPU_COMPARE_DENSE", ... -- DESIGN.md section 7
 * lists them), value = what the environment variable would hold; NULL removes the setting.  A knob that is not set
 * on the context is looked up in the environment under the same name, so existing scripts keep working.  Two
 * knobs inside kernel launchers (MASHGPU_SPARSE_PACK_MIN, MASHGPU_SPARSE_MERGE_WINDOWS) and MASHGPU_COMM_FORCE_RCCL
 * remain process-wide. */
int         mg_ctx_set_option(mg_ctx *ctx, const char *name, const char *value);
/* Device blocks that finished calls and freed / invalidated tables handed back are kept by the context for the
 * next call (small scratch: 256 MiB; large blocks -- the inverted index of a table, candidate lists -- up to
 * 40 % of the device's memory, so that the next table of the same shape pays no hipMalloc).  mg_ctx_trim waits for the context's
 * stream and returns all of them to the driver; they are also dropped whenever an allocation fails. */
int         mg_ctx_trim(mg_ctx *ctx);
/* Number of CUs of the device (for callers sizing work). */
int         mg_ctx_cu_count(mg_ctx *ctx);

/* setAlphabetFromString + sketchParameterSetup core (Sketch.cpp:1108-1137,
 * sketchParameterSetup.cpp:15-105): fills alphabet[], alphabet_size, use64. */
int mg_params_init(mg_params *p, int kmer_size, uint64_t sketch_size, uint32_t seed,
                   const char *alphabet, int noncanonical, int preserve_case);

/* ---- sketching ------------------------------------------------------------
 * Replaces sketchFile / sketchSequence / addMinHashes / MinHashHeap::tryInsert /
 * setMinHashesForReference (Sketch.cpp:1147-1365, :512-583, :1139-1145;
 * MinHashHeap.cpp:68-145; HashSet.cpp:78-118) for minCov==1, no Bloom filter.
 *
 * bases[nbases]: all input bytes. Sketch i covers bytes
 * [sketch_off[i], sketch_off[i+1]); inside it records are separated by
 * MG_RECORD_SEP (concatenated mode = many records in one range; `-i` = one
 * record per range).  Bytes are exactly what kseq hands to addMinHashes (any
 * case; uppercasing happens on device unless preserve_case).  Bytes >= 0x80
 * are invalid bases (the reference's behaviour there is undefined).
 *
 * Outputs (row i = sketch i): hashes_out[nsketch * sketch_size] ascending,
 * distinct, padded with MG_HASH_PAD (32-bit hashes are zero-extended);
 * nhash_out[nsketch] = valid entries; counts_out (nullable) multiplicities.
 */
int mg_sketch_host(mg_ctx *ctx, const mg_params *p,
                   const uint8_t *bases, uint64_t nbases,
                   const uint64_t *sketch_off, uint64_t nsketch,
                   uint64_t *hashes_out, uint32_t *nhash_out, uint32_t *counts_out);
int mg_sketch_dev(mg_ctx *ctx, const mg_params *p,
                  const uint8_t *bases_dev, uint64_t nbases,
                  const uint64_t *sketch_off_host, uint64_t nsketch,
                  uint64_t *hashes_out_dev, uint32_t *nhash_out_dev, uint32_t *counts_out_dev);

/* Packed nucleotide input (BASELINE.json north_star: "packed bases"): the same call as mg_sketch_host / mg_sketch_dev
 * for the ACGT alphabet, with the bases handed over at 3 bits each instead of 8.
 *
 *   packed[(nbases + 3) / 4]        two bits per base, base i in bits 2 (i % 4) .. of byte i / 4,
 *                                   code = (ASCII >> 1) & 3:  A 0, C 1, T 2, G 3 (either case)
 *   invalid_mask[(nbases + 7) / 8]  bit i % 8 of byte i / 8 set: base i is none of ACGT (N, IUPAC codes,
 *                                   MG_RECORD_SEP between two records, lower case under preserve_case); its code
 *                                   bits mean nothing.  NULL: no such base in the whole input.
 *
 * That is all addMinHashes reads of a nucleotide sequence (Sketch.cpp:512-583: upper-cased unless preserveCase, a
 * k-mer over a character outside the alphabet is skipped, the hash runs over the k-mer's characters), so the
 * results are those of mg_sketch_host on the unpacked bytes -- same hashes, same counts.  sketch_off[] counts
 * BASES, as there, and need not be multiples of four.  MG_ERR_UNSUPPORTED for other alphabets.
 *
 * mg_pack_bases is the host side (what a parse thread runs over the bytes kseq hands it, kseq.h:171-208): no
 * context, no device, thread-safe; disjoint ranges whose starts are multiples of 8 bases can be packed by
 * different threads into the same arrays.  *ninvalid_out (nullable) = bits set in the mask.
 * mg_sketch_host_packed copies the input to the device in pieces of whole sketches while the previous piece is
 * being sketched; mg_sketch_dev_packed takes packed input that already is in device memory (16-byte aligned,
 * readable 8 bytes past the end of both arrays). */
uint64_t mg_packed_bytes(uint64_t nbases);
uint64_t mg_packed_mask_bytes(uint64_t nbases);
int mg_pack_bases(const uint8_t *ascii, uint64_t nbases, int preserve_case, uint8_t *packed, uint8_t *invalid_mask,
                  uint64_t *ninvalid_out);
int mg_sketch_host_packed(mg_ctx *ctx, const mg_params *p, const uint8_t *packed, const uint8_t *invalid_mask,
                          uint64_t nbases, const uint64_t *sketch_off, uint64_t nsketch,
                          uint64_t *hashes_out, uint32_t *nhash_out, uint32_t *counts_out);
int mg_sketch_dev_packed(mg_ctx *ctx, const mg_params *p, const uint8_t *packed_dev, const uint8_t *invalid_mask_dev,
                         uint64_t nbases, const uint64_t *sketch_off_host, uint64_t nsketch,
                         uint64_t *hashes_out_dev, uint32_t *nhash_out_dev, uint32_t *counts_out_dev);

/* Streamed ingest (replaces the reader side of sketchFile's overlap of parsing and sketching,
 * ThreadPool.hxx:127-167 + kseq.h:171-208 feeding addMinHashes): the caller hands over bytes as it
 * parses them -- no concatenated batch on the host.  mg_sketch_add appends to the CURRENT sketch's
 * byte range (records separated by MG_RECORD_SEP by the caller); mg_sketch_end_sketch closes it.
 * The bytes are packed into two pinned staging buffers and copied to the device on a copy stream
 * while the caller goes on parsing.  mg_sketch_finish sketches everything closed so far: row i of
 * hashes_out[n * sketch_size] / nhash_out[n] / counts_out (nullable) = the i-th closed sketch,
 * n = mg_sketch_pending(); the session is then empty and can be filled again.  Results are those of
 * mg_sketch_host on the concatenation.  One thread at a time per session.
 *
 * Zero-copy form of mg_sketch_add for callers that parse on several threads (the reference's -p workers,
 * Sketch.cpp:211,354): mg_sketch_stage lends a window of `len` contiguous bytes of the pinned staging
 * buffer at the stream's current end (len <= mg_sketch_stage_capacity(); the buffer in use is sent first if
 * the window does not fit behind its contents); the caller fills it -- from as many threads as it likes --
 * and mg_sketch_commit(n <= len) appends its first n bytes to the stream (several commits may consume one
 * window piece by piece, with mg_sketch_end_sketch between them).  No other session call between stage
 * and the last commit of its window. */
typedef struct mg_sketch_session mg_sketch_session;
int      mg_sketch_begin(mg_ctx *ctx, const mg_params *p, mg_sketch_session **out);
int      mg_sketch_add(mg_sketch_session *ss, const uint8_t *bytes, uint64_t len);
uint64_t mg_sketch_stage_capacity(const mg_sketch_session *ss);
int      mg_sketch_stage(mg_sketch_session *ss, uint64_t len, uint8_t **window);
int      mg_sketch_commit(mg_sketch_session *ss, uint64_t len);
int      mg_sketch_end_sketch(mg_sketch_session *ss);
uint64_t mg_sketch_pending(const mg_sketch_session *ss);
int      mg_sketch_finish(mg_sketch_session *ss, uint64_t *hashes_out, uint32_t *nhash_out, uint32_t *counts_out);
void     mg_sketch_session_free(mg_sketch_session *ss);

/* Reads mode with the early stop of `mash sketch -r -c <cov>` (Sketch.cpp:1200-1270): ONE sketch
 * over all records of `bases` (separated by MG_RECORD_SEP, in the order the reference reads
 * them); after every record the reference stops once the heap's average multiplicity
 * (estimateMultiplicity, MinHashHeap.h:44) has reached p->target_cov.  That is a property of the
 * sequential heap; it is reproduced exactly: the device emits every k-mer hash that can still
 * change the heap (those below its current top), the host replays MinHashHeap::tryInsert
 * (incl. min_copies) over that thinned stream.  hashes_out[sketch_size], counts_out[sketch_size]
 * (nullable), *records_used_out = records (>= k long) consumed -- the "Reads used" line
 * (Sketch.cpp:1324-1327).  With target_cov == 0 this is mg_sketch_host for one sketch.
 * p->bloom_bytes > 0 (`-b`): the replayed heap has the reference's Bloom filter in front
 * (MinHashHeap.cpp:78-94), with or without target_cov; until the sketch is full every k-mer
 * hash is such an event, afterwards only those below its top. */
int mg_sketch_reads_host(mg_ctx *ctx, const mg_params *p, const uint8_t *bases, uint64_t nbases,
                         uint64_t *hashes_out, uint32_t *nhash_out, uint32_t *counts_out,
                         uint64_t *records_used_out);

/* The same as a session: chunks of WHOLE records in reading order (any chunk size); the heap lives on
 * the host between chunks, the device holds one chunk at a time, and *stopped_out turns 1 with the
 * chunk in which the target coverage is reached -- the caller stops reading its files there, as the
 * reference's reader loop does (Sketch.cpp:1258).  Results are those of mg_sketch_reads_host on
 * the concatenation of the chunks.  Every reads option goes through it -- also plain -r / -m (neither
 * target_cov nor bloom_bytes): host and device memory are then bounded by one chunk, as the reference's
 * are by its heap (Sketch.cpp:1196-1270), where mg_sketch_host over the whole read set needs it in HBM;
 * the result, incl. the order-dependent multiplicity of the largest kept hash under min_copies > 1
 * (MinHashHeap.cpp:96-144), is the same. */
typedef struct mg_reads_session mg_reads_session;
int  mg_reads_begin(mg_ctx *ctx, const mg_params *p, mg_reads_session **out);
int  mg_reads_add_host(mg_reads_session *rs, const uint8_t *bases, uint64_t nbases, int *stopped_out);
int  mg_reads_finish(mg_reads_session *rs, uint64_t *hashes_out, uint32_t *nhash_out, uint32_t *counts_out,
                     uint64_t *records_used_out);
/* an empty heap again (same parameters): the next read set through the same session -- its device buffers
 * (128 MiB of event space) are allocated once, not per input file (`mash dist -r ref.msh a.fq b.fq ...`) */
int  mg_reads_reset(mg_reads_session *rs);
void mg_reads_free(mg_reads_session *rs);

/* ---- sketch tables ----------------------------------------------------------
 * Dense replacement for vector<Sketch::Reference> (Sketch.h:131-139):
 * hashes[n * s] row-major ascending + nhash[n] + lengths[n] (Reference::length). */
int  mg_table_upload(mg_ctx *ctx, const uint64_t *hashes, const uint32_t *nhash,
                     const uint64_t *lengths, uint64_t n, uint64_t s, mg_table **out);
/* Adopt device buffers without copying.  They must outlive the table and must not change
 * while it exists: the compare path caches derived data per table (row maxima, density classes,
 * 32-bit prefix images). */
int  mg_table_wrap_dev(mg_ctx *ctx, const uint64_t *hashes_dev, const uint32_t *nhash_dev,
                       const uint64_t *lengths_dev, uint64_t n, uint64_t s, mg_table **out);
void mg_table_free(mg_table *t);
/* The contents of the table's buffers changed (a wrapped buffer was refilled, an uploaded one written to): drop
 * everything derived from them -- row maxima, density classes, prefix images, window offsets, the inverted index
 * and its plans.  The next compare call rebuilds what it needs, as the first call on a fresh table does (the
 * reference pays that per command: it re-reads and re-walks its sketches for every `mash triangle`,
 * CommandTriangle.cpp:101-139).  Without it a table whose buffers changed is answered from a stale index; the
 * candidate-count check of a pass catches most such cases ("the table changed since its index was built"), not all. */
int  mg_table_invalidate(mg_table *t);
uint64_t mg_table_rows(const mg_table *t);
uint64_t mg_table_sketch_size(const mg_table *t);

/* ---- comparing --------------------------------------------------------------
 * Merge counts of compareSketches (CommandDistance.cpp:336-385).
 *
 * Triangle (replaces compare(TriangleInput*), CommandTriangle.cpp:200-214):
 * rows [row_begin,row_end) of the lower triangle, output in reference order:
 * for i in rows, for j in [0,i): out[i*(i-1)/2 + j - row_begin*(row_begin-1)/2].
 *
 * Rect (replaces compare(CompareInput*), CommandDistance.cpp:306-334):
 * queries [q_begin,q_end) x all refs, query-major: out[(q-q_begin)*nref + r].
 * sketch_size = min of the two tables' sizes (CommandDistance.cpp:313-315).
 */
int mg_compare_tri_dev(mg_ctx *ctx, const mg_table *t, uint64_t row_begin, uint64_t row_end,
                       mg_counts *out_dev);
int mg_compare_tri_host(mg_ctx *ctx, const mg_table *t, uint64_t row_begin, uint64_t row_end,
                        mg_counts *out_host);
int mg_compare_rect_dev(mg_ctx *ctx, const mg_table *ref, const mg_table *qry,
                        uint64_t q_begin, uint64_t q_end, mg_counts *out_dev);
int mg_compare_rect_host(mg_ctx *ctx, const mg_table *ref, const mg_table *qry,
                         uint64_t q_begin, uint64_t q_end, mg_counts *out_host);

/* Thresholded all-pairs ("edge list": `mash dist -d`, `mash triangle -E -d`).
 * Same pairs and order as the calls above, but the distance filter of
 * compareSketches (CommandDistance.cpp:409-412, `distance > maxDistance` rejects)
 * and the compaction run on the device: only surviving pairs cross PCIe.
 * The filter is exact: the threshold is converted on the host (same libm as
 * mg_finish_*) into the smallest passing numer per denom and the device compares
 * integers.  out_host receives the survivors in reference order (row-major;
 * row = triangle row i / query index, col = j / reference index); *count_out is
 * their total number.  If it exceeds `capacity`, MG_ERR_NOMEM is returned, the
 * content of out_host is unspecified and the caller retries with *count_out
 * entries.  The p-value filter (:419-422) stays with the caller (mg_p_value). */
typedef struct mg_edge { uint32_t row, col, numer, denom; } mg_edge;
int mg_compare_tri_filter_host(mg_ctx *ctx, const mg_table *t, uint64_t row_begin, uint64_t row_end,
                               int kmer_size, double max_distance, mg_edge *out_host,
                               uint64_t capacity, uint64_t *count_out);
int mg_compare_rect_filter_host(mg_ctx *ctx, const mg_table *ref, const mg_table *qry,
                                uint64_t q_begin, uint64_t q_end, int kmer_size, double max_distance,
                                mg_edge *out_host, uint64_t capacity, uint64_t *count_out);

/* The WHOLE matrix without moving it: a pair that shares no hash among its first s union elements is
 * {0, min(s, |A| + |B|)} (|X| = min(nhash, s): the merge of compareSketches, CommandDistance.cpp:347-385,
 * never takes its equal branch) -- and in a collection nearly every pair is such a pair; `mash triangle`
 * prints mostly that constant (CommandTriangle.cpp:159-198).  These calls return the EXCEPTIONS: every
 * pair with numer >= 1 as {row, col, numer, denom}, in reference order; together with the rule and the
 * tables' nhash the caller has every {numer, denom} of the job -- 80 MB instead of 40 GB for 100 000
 * sketches (mg_expand_tri_sparse writes the dense form out of it).  capacity / *count_out / MG_ERR_NOMEM
 * as for the filter calls above. */
int mg_compare_tri_sparse_host(mg_ctx *ctx, const mg_table *t, uint64_t row_begin, uint64_t row_end,
                               mg_edge *out_host, uint64_t capacity, uint64_t *count_out);
int mg_compare_rect_sparse_host(mg_ctx *ctx, const mg_table *ref, const mg_table *qry,
                                uint64_t q_begin, uint64_t q_end, mg_edge *out_host, uint64_t capacity,
                                uint64_t *count_out);
/* Host helper (no device): the dense triangle mg_compare_tri_host would have returned, from the rule and
 * the exceptions.  nhash: the table's hash counts (host), s: its sketch size. */
int mg_expand_tri_sparse(const mg_edge *edges, uint64_t count, const uint32_t *nhash, uint64_t sketch_size,
                         uint64_t row_begin, uint64_t row_end, mg_counts *out_host);

/* Distance + p-value + filters for pairs already counted (the tail of
 * compareSketches, CommandDistance.cpp:387-424, and pValue, :427-448).
 * Host arithmetic (glibc log, the same libm the reference links), so distances
 * are bit-identical to the reference build on this box.
 * len_ref/len_qry: Reference::length of the two sketches of each pair are taken
 * from the tables by index: triangle pairs use (i,j) in reference order.
 * max_distance / max_p_value < 0 disable the filters (the CLI passes 1 / 1).
 * Batches of 2^20 pairs and more are split over up to 16 host threads (the reference runs
 * compare jobs on its -p pool); every pair is independent, the output does not depend on it. */
int mg_finish_tri_host(const mg_counts *counts, const uint64_t *lengths, uint64_t row_begin,
                       uint64_t row_end, int kmer_size, double kmer_space,
                       double max_distance, double max_p_value, mg_pair *out);
int mg_finish_rect_host(const mg_counts *counts, const uint64_t *len_ref, uint64_t nref,
                        const uint64_t *len_qry, uint64_t nqry, int kmer_size, double kmer_space,
                        double max_distance, double max_p_value, mg_pair *out);
/* The same tail ON THE DEVICE (finish.hip): distances from a table the host builds with its libm
 * (one row per denominator that occurs), p-values by the exact double-double binomial tail of
 * pvalue.h -- both bit-identical to mg_finish_*_host -- and both filters before anything crosses
 * PCIe.  `t` / `ref`,`qry` supply Reference::length by index (the tables must carry lengths).
 * max_distance < 0 or >= 1 and max_p_value < 0 or >= 1 disable the respective filter.
 *   mg_finish_*_dev          counts (device, layout of mg_compare_*_dev) -> mg_pair (device)
 *   mg_compare_*_pairs_host  compare + finish, every pair, 32 B per pair to the host
 *   mg_compare_*_results_host compare + both filters + ordered compaction: survivors only, in
 *                            reference order; capacity / *count_out / MG_ERR_NOMEM as for
 *                            mg_compare_*_filter_host.  This is what `mash dist -d/-v` and
 *                            `mash triangle -E` print (CommandDistance.cpp:247-304). */
typedef struct mg_result { uint32_t row, col, numer, denom; double distance, p_value; } mg_result;
int mg_finish_tri_dev(mg_ctx *ctx, const mg_table *t, const mg_counts *counts_dev, uint64_t row_begin,
                      uint64_t row_end, int kmer_size, double kmer_space, double max_distance,
                      double max_p_value, mg_pair *out_dev);
int mg_finish_rect_dev(mg_ctx *ctx, const mg_table *ref, const mg_table *qry, const mg_counts *counts_dev,
                       uint64_t q_begin, uint64_t q_end, int kmer_size, double kmer_space,
                       double max_distance, double max_p_value, mg_pair *out_dev);
int mg_compare_tri_pairs_host(mg_ctx *ctx, const mg_table *t, uint64_t row_begin, uint64_t row_end,
                              int kmer_size, double kmer_space, double max_distance, double max_p_value,
                              mg_pair *out_host);
int mg_compare_rect_pairs_host(mg_ctx *ctx, const mg_table *ref, const mg_table *qry, uint64_t q_begin,
                               uint64_t q_end, int kmer_size, double kmer_space, double max_distance,
                               double max_p_value, mg_pair *out_host);
int mg_compare_tri_results_host(mg_ctx *ctx, const mg_table *t, uint64_t row_begin, uint64_t row_end,
                                int kmer_size, double kmer_space, double max_distance, double max_p_value,
                                mg_result *out_host, uint64_t capacity, uint64_t *count_out);
int mg_compare_rect_results_host(mg_ctx *ctx, const mg_table *ref, const mg_table *qry, uint64_t q_begin,
                                 uint64_t q_end, int kmer_size, double kmer_space, double max_distance,
                                 double max_p_value, mg_result *out_host, uint64_t capacity,
                                 uint64_t *count_out);
/* Scalar helpers (same arithmetic as the bulk calls). */
double mg_distance(uint32_t numer, uint32_t denom, int kmer_size);
double mg_p_value(uint64_t x, uint64_t len_ref, uint64_t len_qry, double kmer_space,
                  uint64_t sketch_size);

/* ---- several GPUs ---------------------------------------------------------------------
 * Replaces the fan-out loops of the reference's compare commands (one ThreadPool job per row,
 * CommandTriangle.cpp:129-139; per block of pairs, CommandDistance.cpp:195-232): every pair is
 * independent, so rows (triangle) / queries (dist) are cut into one block per GPU against a
 * sketch table resident on EVERY GPU.  The one exchange is the broadcast of that table from GPU 0
 * (RCCL, xGMI); the compare data path has no collective and each GPU writes its own slice of
 * the reference-ordered output.
 *
 * local communicator : one process drives all GPUs (the `mash` CLI) -- a context per device,
 *                      ncclCommInitAll; mg_dtable = a table replicated on every device;
 *                      mg_compare_*_sharded_host = the single-GPU call of the same name, rows
 *                      split into equal-AREA blocks (triangle) or evenly (rect), one host thread
 *                      per GPU, output byte-identical to the single-GPU call.
 * rank communicator  : one process per GPU (bench.py under torchrun) -- ncclCommInitRank on
 *                      128 bytes from mg_comm_unique_id that the caller hands to every rank;
 *                      mg_table_broadcast, then each rank compares the rows mg_shard_tri_rows
 *                      gives it with the ordinary mg_compare_*_dev.
 * A device list that repeats a device (tests on a one-GPU box) exchanges by device copies. */
typedef struct mg_comm   mg_comm;
typedef struct mg_dtable mg_dtable;
int      mg_comm_create_local(const int *devices, int n, mg_comm **out);
int      mg_comm_unique_id(void *id_out, size_t id_bytes);                 /* >= 128 bytes */
int      mg_comm_create_rank(mg_ctx *ctx, const void *id, size_t id_bytes, int nranks, int rank, mg_comm **out);
void     mg_comm_destroy(mg_comm *c);
int      mg_comm_size(const mg_comm *c);
int      mg_comm_rank(const mg_comm *c);
int      mg_comm_uses_rccl(const mg_comm *c);                              /* 0: one device / repeated devices */
mg_ctx  *mg_comm_ctx(mg_comm *c, int i);                                   /* local: context of device i */
const char *mg_comm_last_error(mg_comm *c);
/* rows [row_begin,row_end) of the lower triangle -> block of `rank`: equal numbers of pairs */
void     mg_shard_tri_rows(uint64_t row_begin, uint64_t row_end, int nranks, int rank, uint64_t *b_out, uint64_t *e_out);
/* The same with a cost per row on top of the cost per pair (row i costs i + row_weight pair-units): the inverted-index
 * engine writes 8 bytes per pair but discovers and merges per row, so equal areas overload the block of the short rows.
 * bench.py measures row_weight on the table in its warm-up (time per row of discover + merge over time per pair of the
 * fill, summed over the ranks).  row_weight <= 0: mg_shard_tri_rows. */
void     mg_shard_tri_rows_weighted(uint64_t row_begin, uint64_t row_end, int nranks, int rank, double row_weight,
                                    uint64_t *b_out, uint64_t *e_out);
/* ... and with a cost per row of the table BELOW a block's end (prefix_weight hi pair-units for the block [lo, hi)): a rank
 * builds its inverted index over the rows below its block's end only (a triangle job over rows [lo, hi) looks at no row from hi
 * on, CommandTriangle.cpp:200-214), so late blocks pay for more of the table and get fewer pairs.  bench.py measures
 * prefix_weight in its warm-up (index time per row of the view over fill time per pair).  prefix_weight <= 0: the call above. */
void     mg_shard_tri_rows_costed(uint64_t row_begin, uint64_t row_end, int nranks, int rank, double row_weight, double prefix_weight,
                                  uint64_t *b_out, uint64_t *e_out);
void     mg_shard_rows(uint64_t row_begin, uint64_t row_end, int nranks, int rank, uint64_t *b_out, uint64_t *e_out);
int      mg_dtable_upload(mg_comm *c, const uint64_t *hashes, const uint32_t *nhash, const uint64_t *lengths,
                          uint64_t n, uint64_t s, mg_dtable **out);         /* host -> GPU 0 -> broadcast */
/* The larger side of a rect job need not be replicated: device g gets a block of consecutive rows (host ->
 * each GPU its own rows, no exchange).  Such a table can only be the REFERENCE side of
 * mg_compare_rect_*_sharded_host, which then cuts the job by reference rows and puts the blocks back into
 * the reference's query-major order (SURVEY.md 8e "broadcast the smaller side, shard the larger side by
 * rows"; the reference cuts the same grid into jobs of 0x1000 pairs, CommandDistance.cpp:195-232).  With
 * replicated tables the rect calls cut whichever side is larger (one query against a million references
 * uses every GPU). */
int      mg_dtable_upload_rows(mg_comm *c, const uint64_t *hashes, const uint32_t *nhash, const uint64_t *lengths,
                               uint64_t n, uint64_t s, mg_dtable **out);
void     mg_dtable_free(mg_dtable *d);
mg_table *mg_dtable_local(mg_dtable *d, int i);
int      mg_table_broadcast(mg_comm *c, const mg_table *src, int root, uint64_t n, uint64_t s, mg_table **out);
int      mg_comm_allreduce_u32_sum(mg_comm *c, uint32_t *buf_dev, uint64_t count);
/* mg_sketch_host on every GPU: blocks of consecutive sketches balanced by bytes, one host thread per device,
 * no collective; rows of the outputs in input order (replaces the fan-out of sketchFile / sketchSequence over
 * the -p threads, Sketch.cpp:211,354, whose results ThreadPool.hxx:127-167 hands back in submission order). */
int mg_sketch_sharded_host(mg_comm *c, const mg_params *p, const uint8_t *bases, uint64_t nbases, const uint64_t *sketch_off,
                           uint64_t nsketch, uint64_t *hashes_out, uint32_t *nhash_out, uint32_t *counts_out);
int mg_compare_tri_sharded_host(mg_comm *c, const mg_dtable *t, uint64_t row_begin, uint64_t row_end, mg_counts *out_host);
int mg_compare_rect_sharded_host(mg_comm *c, const mg_dtable *ref, const mg_dtable *qry, uint64_t q_begin, uint64_t q_end,
                                 mg_counts *out_host);
int mg_compare_tri_pairs_sharded_host(mg_comm *c, const mg_dtable *t, uint64_t row_begin, uint64_t row_end, int kmer_size,
                                      double kmer_space, double max_distance, double max_p_value, mg_pair *out_host);
int mg_compare_rect_pairs_sharded_host(mg_comm *c, const mg_dtable *ref, const mg_dtable *qry, uint64_t q_begin,
                                       uint64_t q_end, int kmer_size, double kmer_space, double max_distance,
                                       double max_p_value, mg_pair *out_host);
int mg_compare_tri_results_sharded_host(mg_comm *c, const mg_dtable *t, uint64_t row_begin, uint64_t row_end, int kmer_size,
                                        double kmer_space, double max_distance, double max_p_value, mg_result *out_host,
                                        uint64_t capacity, uint64_t *count_out);
int mg_compare_rect_results_sharded_host(mg_comm *c, const mg_dtable *ref, const mg_dtable *qry, uint64_t q_begin,
                                         uint64_t q_end, int kmer_size, double kmer_space, double max_distance,
                                         double max_p_value, mg_result *out_host, uint64_t capacity, uint64_t *count_out);

/* ---- screening (containment of sketches in a mixture) -----------------------------
 * Replaces, for nucleotide query sketches, the data-parallel part of `mash screen`:
 * the hashTable / hashCounts build (CommandScreen.cpp:99-114), hashSequence over the
 * mixture (CommandScreen.cpp:484-599: every valid canonical k-mer is hashed, counted if
 * it occurs in any query sketch, and offered to the mixture's own bottom-s heap) and the
 * per-hash observation lookup behind `shared` / median multiplicity (:338-355, :402-455).
 *
 * mg_screen_create builds the device table of distinct hashes of `db` (which must outlive
 * the screen; beyond 2^31 hashes -- two million sketches of s = 1000 -- only the dense results, mg_screen_finish_host /
 * mg_screen_counts_dev, are available: mg_screen_reset and mg_screen_finish_sparse_host return MG_ERR_UNSUPPORTED).  mg_screen_add_* consumes one batch of the mixture: records separated by
 * MG_RECORD_SEP, any case, as kseq delivers them (records shorter than k contribute no
 * k-mer).  mg_screen_finish_host returns counts_out[db_rows * db_s] = observations of
 * every sketch hash in the mixture (0 beyond nhash), the mixture's bottom-s sketch
 * (mix_hashes_out[sketch_size], ascending, padded; its estimateSetSize feeds the p-value,
 * CommandScreen.cpp:322) and the number of distinct hashes in the table.
 */
typedef struct mg_screen mg_screen;
int  mg_screen_create(mg_ctx *ctx, const mg_params *p, const mg_table *db, mg_screen **out);
/* Amino-acid query sketches (`trans`, CommandScreen.cpp:120): the mixture stays nucleotide and
 * every batch is translated in six frames on the device (:516-531, translate/aaFromCodon
 * :617-809; codons holding anything but ACGT give '*', which ends k-mers) before the same pass.
 * `p` must carry the sketches' amino-acid alphabet (noncanonical). */
int  mg_screen_create_translated(mg_ctx *ctx, const mg_params *p, const mg_table *db, mg_screen **out);
int  mg_screen_add_host(mg_screen *sc, const uint8_t *bases, uint64_t nbases);
int  mg_screen_add_dev(mg_screen *sc, const uint8_t *bases_dev, uint64_t nbases);
int  mg_screen_finish_host(mg_screen *sc, uint32_t *counts_out, uint64_t *mix_hashes_out,
                           uint32_t *mix_nhash_out, uint64_t *distinct_out);
/* Device-resident counts_out[db_rows * db_s] (same content as mg_screen_finish_host's):
 * the operand of the one collective of a read-sharded screen -- every rank screens
 * its share of the mixture against the same db, the counts are summed across
 * ranks (RCCL all-reduce, u32 sum) and the per-rank mixture sketches are merged
 * (bottom-s of their union); see mash_amd/screen_dist.py. */
int  mg_screen_counts_dev(mg_screen *sc, uint32_t *counts_out_dev);
/* A database that stays resident (the reference rebuilds hashTable / hashCounts for every run,
 * CommandScreen.cpp:93-116; a service screens mixture after mixture against one database):
 * mg_screen_reset makes the screen ready for the NEXT mixture -- counters of the hashes the last one
 * touched back to 0, mixture sketch emptied; the key table, its two-tier bound and the rows-by-hash
 * index stay.  mg_screen_finish_sparse_host returns what a mixture touched instead of a dense matrix:
 * one hit per (row, hash) whose hash was observed -- {row, count = observations, hash} -- i.e. the
 * non-zero cells of mg_screen_finish_host's counts_out (`shared` of a row = its hits, its median
 * multiplicity = the median of their counts; CommandScreen.cpp:338-355), ordered by row, then hash.
 * *nhits_out = their number; the first min(capacity, *nhits_out) are written (call with capacity 0 to
 * size the buffer).  Cost proportional to what was touched, not to the database.
 * mg_screen_tier_note: how the key bound was laid out for this database (one tier / two tiers, see
 * DESIGN.md section 7), for logs. */
typedef struct mg_screen_hit { uint32_t row, count; uint64_t hash; } mg_screen_hit;
int  mg_screen_reset(mg_screen *sc);
int  mg_screen_finish_sparse_host(mg_screen *sc, mg_screen_hit *hits_out, uint64_t capacity, uint64_t *nhits_out,
                                  uint64_t *mix_hashes_out, uint32_t *mix_nhash_out, uint64_t *distinct_out);
const char *mg_screen_tier_note(const mg_screen *sc);
void mg_screen_free(mg_screen *sc);
/* estimateIdentity (CommandScreen.cpp:463-482) and pValueWithin (:601-615), host arithmetic. */
double mg_identity(uint64_t common, uint64_t denom, int kmer_size);
double mg_p_value_within(uint64_t x, uint64_t set_size, double kmer_space, uint64_t sketch_size);

/* Screening on every GPU of a local communicator: the mixture is sharded by batch (batch b ->
 * device b mod G, each with its own replica of the query table and its own host thread), the
 * observation counters are summed at the end (ncclReduce to GPU 0) and the per-device mixture
 * sketches merged -- SURVEY.md section 8e, the C++ form of mash_amd/screen_dist.py.  Results equal
 * one mg_screen fed every batch.  mg_dscreen_add_host returns once the batch is copied. */
typedef struct mg_dscreen mg_dscreen;
int  mg_dscreen_create(mg_comm *c, const mg_params *p, const mg_dtable *db, int translated, mg_dscreen **out);
int  mg_dscreen_add_host(mg_dscreen *d, const uint8_t *bases, uint64_t nbases);
int  mg_dscreen_finish_host(mg_dscreen *d, uint32_t *counts_out, uint64_t *mix_hashes_out, uint32_t *mix_nhash_out,
                            uint64_t *distinct_out);
/* The sparse form over all devices: every device's hits (mg_screen_finish_sparse_host) are merged on the
 * host -- counts of the same (row, hash) summed -- so the exchange is megabytes over PCIe, no collective;
 * mg_dscreen_reset readies every device for the next mixture. */
int  mg_dscreen_finish_sparse_host(mg_dscreen *d, mg_screen_hit *hits_out, uint64_t capacity, uint64_t *nhits_out,
                                   uint64_t *mix_hashes_out, uint32_t *mix_nhash_out, uint64_t *distinct_out);
int  mg_dscreen_reset(mg_dscreen *d);
void mg_dscreen_free(mg_dscreen *d);

/* ---- timing hook for bench.py ----------------------------------------------
 * Average duration (ms) of the last `name` kernel launches recorded with HIP
 * events on the context's stream since mg_prof_reset; name = "compare" or
 * "sketch", or a phase of the inverted-index compare engine: "compare_index",
 * "compare_discover", "compare_fill", "compare_dense", "compare_merge",
 * "compare_join", and "compare_fill_aside" -- the fill's launch beside the
 * index build of a per-table job, on a stream of the library's own (ordered
 * with the context's stream by events: the caller sees one stream).
 * launches_out receives the number of launches averaged. */
int    mg_prof_enable(mg_ctx *ctx, int on);
void   mg_prof_reset(mg_ctx *ctx);
double mg_prof_avg_ms(mg_ctx *ctx, const char *name, uint64_t *launches_out);

#ifdef __cplusplus
}
#endif
#endif /* MASHGPU_H */
