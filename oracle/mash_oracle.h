/*
 * mash_oracle.h — CPU ORACLE for the Mash sketch + distance hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product path: only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load this library, and only as the checker.
 *
 * It is a plain-C restatement of the reference algorithm (marbl/Mash 2.3),
 * each function citing the reference file:line it follows.  It is pinned
 * against the reference's own golden vectors (tests/golden/, see
 * tests/test_oracle_golden.py): test/ref/reads.json (1000 hashes +
 * length 502359 from test/reads{1,2}.fastq), test/ref/genomes.dist
 * (41/35/41 of 1000, distances, p-values to 6 digits) and the tutorial
 * known-answers (456/1000 -> 0.0222766).  Where oracle/_ref can be built
 * (this container) the restatement is additionally cross-checked against the
 * reference's own compiled objects on random inputs.
 *
 * Third-party arithmetic absent from /root/reference: the binomial tail
 * (GSL gsl_cdf_binomial_Q or Boost.Math binomial, version unpinned by the
 * reference, CommandDistance.cpp:443-447).  Restated here from the published
 * definition Q(x-1; n, r) = I_r(x, n-x+1) (regularized incomplete beta,
 * continued fraction, Lentz).  The reference pins it to 6 significant digits
 * only (test/ref/genomes.dist, test/ref/screen); finer agreement is
 * "parity unpinned" and is checked against scipy.stats.binom.sf fixtures.
 */
#ifndef MASH_ORACLE_H
#define MASH_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int      kmer_size;        /* Sketch::Parameters::kmerSize          (Sketch.h:86)  */
    uint64_t sketch_size;      /* ::minHashesPerWindow                  (Sketch.h:94)  */
    uint32_t seed;             /* ::seed                                (Sketch.h:91)  */
    int      use64;            /* ::use64                               (Sketch.h:90)  */
    int      noncanonical;     /* ::noncanonical                        (Sketch.h:98)  */
    int      preserve_case;    /* ::preserveCase                        (Sketch.h:89)  */
    uint8_t  alphabet[256];    /* ::alphabet                            (Sketch.h:87)  */
    uint32_t min_copies;       /* ::minCov (reads mode, -m)             (Sketch.h:102); 0 or 1 = off */
    double   target_cov;       /* ::targetCov (reads mode, -c)          (Sketch.h:103); 0 = off      */
    uint64_t bloom_bytes;      /* ::memoryBound (reads mode, -b)        (Sketch.h:101); 0 = off      */
} oracle_params;

/* setAlphabetFromString, Sketch.cpp:1108-1137 (fills alphabet + use64). */
void oracle_set_alphabet(oracle_params *p, const char *characters);

/* MurmurHash3_x64_128, MurmurHash3.cpp:255-335. */
void oracle_murmur3_x64_128(const void *key, int len, uint32_t seed, uint64_t out[2]);

/* getHash, hash.cpp:10-38: low 64 bits (use64) or low 32 bits of h1. */
uint64_t oracle_get_hash(const char *kmer, int k, uint32_t seed, int use64);

/* Opaque MinHashHeap (MinHashHeap.cpp:68-145).  oracle_heap_new: minCov 1;
 * oracle_heap_new_m: multiplicityMinimum >= 1 with the pending set / pending queue (:101-118,
 * :131-141); oracle_heap_new_b: with the Bloom filter of `-b <bytes>` (:19-41, :78-94; geometry
 * and hash: see mash_oracle.c). */
typedef struct oracle_heap oracle_heap;
oracle_heap *oracle_heap_new(uint64_t cardinality_max, int use64);
oracle_heap *oracle_heap_new_m(uint64_t cardinality_max, int use64, uint64_t multiplicity_min);
oracle_heap *oracle_heap_new_b(uint64_t cardinality_max, int use64, uint64_t bloom_bytes);
/* hash_ap of the vendored bloom_filter.hpp (:526-568) over the 8 / 4 bytes of a hash, with the
 * filter's single salt; the bit a hash maps to is this value modulo bloom_bytes * 8 (:443-447) */
uint32_t     oracle_bloom_hash(uint64_t hash, int use64);
void         oracle_heap_free(oracle_heap *h);
void         oracle_heap_try_insert(oracle_heap *h, uint64_t hash);
uint64_t     oracle_heap_size(const oracle_heap *h);
/* estimateSetSize / estimateMultiplicity, MinHashHeap.h:44-45 */
double       oracle_heap_estimate_set_size(const oracle_heap *h);
double       oracle_heap_estimate_multiplicity(const oracle_heap *h);
/* toHashList, HashSet.cpp:78-118: ascending hashes + parallel counts. Returns n. */
uint64_t     oracle_heap_to_list(const oracle_heap *h, uint64_t *hashes, uint32_t *counts);

/* addMinHashes, Sketch.cpp:512-583.  `seq` is modified in place (uppercased)
 * exactly as the reference does. */
void oracle_add_min_hashes(oracle_heap *h, char *seq, uint64_t length, const oracle_params *p);
/* the same loop without a heap: every valid k-mer hash of every record, in order (what mash screen's
 * hashSequence looks up, CommandScreen.cpp:533-575); returns their number */
uint64_t oracle_kmer_hashes(char *bases, const uint64_t *rec_off, uint64_t nrec, const oracle_params *p, uint64_t *out);

/*
 * One sketch from a list of records, following sketchFile (Sketch.cpp:1147-1336,
 * concatenated mode, records shorter than k skipped and not counted in length)
 * or sketchSequence (Sketch.cpp:1338-1365) when nrec == 1.
 * bases: all records back to back; rec_off[nrec+1] byte offsets.
 * Outputs: hashes_out/counts_out (capacity sketch_size), *n_out, *length_out
 * (sum of record lengths >= k), *set_size_out (estimateSetSize).
 * Returns 0, or -1 if no record is >= k long.
 */
int oracle_sketch_records(const char *bases, const uint64_t *rec_off, uint64_t nrec,
                          const oracle_params *p,
                          uint64_t *hashes_out, uint32_t *counts_out, uint64_t *n_out,
                          uint64_t *length_out, double *set_size_out);

/* translate / aaFromCodon, CommandScreen.cpp:617-809: dst[a] = amino acid of the codon
 * src[3a..3a+2] (standard code, upper-case ACGT only); any other byte in the codon gives '*'. */
void oracle_translate(const char *src, char *dst, uint64_t len);

/* Same, reads mode with the early stop of -c (Sketch.cpp:1258): after every record the loop
 * ends once estimateMultiplicity() >= p->target_cov.  *used_out = records (>= k long) consumed,
 * the "Reads used" line (:1324-1327); *mult_out = estimateMultiplicity at the end. */
int oracle_sketch_reads(const char *bases, const uint64_t *rec_off, uint64_t nrec,
                        const oracle_params *p,
                        uint64_t *hashes_out, uint32_t *counts_out, uint64_t *n_out,
                        uint64_t *length_out, double *set_size_out, uint64_t *used_out, double *mult_out);

typedef struct {
    uint64_t numer;     /* PairOutput::numer    CommandDistance.h:63-70 */
    uint64_t denom;
    double   distance;
    double   p_value;
    int      pass;
} oracle_pair;

/* compareSketches, CommandDistance.cpp:336-425. Fields other than `pass`
 * are left untouched when the distance filter rejects, as in the reference. */
void oracle_compare_sketches(oracle_pair *out,
                             const uint64_t *ref, uint64_t nref, uint64_t len_ref,
                             const uint64_t *qry, uint64_t nqry, uint64_t len_qry,
                             uint64_t sketch_size, int kmer_size, double kmer_space,
                             double max_distance, double max_p_value);

/* pValue, CommandDistance.cpp:427-448 (binomial tail restated, see header). */
double oracle_p_value(uint64_t x, uint64_t len_ref, uint64_t len_qry,
                      double kmer_space, uint64_t sketch_size);

/* Q(k; n, p) = P[Binomial(n,p) > k]  (gsl_cdf_binomial_Q semantics). */
double oracle_binomial_q(unsigned int k, double p, unsigned int n);

/* Bulk drivers used by bench.py's cpu_baseline leg and by tests:
 * triangle rows [row_begin,row_end): out pairs in reference order
 * (CommandTriangle.cpp:200-214: for i in rows, for j<i). table is n x s,
 * row-padded; nhash[i] = valid entries. Writes numer/denom only when
 * want_stats == 0. Returns pair count. */
uint64_t oracle_triangle(const uint64_t *table, const uint32_t *nhash, const uint64_t *lengths,
                         uint64_t n, uint64_t s, uint64_t row_begin, uint64_t row_end,
                         int kmer_size, double kmer_space, int want_stats,
                         uint32_t *numer_out, uint32_t *denom_out,
                         double *dist_out, double *pval_out);

#ifdef __cplusplus
}
#endif
#endif
