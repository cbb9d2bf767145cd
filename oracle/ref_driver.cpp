/*
 * ref_driver.cpp — C-ABI wrapper around the REFERENCE's own hot-path code.
 *
 * TEST INFRASTRUCTURE ONLY (see mash_oracle.h).  Built by oracle/Makefile into
 * oracle/_ref/libmash_ref.so.  No reference source is copied into this repo:
 * the reference's leaf translation units (MurmurHash3.cpp hash.cpp HashList.cpp
 * HashPriorityQueue.cpp HashSet.cpp MinHashHeap.cpp) are compiled where they
 * lie under /root/reference/src/mash, and the free functions that live inside
 * Sketch.cpp / CommandDistance.cpp (which cannot be compiled whole here:
 * Sketch.cpp needs libcapnp, absent) are pulled in at BUILD time as generated
 * includes under oracle/_ref/gen/ (sed line ranges, never committed):
 *   sketch_hotpath.inc  = Sketch.cpp:1070-1106 (complement table, reverseComplement),
 *                         :512-583 (addMinHashes), :1108-1145 (setAlphabetFromString,
 *                         setMinHashesForReference)
 *   compare_hotpath.inc = CommandDistance.cpp:336-448 (compareSketches, pValue)
 * GSL is absent: gsl_cdf_binomial_Q is supplied by the oracle's restated
 * regularized incomplete beta (mash_oracle.c), so p-values from this library
 * are NOT an independent check of the binomial tail — merge counts, distances
 * and hash lists are.
 */
#include "mash/Sketch.h"
#include "mash/CommandDistance.h"
#include "mash/MurmurHash3.h"
#include "mash/hash.h"
#include "mash_oracle.h"

#include <cstring>
#include <cmath>
#include <vector>

using namespace std;

extern "C" double gsl_cdf_binomial_Q(unsigned int k, double p, unsigned int n)
{
    return oracle_binomial_q(k, p, n);
}
extern "C" double gsl_cdf_binomial_P(unsigned int k, double p, unsigned int n)
{
    return 1.0 - oracle_binomial_q(k, p, n);
}

/* forward declarations the .inc bodies expect (declared in Sketch.h:226-236) */
#include "gen/sketch_hotpath.inc"

namespace mash {
#include "gen/compare_hotpath.inc"
}

namespace mashscreen {                       /* CommandScreen.cpp:617-809 */
char aaFromCodon(const char *codon);
#include "gen/translate.inc"
}

static void fill_params(Sketch::Parameters &P, const oracle_params *p)
{
    P.kmerSize = p->kmer_size;
    P.minHashesPerWindow = p->sketch_size;
    P.seed = p->seed;
    P.noncanonical = p->noncanonical != 0;
    P.preserveCase = p->preserve_case != 0;
    P.alphabetSize = 0;
    for (int i = 0; i < 256; i++) {
        P.alphabet[i] = p->alphabet[i] != 0;
        P.alphabetSize += p->alphabet[i] ? 1 : 0;
    }
    P.use64 = pow(P.alphabetSize, P.kmerSize) > pow(2, 32);
}

extern "C" {

void ref_murmur3_x64_128(const void *key, int len, uint32_t seed, uint64_t out[2])
{
    MurmurHash3_x64_128(key, len, seed, out);
}

uint64_t ref_get_hash(const char *kmer, int k, uint32_t seed, int use64)
{
    hash_u h = getHash(kmer, k, seed, use64 != 0);
    return use64 ? h.hash64 : (uint64_t)h.hash32;
}

/* same contract as oracle_sketch_records */
int ref_sketch_reads(const char *bases, const uint64_t *rec_off, uint64_t nrec,
                     const oracle_params *p,
                     uint64_t *hashes_out, uint32_t *counts_out, uint64_t *n_out,
                     uint64_t *length_out, double *set_size_out, uint64_t *used_out, double *mult_out);

int ref_sketch_records(const char *bases, const uint64_t *rec_off, uint64_t nrec,
                       const oracle_params *p,
                       uint64_t *hashes_out, uint32_t *counts_out, uint64_t *n_out,
                       uint64_t *length_out, double *set_size_out)
{
    return ref_sketch_reads(bases, rec_off, nrec, p, hashes_out, counts_out, n_out, length_out, set_size_out, 0, 0);
}

/* the record loop of sketchFile with the reference's MinHashHeap and its -c early stop (Sketch.cpp:1258) */
int ref_sketch_reads(const char *bases, const uint64_t *rec_off, uint64_t nrec,
                     const oracle_params *p,
                     uint64_t *hashes_out, uint32_t *counts_out, uint64_t *n_out,
                     uint64_t *length_out, double *set_size_out, uint64_t *used_out, double *mult_out)
{
    Sketch::Parameters P;
    fill_params(P, p);
    MinHashHeap heap(P.use64, P.minHashesPerWindow, p->min_copies > 1 ? p->min_copies : 1, p->bloom_bytes);   // Sketch.cpp:1156
    uint64_t length = 0, used = 0;
    bool any = false;
    for (uint64_t r = 0; r < nrec; r++) {
        uint64_t l = rec_off[r + 1] - rec_off[r];
        if (l < (uint64_t)P.kmerSize) continue;
        any = true;
        length += l;
        vector<char> copy(bases + rec_off[r], bases + rec_off[r] + l);
        copy.push_back(0);
        addMinHashes(heap, copy.data(), l, P);
        used++;
        if (p->target_cov > 0 && heap.estimateMultiplicity() >= p->target_cov) break;
    }
    if (used_out) *used_out = used;
    if (mult_out) *mult_out = heap.estimateMultiplicity();
    Sketch::Reference ref;
    ref.hashesSorted.setUse64(P.use64);
    setMinHashesForReference(ref, heap);
    uint64_t n = ref.hashesSorted.size();
    for (uint64_t i = 0; i < n; i++) {
        hash_u h = ref.hashesSorted.at(i);
        if (hashes_out) hashes_out[i] = P.use64 ? h.hash64 : (uint64_t)h.hash32;
        if (counts_out) counts_out[i] = ref.counts[i];
    }
    if (n_out) *n_out = n;
    if (length_out) *length_out = length;
    if (set_size_out) *set_size_out = heap.estimateSetSize();
    return any ? 0 : -1;
}

static void to_reference(Sketch::Reference &r, const uint64_t *h, uint64_t n, uint64_t len, bool use64)
{
    r.length = len;
    r.hashesSorted.setUse64(use64);
    for (uint64_t i = 0; i < n; i++) {
        if (use64) r.hashesSorted.push_back64(h[i]);
        else r.hashesSorted.push_back32((uint32_t)h[i]);
    }
}

void ref_compare_sketches(oracle_pair *out,
                          const uint64_t *ref, uint64_t nref, uint64_t len_ref,
                          const uint64_t *qry, uint64_t nqry, uint64_t len_qry,
                          uint64_t sketch_size, int kmer_size, double kmer_space,
                          double max_distance, double max_p_value, int use64)
{
    Sketch::Reference a, b;
    to_reference(a, ref, nref, len_ref, use64 != 0);
    to_reference(b, qry, nqry, len_qry, use64 != 0);
    mash::CommandDistance::CompareOutput::PairOutput po;
    po.numer = out->numer; po.denom = out->denom;
    po.distance = out->distance; po.pValue = out->p_value; po.pass = false;
    mash::compareSketches(&po, a, b, sketch_size, kmer_size, kmer_space, max_distance, max_p_value);
    out->numer = po.numer; out->denom = po.denom;
    out->distance = po.distance; out->p_value = po.pValue; out->pass = po.pass ? 1 : 0;
}

/* CPU baseline driver: the reference's compareSketches over triangle rows
 * [row_begin,row_end) of a dense table (CommandTriangle.cpp:200-214 order).
 * References are materialised once, outside any timed region, by the caller
 * through ref_table_new / ref_table_free. */
struct ref_table { vector<Sketch::Reference> refs; uint64_t s; };

void *ref_table_new(const uint64_t *table, const uint32_t *nhash, const uint64_t *lengths,
                    uint64_t n, uint64_t s)
{
    ref_table *t = new ref_table;
    t->s = s;
    t->refs.resize(n);
    for (uint64_t i = 0; i < n; i++)
        to_reference(t->refs[i], table + i * s, nhash[i], lengths ? lengths[i] : 1, true);
    return t;
}

void ref_table_free(void *t) { delete (ref_table *)t; }

uint64_t ref_triangle(void *tv, uint64_t row_begin, uint64_t row_end, int kmer_size,
                      double kmer_space, uint32_t *numer_out, uint32_t *denom_out,
                      double *dist_out, double *pval_out)
{
    ref_table *t = (ref_table *)tv;
    uint64_t idx = 0;
    if (row_end > t->refs.size()) row_end = t->refs.size();
    for (uint64_t i = row_begin; i < row_end; i++)
        for (uint64_t j = 0; j < i; j++, idx++) {
            mash::CommandDistance::CompareOutput::PairOutput po;
            mash::compareSketches(&po, t->refs[i], t->refs[j], t->s, kmer_size, kmer_space, -1.0, -1.0);
            if (numer_out) numer_out[idx] = (uint32_t)po.numer;
            if (denom_out) denom_out[idx] = (uint32_t)po.denom;
            if (dist_out) dist_out[idx] = po.distance;
            if (pval_out) pval_out[idx] = po.pValue;
        }
    return idx;
}

/* translate, CommandScreen.cpp:617-623: dst[a] = aaFromCodon(src + 3a), a < len */
void ref_translate(const char *src, char *dst, uint64_t len) { mashscreen::translate(src, dst, len); }

} /* extern "C" */
