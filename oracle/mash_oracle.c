/*
 * mash_oracle.c — CPU ORACLE (test infrastructure, never shipped; see mash_oracle.h).
 *
 * Plain-C restatement of the Mash 2.3 hot path.  Citations are into
 * /root/reference/src/mash/.
 */
#define _DEFAULT_SOURCE                 /* lgamma_r under -std=c11 */
#include "mash_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* MurmurHash3_x64_128 — MurmurHash3.cpp:255-335 (Austin Appleby, public domain
 * algorithm).  Little-endian block loads (MurmurHash3.cpp:60-63).           */

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

/* fmix64 — MurmurHash3.cpp:81-90 */
static inline uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

void oracle_murmur3_x64_128(const void *key, int len, uint32_t seed, uint64_t out[2])
{
    const uint8_t *data = (const uint8_t *)key;
    const int nblocks = len / 16;
    uint64_t h1 = seed, h2 = seed;
    const uint64_t c1 = 0x87c37b91114253d5ULL;
    const uint64_t c2 = 0x4cf5ad432745937fULL;

    for (int i = 0; i < nblocks; i++) {            /* body, :272-284 */
        uint64_t k1, k2;
        memcpy(&k1, data + 16 * i, 8);
        memcpy(&k2, data + 16 * i + 8, 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }

    const uint8_t *tail = data + nblocks * 16;     /* tail, :289-315 */
    const int rem = len & 15;
    uint64_t k1 = 0, k2 = 0;
    for (int b = rem - 1; b >= 8; b--) k2 ^= (uint64_t)tail[b] << (8 * (b - 8));
    if (rem > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
    for (int b = (rem > 8 ? 8 : rem) - 1; b >= 0; b--) k1 ^= (uint64_t)tail[b] << (8 * b);
    if (rem > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }

    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;     /* finalization, :319-331 */
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    out[0] = h1; out[1] = h2;
}

/* getHash — hash.cpp:10-38 (non-ARCH_32 branch): first 8 (use64) or 4 bytes. */
uint64_t oracle_get_hash(const char *kmer, int k, uint32_t seed, int use64)
{
    uint64_t out[2];
    oracle_murmur3_x64_128(kmer, k, seed, out);
    return use64 ? out[0] : (uint64_t)(uint32_t)out[0];
}

/* setAlphabetFromString — Sketch.cpp:1108-1137 */
void oracle_set_alphabet(oracle_params *p, const char *characters)
{
    memset(p->alphabet, 0, 256);
    for (const char *c = characters; *c; c++) {
        char u = *c;
        if (!p->preserve_case && u > 96 && u < 123) u -= 32;
        p->alphabet[(unsigned char)u] = 1;
    }
    unsigned n = 0;
    for (int i = 0; i < 256; i++) n += p->alphabet[i] ? 1 : 0;
    p->use64 = pow((double)n, (double)p->kmer_size) > pow(2.0, 32.0);   /* :1136 */
}

/* ------------------------------------------------------------------------- */
/* MinHashHeap — MinHashHeap.cpp:68-145 for multiplicityMinimum==1 and no
 * bloom filter.  The reference keeps an unordered map {hash->count} plus a
 * max-heap; the observable state is "set of kept hashes, their counts, the
 * maximum kept hash".  A sorted array gives the same observable state.      */

struct oracle_heap {
    uint64_t  cap;      /* cardinalityMaximum */
    uint64_t  n;
    uint64_t  msum;     /* multiplicitySum */
    int       use64;
    uint64_t *v;        /* ascending, distinct, n <= cap+1 */
    uint32_t *c;
    /* multiplicityMinimum > 1: hashesPending (set with counts) + hashesQueuePending (max-heap
     * that may hold zombies already erased from the set), MinHashHeap.h:33-34 */
    uint64_t  mmin;
    uint64_t *pv; uint32_t *pc; uint64_t pn, pcap;      /* pending set, ascending */
    uint64_t *pq; uint64_t qn, qcap;                    /* pending queue: multiset, ascending */
    /* -b: Bloom filter in front of the kept set (MinHashHeap.cpp:78-94) */
    uint8_t  *bloom;
    uint64_t  bloom_bits;
};

static uint64_t lower_bound64(const uint64_t *v, uint64_t n, uint64_t x)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (v[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

oracle_heap *oracle_heap_new(uint64_t cardinality_max, int use64)
{
    oracle_heap *h = (oracle_heap *)calloc(1, sizeof *h);
    h->cap = cardinality_max;
    h->use64 = use64;
    h->v = (uint64_t *)malloc((cardinality_max + 2) * sizeof(uint64_t));
    h->c = (uint32_t *)malloc((cardinality_max + 2) * sizeof(uint32_t));
    return h;
}

oracle_heap *oracle_heap_new_m(uint64_t cardinality_max, int use64, uint64_t multiplicity_min)
{
    oracle_heap *h = oracle_heap_new(cardinality_max, use64);
    h->mmin = multiplicity_min;
    return h;
}

/* The Bloom filter of `mash sketch -b <bytes>` (MinHashHeap.cpp:19-41) is the vendored "Open Bloom
 * Filter" (bloom_filter.hpp) set up with projected_element_count 10^9, false_positive_probability 0
 * and maximum_size = bytes * 8.  With probability 0, compute_optimal_parameters
 * (bloom_filter.hpp:107-155) evaluates -k * 10^9 / log(1 - 0^(1/k)) = -inf already for k = 1, so
 * number_of_hashes = 1, and table_size = (unsigned long long)(-inf): undefined behaviour in C++.
 * x86-64 compilers emit cvttsd2si, whose out-of-range result is 2^63, and the clamp (:149-152)
 * turns that into maximum_size.  This DE-FACTO geometry -- ONE hash function over bytes * 8 bits --
 * is what is restated here; it is pinned by sketches that the reference's own MinHashHeap /
 * bloom_filter.hpp produced in this container (tests/golden/ref_sketch_vectors_b.npz, the `-b`
 * fixtures of tests/golden/cli).
 *   salt  (:449-508, salt_count 1): predef_salt[0] = 0xAAAAAAAA, then salt = salt * salt +
 *         (uint32) random_seed_, random_seed_ = 0xA5A5A5A55A5A5A5A * 0xA5A5A5A5 + 1 (:180)
 *   hash  hash_ap (:526-568) over the 8 bytes of a 64-bit hash (one round of the two-word mix)
 *         or the 4 bytes of a 32-bit hash (the single-word branch with loop = 0)
 *   bit   hash % table_size (:443-447); insert sets it (:281-291), contains tests it (:318-332). */
uint32_t oracle_bloom_hash(uint64_t hash, int use64)
{
    const uint64_t seed = 0xA5A5A5A55A5A5A5AULL * 0xA5A5A5A5ULL + 1ULL;
    uint32_t h = 0xAAAAAAAAu * 0xAAAAAAAAu + (uint32_t)seed;
    if (use64) {
        const uint32_t i1 = (uint32_t)hash, i2 = (uint32_t)(hash >> 32);
        h ^= (h << 7) ^ (i1 * (h >> 3)) ^ (~((h << 11) + (i2 ^ (h >> 5))));
    } else {
        const uint32_t i = (uint32_t)hash;
        h ^= ~((h << 11) + (i ^ (h >> 5)));
    }
    return h;
}

oracle_heap *oracle_heap_new_b(uint64_t cardinality_max, int use64, uint64_t bloom_bytes)
{
    oracle_heap *h = oracle_heap_new(cardinality_max, use64);
    h->mmin = 1;
    if (bloom_bytes) {
        h->bloom_bits = bloom_bytes * 8;
        /* a 32-bit hash never reaches beyond bit 2^32 - 1 */
        uint64_t bytes = bloom_bytes < (1ULL << 29) ? bloom_bytes : (1ULL << 29);
        h->bloom = (uint8_t *)calloc(bytes, 1);
    }
    return h;
}

void oracle_heap_free(oracle_heap *h)
{
    if (!h) return;
    free(h->v); free(h->c); free(h->pv); free(h->pc); free(h->pq); free(h->bloom); free(h);
}

static void pending_set_erase(oracle_heap *h, uint64_t x)
{
    uint64_t i = lower_bound64(h->pv, h->pn, x);
    if (i < h->pn && h->pv[i] == x) {
        memmove(h->pv + i, h->pv + i + 1, (h->pn - i - 1) * sizeof(uint64_t));
        memmove(h->pc + i, h->pc + i + 1, (h->pn - i - 1) * sizeof(uint32_t));
        h->pn--;
    }
}

/* the multiplicityMinimum > 1 branch of tryInsert (MinHashHeap.cpp:96-118 and :126-144) */
static void heap_try_insert_m(oracle_heap *h, uint64_t hash)
{
    if (!(h->n < h->cap || hash < h->v[h->n - 1])) return;          /* :70-74 */
    uint64_t lo = lower_bound64(h->v, h->n, hash);
    if (lo < h->n && h->v[lo] == hash) {                              /* :120-124 */
        h->c[lo]++;
        h->msum++;
    } else {
        uint64_t pi = lower_bound64(h->pv, h->pn, hash);
        const int pend = pi < h->pn && h->pv[pi] == hash;
        const uint64_t pcount = pend ? h->pc[pi] : 0;
        if (pcount == h->mmin - 1) {                                  /* :96-109 promote */
            memmove(h->v + lo + 1, h->v + lo, (h->n - lo) * sizeof(uint64_t));
            memmove(h->c + lo + 1, h->c + lo, (h->n - lo) * sizeof(uint32_t));
            h->v[lo] = hash; h->c[lo] = (uint32_t)h->mmin;
            h->n++; h->msum += h->mmin;
            pending_set_erase(h, hash);                               /* the queue keeps a zombie */
        } else {                                                      /* :110-118 */
            if (!pend) {
                if (h->qn == h->qcap) { h->qcap = h->qcap ? 2 * h->qcap : 1024; h->pq = (uint64_t *)realloc(h->pq, h->qcap * 8); }
                uint64_t qi = lower_bound64(h->pq, h->qn, hash);      /* hashesQueuePending.push */
                memmove(h->pq + qi + 1, h->pq + qi, (h->qn - qi) * 8);
                h->pq[qi] = hash; h->qn++;
                if (h->pn == h->pcap) {
                    h->pcap = h->pcap ? 2 * h->pcap : 1024;
                    h->pv = (uint64_t *)realloc(h->pv, h->pcap * 8);
                    h->pc = (uint32_t *)realloc(h->pc, h->pcap * 4);
                }
                memmove(h->pv + pi + 1, h->pv + pi, (h->pn - pi) * 8);
                memmove(h->pc + pi + 1, h->pc + pi, (h->pn - pi) * 4);
                h->pv[pi] = hash; h->pc[pi] = 0; h->pn++;
            }
            h->pc[pi]++;                                              /* hashesPending.insert(hash, 1) */
        }
    }
    if (h->n > h->cap) {                                              /* :126-144 */
        const uint64_t top = h->v[h->n - 1];
        h->msum -= h->c[h->n - 1];
        h->n--;
        /* drop pending hashes above the evicted top (zombies included) */
        while (h->qn > 0 && top < h->pq[h->qn - 1]) {
            pending_set_erase(h, h->pq[h->qn - 1]);
            h->qn--;
        }
    }
}

uint64_t oracle_heap_size(const oracle_heap *h) { return h->n; }

void oracle_heap_try_insert(oracle_heap *h, uint64_t hash)
{
    if (h->mmin > 1) { heap_try_insert_m(h, hash); return; }
    /* :70-74  size < max || hash < top */
    if (!(h->n < h->cap || hash < h->v[h->n - 1])) return;
    /* lower bound */
    uint64_t lo = 0, hi = h->n;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (h->v[mid] < hash) lo = mid + 1; else hi = mid;
    }
    if (lo < h->n && h->v[lo] == hash) {          /* :120-124 already kept: count++ */
        h->c[lo]++;
        h->msum++;
    } else if (h->bloom) {                         /* :78-94 Bloom filter first */
        const uint64_t bit = (uint64_t)oracle_bloom_hash(hash, h->use64) % h->bloom_bits;
        if (h->bloom[bit >> 3] & (1u << (bit & 7))) {    /* seen (or aliased) before: kept with count 2 */
            memmove(h->v + lo + 1, h->v + lo, (h->n - lo) * sizeof(uint64_t));
            memmove(h->c + lo + 1, h->c + lo, (h->n - lo) * sizeof(uint32_t));
            h->v[lo] = hash; h->c[lo] = 2;
            h->n++; h->msum += 2;
        } else {
            h->bloom[bit >> 3] |= (uint8_t)(1u << (bit & 7));
        }
    } else {                                       /* :96-100 insert with count 1 */
        memmove(h->v + lo + 1, h->v + lo, (h->n - lo) * sizeof(uint64_t));
        memmove(h->c + lo + 1, h->c + lo, (h->n - lo) * sizeof(uint32_t));
        h->v[lo] = hash; h->c[lo] = 1;
        h->n++; h->msum++;
    }
    if (h->n > h->cap) {                           /* :126-144 evict current max */
        h->msum -= h->c[h->n - 1];
        h->n--;
    }
}

/* MinHashHeap.h:44-45 */
double oracle_heap_estimate_multiplicity(const oracle_heap *h)
{
    return h->n ? (double)h->msum / (double)h->n : 0.0;
}

double oracle_heap_estimate_set_size(const oracle_heap *h)
{
    if (!h->n) return 0.0;
    return pow(2.0, h->use64 ? 64.0 : 32.0) * (double)h->n / (double)h->v[h->n - 1];
}

uint64_t oracle_heap_to_list(const oracle_heap *h, uint64_t *hashes, uint32_t *counts)
{
    if (hashes) memcpy(hashes, h->v, h->n * sizeof(uint64_t));
    if (counts) memcpy(counts, h->c, h->n * sizeof(uint32_t));
    return h->n;
}

/* ------------------------------------------------------------------------- */
/* reverseComplement — Sketch.cpp:1071-1106 (26-entry IUPAC table).          */
static const char complement_tab[26] = {
    'T','V','G','H','N','N','C','D','N','N','M','N','K',
    'N','N','N','N','Y','S','A','A','B','W','N','R','N'
};

/* addMinHashes — Sketch.cpp:512-583.  With h == NULL the hashes go to out[] in k-mer order instead
 * of a heap (the loop mash screen's hashSequence runs over every record, CommandScreen.cpp:533-575,
 * is this one with a table lookup in place of the heap); returns their number. */
static uint64_t scan_kmers(oracle_heap *h, uint64_t *out, char *seq, uint64_t length, const oracle_params *p)
{
    uint64_t emitted = 0;
    const int k = p->kmer_size;
    if (!p->preserve_case)                          /* :524-530 */
        for (uint64_t i = 0; i < length; i++)
            if (seq[i] > 96 && seq[i] < 123) seq[i] -= 32;

    char *rev = NULL;
    if (!p->noncanonical) {                         /* :534-538 */
        rev = (char *)malloc(length ? length : 1);
        for (uint64_t i = 0; i < length; i++) {
            int idx = (int)seq[length - i - 1] - 'A';
            /* the reference indexes its table unchecked (UB for non-letters);
             * such bytes are never inside a hashed k-mer, any value will do */
            rev[i] = (idx >= 0 && idx < 26) ? complement_tab[idx] : 'N';
        }
    }

    if (length >= (uint64_t)k) {
        uint64_t j = 0;
        for (uint64_t i = 0; i + k <= length; i++) { /* :542-581 */
            int bad = 0;
            for (; j < i + k; j++) {
                if (!p->alphabet[(unsigned char)seq[j]]) {
                    i = j++;                        /* skip past the bad character */
                    bad = 1;
                    break;
                }
            }
            if (bad) continue;
            const char *fwd = seq + i;
            const char *kmer = fwd;
            if (!p->noncanonical) {
                const char *rv = rev + length - i - k;
                if (memcmp(fwd, rv, (size_t)k) > 0) kmer = rv;   /* :569-571 */
            }
            const uint64_t hv = oracle_get_hash(kmer, k, p->seed, p->use64);
            if (h) oracle_heap_try_insert(h, hv);
            else out[emitted] = hv;
            emitted++;
        }
    }
    free(rev);
    return emitted;
}

void oracle_add_min_hashes(oracle_heap *h, char *seq, uint64_t length, const oracle_params *p)
{
    (void)scan_kmers(h, NULL, seq, length, p);
}

/* Every valid k-mer hash of every record (records of at least k bytes), in order: what hashSequence
 * looks up in the screen's table.  out must hold sum(max(0, len - k + 1)) entries; seq bytes are
 * uppercased in place as the reference does. */
uint64_t oracle_kmer_hashes(char *bases, const uint64_t *rec_off, uint64_t nrec, const oracle_params *p, uint64_t *out)
{
    uint64_t n = 0;
    for (uint64_t r = 0; r < nrec; r++) {
        const uint64_t l = rec_off[r + 1] - rec_off[r];
        if (l < (uint64_t)p->kmer_size) continue;
        n += scan_kmers(NULL, out + n, bases + rec_off[r], l, p);
    }
    return n;
}

/* sketchFile (concatenated) — Sketch.cpp:1147-1336; sketchSequence — :1338-1365 */
int oracle_sketch_records(const char *bases, const uint64_t *rec_off, uint64_t nrec,
                          const oracle_params *p,
                          uint64_t *hashes_out, uint32_t *counts_out, uint64_t *n_out,
                          uint64_t *length_out, double *set_size_out)
{
    return oracle_sketch_reads(bases, rec_off, nrec, p, hashes_out, counts_out, n_out, length_out, set_size_out,
                               NULL, NULL);
}

int oracle_sketch_reads(const char *bases, const uint64_t *rec_off, uint64_t nrec,
                        const oracle_params *p,
                        uint64_t *hashes_out, uint32_t *counts_out, uint64_t *n_out,
                        uint64_t *length_out, double *set_size_out, uint64_t *used_out, double *mult_out)
{
    /* Sketch.cpp:1156: minCov applies in reads mode; callers set min_copies only then */
    oracle_heap *h = p->bloom_bytes ? oracle_heap_new_b(p->sketch_size, p->use64, p->bloom_bytes)
                                    : oracle_heap_new_m(p->sketch_size, p->use64, p->min_copies > 1 ? p->min_copies : 1);
    uint64_t length = 0, used = 0;
    int any = 0;
    for (uint64_t r = 0; r < nrec; r++) {
        uint64_t l = rec_off[r + 1] - rec_off[r];
        if (l < (uint64_t)p->kmer_size) continue;   /* :1222-1226 skipped, not counted */
        any = 1;
        length += l;                                /* :1253 */
        char *copy = (char *)malloc(l + 1);
        memcpy(copy, bases + rec_off[r], l);
        copy[l] = 0;
        oracle_add_min_hashes(h, copy, l, p);
        free(copy);
        used++;
        /* :1258 (reads mode): stop as soon as the average multiplicity reaches the target */
        if (p->target_cov > 0 && oracle_heap_estimate_multiplicity(h) >= p->target_cov) break;
    }
    if (used_out) *used_out = used;
    if (mult_out) *mult_out = oracle_heap_estimate_multiplicity(h);
    uint64_t n = oracle_heap_to_list(h, hashes_out, counts_out);
    if (n_out) *n_out = n;
    if (length_out) *length_out = length;
    if (set_size_out) *set_size_out = oracle_heap_estimate_set_size(h);
    oracle_heap_free(h);
    return any ? 0 : -1;
}

/* ------------------------------------------------------------------------- */
/* translate / aaFromCodon — CommandScreen.cpp:617-809.  The nested switch of the reference
 * is the standard genetic code; here as a 64-entry table indexed by the codon read as three
 * base-4 digits with A=0, C=1, G=2, T=3 (pinned by tests/golden/codon_table.json, produced by
 * the reference's own function).                                                             */
void oracle_translate(const char *src, char *dst, uint64_t len)
{
    static const char code[65] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";
    for (uint64_t a = 0; a < len; a++) {
        int idx = 0, ok = 1;
        for (int j = 0; j < 3; j++) {
            int d;
            switch (src[3 * a + j]) {
                case 'A': d = 0; break;
                case 'C': d = 1; break;
                case 'G': d = 2; break;
                case 'T': d = 3; break;
                default: d = 0; ok = 0; break;
            }
            idx = idx * 4 + d;
        }
        dst[a] = ok ? code[idx] : '*';            /* :640 default */
    }
}

/* ------------------------------------------------------------------------- */
/* Binomial upper tail.  Reference: gsl_cdf_binomial_Q(x-1, r, n) or Boost
 * cdf(complement(binomial(n,r), x-1)) — CommandDistance.cpp:443-447; both are
 * P[X > k] = I_p(k+1, n-k) (regularized incomplete beta).                    */

static double beta_cont_frac(double a, double b, double x)
{
    /* modified Lentz evaluation of the standard continued fraction for
     * I_x(a,b) (Abramowitz & Stegun 26.5.8) */
    const double tiny = 1e-300, eps = 1e-16;
    double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 100000; m++) {
        double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < eps) break;
    }
    return h;
}

static double reg_inc_beta(double a, double b, double x)
{
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    /* lgamma_r: the plain form stores the sign in the global `signgam`; the baseline's threads
     * (bench.py cpu_baseline) would pass that cache line around.  Same value. */
    int sg;
    double ln_pre = lgamma_r(a + b, &sg) - lgamma_r(a, &sg) - lgamma_r(b, &sg) + a * log(x) + b * log1p(-x);
    if (x < (a + 1.0) / (a + b + 2.0))
        return exp(ln_pre) * beta_cont_frac(a, b, x) / a;
    return 1.0 - exp(ln_pre) * beta_cont_frac(b, a, 1.0 - x) / b;
}

double oracle_binomial_q(unsigned int k, double p, unsigned int n)
{
    if (k >= n) return 0.0;
    return reg_inc_beta((double)k + 1.0, (double)n - (double)k, p);
}

/* pValue — CommandDistance.cpp:427-448 */
double oracle_p_value(uint64_t x, uint64_t len_ref, uint64_t len_qry,
                      double kmer_space, uint64_t sketch_size)
{
    if (x == 0) return 1.0;
    double pX = 1.0 / (1.0 + kmer_space / (double)len_ref);
    double pY = 1.0 / (1.0 + kmer_space / (double)len_qry);
    double r = pX * pY / (pX + pY - pX * pY);
    return oracle_binomial_q((unsigned int)(x - 1), r, (unsigned int)sketch_size);
}

/* compareSketches — CommandDistance.cpp:336-425 */
void oracle_compare_sketches(oracle_pair *out,
                             const uint64_t *ref, uint64_t nref, uint64_t len_ref,
                             const uint64_t *qry, uint64_t nqry, uint64_t len_qry,
                             uint64_t sketch_size, int kmer_size, double kmer_space,
                             double max_distance, double max_p_value)
{
    uint64_t i = 0, j = 0, common = 0, denom = 0;
    out->pass = 0;
    while (denom < sketch_size && i < nref && j < nqry) {   /* :347-365 */
        if (ref[i] < qry[j]) i++;
        else if (qry[j] < ref[i]) j++;
        else { i++; j++; common++; }
        denom++;
    }
    if (denom < sketch_size) {                              /* :367-385 */
        if (i < nref) denom += nref - i;
        if (j < nqry) denom += nqry - j;
        if (denom > sketch_size) denom = sketch_size;
    }
    double distance;
    double jaccard = (double)common / (double)denom;        /* :388 */
    if (common == denom) distance = 0;                      /* :390-407 */
    else if (common == 0) distance = 1.;
    else {
        distance = -log(2 * jaccard / (1. + jaccard)) / kmer_size;
        if (distance > 1) distance = 1;
    }
    if (max_distance >= 0 && distance > max_distance) return;   /* :409-412 */
    out->numer = common;
    out->denom = denom;
    out->distance = distance;
    out->p_value = oracle_p_value(common, len_ref, len_qry, kmer_space, denom);
    if (max_p_value >= 0 && out->p_value > max_p_value) return; /* :419-422 */
    out->pass = 1;
}

/* compare (triangle) — CommandTriangle.cpp:200-214, rows [row_begin,row_end) */
uint64_t oracle_triangle(const uint64_t *table, const uint32_t *nhash, const uint64_t *lengths,
                         uint64_t n, uint64_t s, uint64_t row_begin, uint64_t row_end,
                         int kmer_size, double kmer_space, int want_stats,
                         uint32_t *numer_out, uint32_t *denom_out,
                         double *dist_out, double *pval_out)
{
    uint64_t idx = 0;
    if (row_end > n) row_end = n;
    for (uint64_t i = row_begin; i < row_end; i++) {
        for (uint64_t j = 0; j < i; j++, idx++) {
            oracle_pair po;
            po.numer = po.denom = 0; po.distance = 0; po.p_value = 0;
            if (want_stats) {
                oracle_compare_sketches(&po, table + i * s, nhash[i], lengths ? lengths[i] : 1,
                                        table + j * s, nhash[j], lengths ? lengths[j] : 1,
                                        s, kmer_size, kmer_space, -1.0, -1.0);
            } else {
                /* merge only (what the device compare kernel produces) */
                const uint64_t *a = table + i * s, *b = table + j * s;
                uint64_t na = nhash[i], nb = nhash[j], ia = 0, ib = 0, common = 0, denom = 0;
                while (denom < s && ia < na && ib < nb) {
                    if (a[ia] < b[ib]) ia++;
                    else if (b[ib] < a[ia]) ib++;
                    else { ia++; ib++; common++; }
                    denom++;
                }
                if (denom < s) {
                    if (ia < na) denom += na - ia;
                    if (ib < nb) denom += nb - ib;
                    if (denom > s) denom = s;
                }
                po.numer = common; po.denom = denom;
            }
            if (numer_out) numer_out[idx] = (uint32_t)po.numer;
            if (denom_out) denom_out[idx] = (uint32_t)po.denom;
            if (dist_out) dist_out[idx] = po.distance;
            if (pval_out) pval_out[idx] = po.p_value;
        }
    }
    return idx;
}
