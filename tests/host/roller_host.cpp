// Host-compiled instance of the per-lane k-mer state used by the sketch kernel
// (mash_amd/csrc/kmer_hash.h is host+device).  Lets the CPU test-suite check the
// rolling ASCII / 2-bit windows, canonical choice and murmur against the oracle
// without a GPU.  Test-only; built by tests/test_roller_host.py with g++.
#include <stdint.h>
#include "../../mash_amd/csrc/kmer_hash.h"

template <int K>
static void run(const uint8_t *bytes, uint64_t n, int mode, const uint8_t *alpha, int fold,
                uint32_t seed, int use64, uint8_t *valid_out, uint64_t *hash_out)
{
    if (mode == 0) {
        mg::KmerRoller<K, true> r; r.reset();
        for (uint64_t i = 0; i < n; i++) {
            uint32_t c = bytes[i];
            if (fold) c &= 0xDFu;
            uint32_t code, comp;
            bool v = mg::dna_classify(c, code, comp);
            r.push(c, v, code, comp);
            valid_out[i] = r.kmer_valid();
            hash_out[i] = r.hash(seed, use64 != 0);
        }
    } else {
        mg::KmerRoller<K, false> r; r.reset();
        for (uint64_t i = 0; i < n; i++) {
            uint32_t c = bytes[i];
            bool v;
            if (mode == 1) {
                if (fold) c &= 0xDFu;
                uint32_t code, comp;
                v = mg::dna_classify(c, code, comp);
            } else {
                if (fold) c = mg::fold_upper(c);
                v = alpha[c] != 0;
            }
            r.push(c, v);
            valid_out[i] = r.kmer_valid();
            hash_out[i] = r.hash(seed, use64 != 0);
        }
    }
}

extern "C" int roller_run(int k, const uint8_t *bytes, uint64_t n, int mode, const uint8_t *alpha,
                          int fold, uint32_t seed, int use64, uint8_t *valid_out, uint64_t *hash_out)
{
    switch (k) {
#define C(KK) case KK: run<KK>(bytes, n, mode, alpha, fold, seed, use64, valid_out, hash_out); return 0;
        C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16)
        C(17) C(18) C(19) C(20) C(21) C(22) C(23) C(24) C(25) C(26) C(27) C(28) C(29) C(30) C(31) C(32)
#undef C
    }
    return -1;
}
