// tests/mshio_c.cpp — tiny C surface over the .msh codec and the fastx reader of mash_amd/host/, for the CPU tests
// (ctypes).  Test support: built as tests/libmshio.so by mash_amd/host/Makefile.  Host-only; no GPU dependency.
#include <cstdio>
#include <cstring>
#include <string>

#include "fastx.h"
#include "msh_file.h"

extern "C" {

// parse -> re-serialize -> parse again; returns 0 when both parses agree field by field
int mshio_roundtrip_check(const char *path)
{
    mshio::File a, b;
    if (!mshio::read_msh(path, a).empty()) return -1;
    std::vector<uint64_t> words;
    if (!mshio::serialize_msh(a, words).empty()) return -2;
    if (!mshio::parse_msh(reinterpret_cast<const uint8_t *>(words.data()), words.size() * 8, b, false, 0).empty()) return -3;
    if (a.references.size() != b.references.size()) return 1;
    if (a.header.kmer_size != b.header.kmer_size || a.header.sketch_size != b.header.sketch_size ||
        a.header.seed != b.header.seed || a.header.alphabet != b.header.alphabet ||
        a.header.noncanonical != b.header.noncanonical || a.header.preserve_case != b.header.preserve_case)
        return 2;
    for (size_t i = 0; i < a.references.size(); i++) {
        const auto &x = a.references[i], &y = b.references[i];
        if (x.name != y.name || x.comment != y.comment || x.length != y.length || x.hashes != y.hashes || x.counts != y.counts)
            return 3;
    }
    return 0;
}

// read `in_path`, write it to `out_path` with the given segment limit (small limits force the
// multi-segment layout); returns the number of segments written, or < 0
long mshio_rewrite(const char *in_path, const char *out_path, unsigned long long max_segment_words)
{
    mshio::File a;
    if (!mshio::read_msh(in_path, a).empty()) return -1;
    std::vector<uint64_t> words;
    if (!mshio::serialize_msh(a, words, max_segment_words).empty()) return -2;
    FILE *f = fopen(out_path, "wb");
    if (!f) return -3;
    fwrite(words.data(), 8, words.size(), f);
    fclose(f);
    return (long)(words[0] & 0xFFFFFFFFull) + 1;
}

// parse a raw message image (used with hand-built multi-segment / far-pointer messages)
int mshio_parse_summary(const uint8_t *data, uint64_t size, uint32_t *kmer, uint32_t *sketch_size, uint32_t *seed,
                        uint64_t *nref, uint64_t *first_hash, uint64_t *last_hash, char *name0, uint64_t name_cap)
{
    mshio::File f;
    const std::string e = mshio::parse_msh(data, size, f, false, 0);
    if (!e.empty()) return -1;
    *kmer = f.header.kmer_size;
    *sketch_size = f.header.sketch_size;
    *seed = f.header.seed;
    *nref = f.references.size();
    *first_hash = *last_hash = 0;
    if (!f.references.empty()) {
        const auto &r = f.references.back();
        if (!r.hashes.empty()) { *first_hash = r.hashes.front(); *last_hash = r.hashes.back(); }
        strncpy(name0, f.references[0].name.c_str(), name_cap - 1);
        name0[name_cap - 1] = 0;
    }
    return 0;
}

// a dense sketch table (hashes[n * s] ascending rows, nhash[n], lengths[n]) as a .msh file with names <prefix><row>
// (tools/compare_e2e.py: inputs for the compare commands of both CLIs without going through text)
int mshio_write_table(const char *path, const uint64_t *hashes, const uint32_t *nhash, const uint64_t *lengths, uint64_t n, uint64_t s,
                      uint32_t kmer, uint32_t seed, const char *prefix)
{
    mshio::File f;
    f.header.kmer_size = kmer;
    f.header.sketch_size = (uint32_t)s;
    f.header.seed = seed;
    f.header.alphabet = "ACGT";
    f.header.has_alphabet = true;
    f.header.concatenated = true;
    f.references.resize(n);
    for (uint64_t i = 0; i < n; i++) {
        mshio::Reference &r = f.references[i];
        r.name = std::string(prefix) + std::to_string(i);
        r.length = lengths[i];
        r.hashes.assign(hashes + i * s, hashes + i * s + nhash[i]);
    }
    return mshio::write_msh(path, f).empty() ? 0 : -1;
}

// all records of a file as "name\tcomment\tseq\n" lines in a malloc'ed buffer (differential
// tests against a byte-by-byte restatement of kseq); returns the last status of Reader::next
long fastx_dump(const char *path, char **out, unsigned long long *out_len)
{
    fastx::Reader rd;
    *out = nullptr;
    *out_len = 0;
    if (!rd.open(path)) return -10;
    std::string all;
    fastx::Record rec;
    long l;
    while ((l = rd.next(rec)) >= 0) {
        all += rec.name; all += '\t'; all += rec.comment; all += '\t'; all += rec.seq; all += '\n';
    }
    *out = (char *)malloc(all.size() + 1);
    memcpy(*out, all.data(), all.size());
    *out_len = all.size();
    return l;
}

void fastx_free(char *p) { free(p); }

// count records / total sequence bytes of a fasta/fastq file the way kseq would
long fastx_count(const char *path, long min_len, unsigned long long *total_bases, unsigned long long *name_bytes)
{
    fastx::Reader rd;
    if (!rd.open(path)) return -10;
    fastx::Record rec;
    long l, n = 0;
    *total_bases = 0;
    *name_bytes = 0;
    while ((l = rd.next(rec)) >= 0) {
        if (l < min_len) continue;
        n++;
        *total_bases += (unsigned long long)l;
        *name_bytes += rec.name.size() + rec.comment.size();
    }
    return l == -1 ? n : l;
}

}
