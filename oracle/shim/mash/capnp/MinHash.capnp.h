// oracle/shim/mash/capnp/MinHash.capnp.h — TEST INFRASTRUCTURE, see ../../capnp/message.h.
//
// Reader / Builder classes of the MinHash schema (capnp/MinHash.capnp:12-59 of the reference) with the
// accessors Sketch.cpp calls, over mshio::File.  hashes32 and hashes64 share one zero-extended list in
// that model (the file holds exactly one of them, chosen by use64 = alphabetSize^k > 2^32 on both
// sides, Sketch.cpp:1136); loci are not modelled (the default build writes none, Sketch.cpp:448-471).
#pragma once
#include "capnp/message.h"

namespace capnp {

template <>
struct List<uint64_t> {
    struct Reader {
        const std::vector<uint64_t> *v;
        uint64_t size() const { return v->size(); }
        uint64_t operator[](uint64_t i) const { return (*v)[i]; }
    };
    struct Builder {
        std::vector<uint64_t> *v;
        void set(uint64_t i, uint64_t x) { (*v)[i] = x; }
    };
};

template <>
struct List<uint32_t> {
    struct Reader {                                   // over hashes (zero-extended) or counts
        const std::vector<uint64_t> *v64;
        const std::vector<uint32_t> *v32;
        uint64_t size() const { return v64 ? v64->size() : v32->size(); }
        uint32_t operator[](uint64_t i) const { return v64 ? (uint32_t)(*v64)[i] : (*v32)[i]; }
    };
    struct Builder {
        std::vector<uint64_t> *v64;
        std::vector<uint32_t> *v32;
        void set(uint64_t i, uint32_t x) { if (v64) (*v64)[i] = x; else (*v32)[i] = x; }
    };
};

struct MinHash {
    struct ReferenceList {
        struct Reference {
            struct Reader {
                const mshio::Reference *r;
                TextReader getName() const { return TextReader(r->name); }
                TextReader getComment() const { return TextReader(r->comment); }
                uint64_t getLength64() const { return r->length; }
                uint32_t getLength() const { return 0; }      // legacy field, folded into length64 by the parser
                List<uint64_t>::Reader getHashes64() const { return {&r->hashes}; }
                List<uint32_t>::Reader getHashes32() const { return {&r->hashes, nullptr}; }
                bool hasCounts32() const { return !r->counts.empty(); }
                List<uint32_t>::Reader getCounts32() const { return {nullptr, &r->counts}; }
                bool getCounts32Sorted() const { return r->counts_sorted; }
            };
            struct Builder {
                mshio::Reference *r;
                mshio::File *f;
                void setName(const std::string &s) { r->name = s; }
                void setComment(const std::string &s) { r->comment = s; }
                void setLength64(uint64_t v) { r->length = v; }
                List<uint64_t>::Builder initHashes64(uint64_t n) { r->hashes.assign(n, 0); return {&r->hashes}; }
                List<uint32_t>::Builder initHashes32(uint64_t n) { r->hashes.assign(n, 0); return {&r->hashes, nullptr}; }
                // only called when parameters.counts is set (Sketch.cpp:432): that is the file-level flag
                List<uint32_t>::Builder initCounts32(uint64_t n) { f->header.has_counts = true; r->counts.assign(n, 0); return {nullptr, &r->counts}; }
                void setCounts32Sorted(bool b) { r->counts_sorted = b; }
            };
        };
        struct Reader;
        struct Builder;
    };
    struct LocusList {
        struct Locus {
            struct Reader {
                uint32_t getSequence() const { return 0; }
                uint32_t getPosition() const { return 0; }
                uint64_t getHash64() const { return 0; }
            };
            struct Builder {
                void setSequence(uint32_t) {}
                void setPosition(uint32_t) {}
                void setHash64(uint64_t) {}
            };
        };
        struct Reader;
        struct Builder;
    };
    struct Reader;
    struct Builder;
};

template <>
struct List<MinHash::ReferenceList::Reference> {
    struct Reader {
        const std::vector<mshio::Reference> *v;
        uint64_t size() const { return v->size(); }
        MinHash::ReferenceList::Reference::Reader operator[](uint64_t i) const { return {&(*v)[i]}; }
    };
    struct Builder {
        mshio::File *f;
        MinHash::ReferenceList::Reference::Builder operator[](uint64_t i) { return {&f->references[i], f}; }
    };
};

template <>
struct List<MinHash::LocusList::Locus> {
    struct Reader {
        uint64_t size() const { return 0; }
        MinHash::LocusList::Locus::Reader operator[](uint64_t) const { return {}; }
    };
    struct Builder {
        MinHash::LocusList::Locus::Builder operator[](uint64_t) { return {}; }
    };
};

struct MinHash::ReferenceList::Reader {
    const mshio::File *f;
    List<Reference>::Reader getReferences() const { return {&f->references}; }
};
struct MinHash::ReferenceList::Builder {
    mshio::File *f;
    List<Reference>::Builder initReferences(uint64_t n) { f->references.assign(n, mshio::Reference()); return {f}; }
};
struct MinHash::LocusList::Reader {
    List<Locus>::Reader getLoci() const { return {}; }
};
struct MinHash::LocusList::Builder {
    List<Locus>::Builder initLoci(uint64_t) { return {}; }
};

struct MinHash::Reader {
    const mshio::File *f;
    explicit Reader(const mshio::File *file) : f(file) {}
    uint32_t getKmerSize() const { return f->header.kmer_size; }
    float getError() const { return f->header.error; }
    uint32_t getMinHashesPerWindow() const { return f->header.sketch_size; }
    uint32_t getWindowSize() const { return f->header.window_size; }
    bool getConcatenated() const { return f->header.concatenated; }
    bool getNoncanonical() const { return f->header.noncanonical; }
    bool getPreserveCase() const { return f->header.preserve_case; }
    uint32_t getHashSeed() const { return f->header.seed; }
    bool hasAlphabet() const { return f->header.has_alphabet; }
    TextReader getAlphabet() const { return TextReader(f->header.alphabet); }
    // the parser already resolved "referenceList, else referenceListOld" (Sketch.cpp:300,932)
    ReferenceList::Reader getReferenceList() const { return {f}; }
    ReferenceList::Reader getReferenceListOld() const { return {f}; }
    LocusList::Reader getLocusList() const { return {}; }
};

struct MinHash::Builder {
    mshio::File *f;
    explicit Builder(mshio::File *file) : f(file) {}
    void setKmerSize(uint32_t v) { f->header.kmer_size = v; }
    void setHashSeed(uint32_t v) { f->header.seed = v; }
    void setError(float v) { f->header.error = v; }
    void setMinHashesPerWindow(uint32_t v) { f->header.sketch_size = v; }
    void setWindowSize(uint32_t v) { f->header.window_size = v; }
    void setConcatenated(bool b) { f->header.concatenated = b; }
    void setNoncanonical(bool b) { f->header.noncanonical = b; }
    void setPreserveCase(bool b) { f->header.preserve_case = b; }
    void setAlphabet(const std::string &s) { f->header.alphabet = s; f->header.has_alphabet = true; }
    // the writer picks referenceListOld iff seed == 42, as Sketch.cpp:397 does
    ReferenceList::Builder initReferenceList() { return {f}; }
    ReferenceList::Builder initReferenceListOld() { return {f}; }
    LocusList::Builder initLocusList() { return {}; }
};

}  // namespace capnp
