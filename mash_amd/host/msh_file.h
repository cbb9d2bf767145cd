// msh_file.h — the .msh sketch file (Cap'n Proto message, schema
// /root/reference/src/mash/capnp/MinHash.capnp:12-59) read and written WITHOUT libcapnp.
//
// Replaces Sketch::writeToCapnp / loadCapnp / initParametersFromCapnp
// (Sketch.cpp:384-490, :907-1067, :255-324).  The wire layout (struct sizes, field slots,
// default XOR for hashSeed) is derived from Cap'n Proto's slot-allocation rule
// (SURVEY.md Appendix A).  The reader accepts anything libcapnp can emit for this schema:
// multi-segment messages, far and double-far pointers.  The writer emits one segment (no far
// pointers) while the message fits in 2 GiB, and otherwise keeps the structs in segment 0 and
// moves every list to further segments behind far pointers (30-bit in-segment word offsets
// cannot span more); every Cap'n Proto reader accepts both.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mshio {

struct Reference {
    std::string name, comment;
    uint64_t length = 0;
    std::vector<uint64_t> hashes;        // ascending; 32-bit hashes zero-extended
    std::vector<uint32_t> counts;        // empty when the file has no counts32
    bool counts_sorted = false;
};

struct Header {
    uint32_t kmer_size = 0;
    uint32_t window_size = 0;
    uint32_t sketch_size = 0;            // minHashesPerWindow
    uint32_t seed = 42;
    float error = 0.f;
    bool concatenated = false, noncanonical = false, preserve_case = false;
    bool has_alphabet = false;
    std::string alphabet;                // as stored; "ACGT" when absent (Sketch.cpp:307-314)
    bool has_counts = false;             // references[0].hasCounts32() (Sketch.cpp:304)
    uint64_t reference_count = 0;
};

struct File {
    Header header;
    std::vector<Reference> references;
};

// use64 = alphabetSize^k > 2^32 (Sketch.cpp:1136); the flag is not stored in the file
bool use64_for(const std::string &alphabet, bool preserve_case, uint32_t kmer_size, uint32_t *alphabet_size_out = nullptr);

// Returns "" on success, else an error message.  max_hashes > 0 truncates every hash list
// (and counts) to that many entries, as loadCapnp does (Sketch.cpp:965-968).
std::string read_msh(const std::string &path, File &out, bool header_only = false, uint64_t max_hashes = 0);
std::string write_msh(const std::string &path, const File &in);

// whole file into memory ("" on success): callers that need the header first and the lists after
// parse the same image twice instead of reading the file twice
std::string load_file(const std::string &path, std::vector<uint8_t> &out);
// in-memory variants (tests, pipes)
std::string parse_msh(const uint8_t *data, size_t size, File &out, bool header_only, uint64_t max_hashes);
// framing included; max_segment_words = 0 means the default segment limit (2 GiB)
std::string serialize_msh(const File &in, std::vector<uint64_t> &words_out, uint64_t max_segment_words = 0);

}  // namespace mshio
