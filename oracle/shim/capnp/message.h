// oracle/shim/capnp/message.h — TEST INFRASTRUCTURE (see oracle/Makefile, target `refcli`).
//
// Just enough of the Cap'n Proto C++ surface for the reference's Sketch.cpp to compile UNMODIFIED
// (Sketch.cpp:283-320, 394-486, 924-1064 are its only users): message builder / reader objects over
// an in-memory model of the MinHash schema, serialised by this repository's own wire codec
// (mash_amd/host/msh_file.{h,cpp}) since libcapnp is not available in this image.  With these three
// headers (message.h, serialize.h, mash/capnp/MinHash.capnp.h) and a binomial tail for
// gsl_cdf_binomial_{P,Q}, all of the reference's sources build into `oracle/_ref/mash-ref`: the
// reference CLI itself, used as the end-to-end CPU oracle for the CLI tests' golden outputs.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "msh_file.h"          // -I mash_amd/host

namespace kj {
template <class T>
struct ArrayPtr {
    T *ptr;
    size_t count;
    ArrayPtr(T *p, size_t n) : ptr(p), count(n) {}
};
}  // namespace kj

namespace capnp {

struct word { uint64_t raw; };

struct ReaderOptions {
    uint64_t traversalLimitInWords = 0;
    int nestingLimit = 0;
};

// Text::Reader: assignable to std::string, and .cStr() as Sketch.cpp:309 uses it
struct TextReader : std::string {
    TextReader() = default;
    explicit TextReader(const std::string &s) : std::string(s) {}
    const char *cStr() const { return c_str(); }
};

template <class T>
struct List;                      // specialisations in mash/capnp/MinHash.capnp.h

class MallocMessageBuilder {
public:
    mshio::File file;
    template <class T>
    typename T::Builder initRoot() { return typename T::Builder(&file); }
};

}  // namespace capnp
