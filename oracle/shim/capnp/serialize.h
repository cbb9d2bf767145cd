// oracle/shim/capnp/serialize.h — TEST INFRASTRUCTURE, see message.h in this directory.
#pragma once
#include <unistd.h>

#include <cstdio>
#include <cstdlib>

#include "message.h"

namespace capnp {

// Sketch.cpp:289,929 hand over the mmap'ed file; it is parsed once, here.
class FlatArrayMessageReader {
public:
    mshio::File file;
    FlatArrayMessageReader(kj::ArrayPtr<const word> data, ReaderOptions)
    {
        const std::string err = mshio::parse_msh(reinterpret_cast<const uint8_t *>(data.ptr), data.count * sizeof(word), file, false, 0);
        if (!err.empty()) {
            fprintf(stderr, "ERROR: %s\n", err.c_str());
            exit(1);
        }
    }
    template <class T>
    typename T::Reader getRoot() { return typename T::Reader(&file); }
};

// Sketch.cpp:486
inline void writeMessageToFd(int fd, MallocMessageBuilder &message)
{
    std::vector<uint64_t> words;
    const std::string err = mshio::serialize_msh(message.file, words);
    if (!err.empty()) {
        fprintf(stderr, "ERROR: %s\n", err.c_str());
        exit(1);
    }
    const char *p = reinterpret_cast<const char *>(words.data());
    size_t left = words.size() * 8;
    while (left) {
        const ssize_t n = write(fd, p, left);
        if (n <= 0) { perror("write"); exit(1); }
        p += n;
        left -= (size_t)n;
    }
}

}  // namespace capnp
