// fastx.h — FASTA/FASTQ (optionally gzipped) record reader with the semantics of the
// reference's vendored kseq.h (kseq.h:171-208), which are what decides record boundaries,
// names, comments and which bytes reach the sketching kernel:
//   * a record starts at the next '>' or '@';
//   * name = header up to the first whitespace, comment = rest of the line up to \n;
//   * sequence = every isgraph() byte up to the next '>', '@' or '+' (anywhere, not only
//     at line starts — kseq reads byte by byte);
//   * FASTQ: the '+' line is skipped, then quality bytes (33..127) are consumed until as
//     many as sequence bytes were read; a short quality string is error -2.
#pragma once
#include <zlib.h>

#include <string>

namespace fastx {

struct Record {
    std::string name, comment, seq;
};

class Reader {
public:
    Reader() = default;
    ~Reader() { close(); }
    bool open(const std::string &path);          // "-" = stdin
    void close();
    // >= 0: sequence length; -1: end of file; -2: truncated quality string
    long next(Record &rec);

private:
    int getc_();
    // read until delimiter class: 0 = whitespace, 1 = newline; returns the delimiter or -1
    int until_(int mode, std::string &out);
    gzFile f_ = nullptr;                          // gzipped input and stdin
    int fd_ = -1;                                 // plain files: read(2) into buf_
    unsigned char buf_[1 << 16];
    int begin_ = 0, end_ = 0;
    bool eof_ = false;
    int last_char_ = 0;
};

}  // namespace fastx
