// fastx.cpp — see fastx.h (semantics of kseq.h:67-208, re-implemented over zlib).
#include "fastx.h"

#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>
#include <cstdio>

namespace fastx {

bool Reader::open(const std::string &path)
{
    close();
    begin_ = end_ = 0;
    eof_ = false;
    last_char_ = 0;
    if (path != "-") {
        // plain (not gzipped) files are read with read(2) straight into the parse buffer: zlib's
        // pass-through costs a 512 KiB buffer allocation per file and a copy per byte, which is most
        // of the time of a collection of thousands of small genomes
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        // Only a REGULAR file can be probed and rewound.  A FIFO, a process substitution or
        // /dev/stdin would lose the probed bytes, so the untouched descriptor goes to zlib, which
        // handles both plain and gzipped streams (as the reference's gzopen does).
        struct stat st;
        if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            unsigned char magic[2] = {0, 0};
            const ssize_t got = ::pread(fd, magic, 2, 0);          // does not move the file offset
            if (!(got == 2 && magic[0] == 0x1f && magic[1] == 0x8b)) { fd_ = fd; return true; }
        }
        f_ = gzdopen(fd, "r");                                    // owns fd from here on
        if (!f_) { ::close(fd); return false; }
        gzbuffer(f_, 1 << 18);
        return true;
    }
    f_ = gzdopen(fileno(stdin), "r");
    if (f_) gzbuffer(f_, 1 << 18);
    return f_ != nullptr;
}

void Reader::close()
{
    if (f_) gzclose(f_);
    f_ = nullptr;
    if (fd_ >= 0) ::close(fd_);
    fd_ = -1;
}

int Reader::getc_()
{
    if (begin_ >= end_) {
        if (eof_) return -1;
        begin_ = 0;
        if (fd_ >= 0) {
            // read(2) may return short of a full buffer before the end: only 0 means end of file
            ssize_t n;
            do { n = ::read(fd_, buf_, sizeof buf_); } while (n < 0 && errno == EINTR);
            end_ = n > 0 ? (int)n : 0;
            if (n <= 0) eof_ = true;
        } else {
            end_ = gzread(f_, buf_, sizeof buf_);
            if (end_ < (int)sizeof buf_) eof_ = true;
        }
        if (end_ <= 0) { end_ = 0; return -1; }
    }
    return buf_[begin_++];
}

int Reader::until_(int mode, std::string &out)
{
    out.clear();
    for (;;) {
        const int c = getc_();
        if (c < 0) return -1;
        if (mode == 0 ? isspace(c) : c == '\n') return c;
        out.push_back((char)c);
    }
}

long Reader::next(Record &rec)
{
    int c;
    if (last_char_ == 0) {                         // jump to the next header line
        while ((c = getc_()) != -1 && c != '>' && c != '@') {}
        if (c == -1) return -1;
        last_char_ = c;
    }
    rec.comment.clear();
    rec.seq.clear();
    c = until_(0, rec.name);
    if (c == -1 && rec.name.empty() && begin_ >= end_ && eof_) return -1;
    if (c != '\n' && c != -1) until_(1, rec.comment);
    // sequence bytes: same rule as the byte-by-byte loop of kseq (append isgraph bytes until
    // '>', '+' or '@'), but scanned a buffer at a time: runs of sequence bytes are appended in
    // one go -- this loop is what a multi-gigabyte input spends its time in
    static const struct Classes {
        unsigned char t[256];                      // 0: sequence byte, 1: skipped, 2: terminator
        Classes()
        {
            for (int i = 0; i < 256; i++) t[i] = isgraph(i) ? 0 : 1;
            t[(unsigned char)'>'] = t[(unsigned char)'+'] = t[(unsigned char)'@'] = 2;
        }
    } cls;
    c = -1;
    for (;;) {
        if (begin_ >= end_) {                      // refill
            const int first = getc_();
            if (first < 0) break;
            begin_--;                              // leave it in the buffer
        }
        const unsigned char *b = buf_ + begin_, *e = buf_ + end_, *q = b;
        bool stop = false;
        while (q < e) {
            const unsigned char *run = q;
            while (q < e && cls.t[*q] == 0) q++;
            if (q > run) rec.seq.append(reinterpret_cast<const char *>(run), (size_t)(q - run));
            if (q == e) break;
            if (cls.t[*q] == 2) { c = *q++; stop = true; break; }
            q++;                                   // skipped byte (newline, blank, control)
        }
        begin_ = (int)(q - buf_);
        if (stop) break;
    }
    if (c == '>' || c == '@') last_char_ = c;
    if (c != '+') {
        if (c == -1) last_char_ = 0;
        return (long)rec.seq.size();               // FASTA
    }
    while ((c = getc_()) != -1 && c != '\n') {}    // rest of the '+' line
    if (c == -1) return -2;
    size_t q = 0;
    while (q < rec.seq.size()) {                   // quality bytes, a buffer at a time
        if (begin_ >= end_) {
            if ((c = getc_()) == -1) break;
            begin_--;
        }
        const unsigned char *b = buf_ + begin_, *e = buf_ + end_;
        const size_t need = rec.seq.size() - q;
        while (b < e && q < rec.seq.size()) {
            q += (*b >= 33 && *b <= 127) ? 1 : 0;
            b++;
        }
        (void)need;
        begin_ = (int)(b - buf_);
    }
    // kseq's loop reads a byte BEFORE it tests whether the quality string is complete
    // (`while ((c = ks_getc(ks)) != -1 && qual.l < seq.l)`, kseq.h:201), so one more byte -- the
    // newline of a well-formed file, a header character if there is none -- is consumed
    if (q == rec.seq.size()) (void)getc_();
    last_char_ = 0;
    if (q != rec.seq.size()) return -2;
    return (long)rec.seq.size();
}

}  // namespace fastx
