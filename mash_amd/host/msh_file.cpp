// msh_file.cpp — hand-written Cap'n Proto codec for the MinHash schema (see msh_file.h).
#include "msh_file.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace mshio {

bool use64_for(const std::string &alphabet, bool preserve_case, uint32_t kmer_size, uint32_t *alphabet_size_out)
{
    bool member[256] = {false};
    for (char ch : alphabet) {                       // setAlphabetFromString, Sketch.cpp:1113-1125
        char u = ch;
        if (!preserve_case && u > 96 && u < 123) u -= 32;
        member[(unsigned char)u] = true;
    }
    uint32_t n = 0;
    for (bool m : member) n += m ? 1 : 0;
    if (alphabet_size_out) *alphabet_size_out = n;
    return std::pow((double)n, (double)kmer_size) > std::pow(2.0, 32.0);
}

// ------------------------------------------------------------------------------------------
// reader

namespace {

struct Segments {
    std::vector<const uint64_t *> base;
    std::vector<uint64_t> words;
};

struct ObjRef {            // a resolved pointer
    int kind = -1;         // 0 struct, 1 list, -1 null
    uint32_t seg = 0;
    uint64_t start = 0;    // word index of the content (struct data / list body / composite tag)
    uint32_t data_words = 0, ptr_words = 0;      // struct
    uint32_t elem_code = 0;                      // list
    uint64_t count = 0;                          // list elements (composite: words)
};

struct Reader {
    Segments segs;
    std::string err;

    bool in_bounds(uint32_t seg, uint64_t start, uint64_t nwords) const
    {
        return seg < segs.base.size() && start <= segs.words[seg] && nwords <= segs.words[seg] - start;
    }
    uint64_t word(uint32_t seg, uint64_t idx) const { return segs.base[seg][idx]; }

    // decode the pointer stored at (seg, idx)
    bool resolve(uint32_t seg, uint64_t idx, ObjRef &out)
    {
        if (!in_bounds(seg, idx, 1)) { err = "pointer out of bounds"; return false; }
        uint64_t w = word(seg, idx);
        out = ObjRef();
        if (w == 0) return true;                                   // null
        uint32_t tseg = seg;
        uint64_t after = idx + 1;                                  // offsets are relative to the word after the pointer
        bool have_tag_from_pad = false;
        uint64_t content = 0;
        if ((w & 3) == 2) {                                        // far pointer
            const bool dbl = (w >> 2) & 1;
            const uint64_t pad_off = (w >> 3) & 0x1FFFFFFFull;
            const uint32_t pad_seg = (uint32_t)(w >> 32);
            if (!in_bounds(pad_seg, pad_off, dbl ? 2 : 1)) { err = "far pointer landing pad out of bounds"; return false; }
            if (!dbl) {
                tseg = pad_seg;
                after = pad_off + 1;
                w = word(pad_seg, pad_off);
                if ((w & 3) == 2) { err = "far pointer to far pointer"; return false; }
            } else {
                const uint64_t far2 = word(pad_seg, pad_off);
                const uint64_t tagw = word(pad_seg, pad_off + 1);
                if ((far2 & 3) != 2 || ((far2 >> 2) & 1)) { err = "bad double-far landing pad"; return false; }
                tseg = (uint32_t)(far2 >> 32);
                content = (far2 >> 3) & 0x1FFFFFFFull;
                w = tagw;
                have_tag_from_pad = true;
            }
        }
        const int kind = (int)(w & 3);
        if (kind > 1) { err = "unsupported pointer kind"; return false; }
        if (!have_tag_from_pad) {
            int64_t off = (int64_t)(int32_t)((uint32_t)w & ~3u) >> 2;   // signed 30-bit
            const int64_t c = (int64_t)after + off;
            if (c < 0) { err = "negative pointer target"; return false; }
            content = (uint64_t)c;
        }
        out.kind = kind;
        out.seg = tseg;
        out.start = content;
        if (kind == 0) {
            out.data_words = (uint32_t)((w >> 32) & 0xFFFF);
            out.ptr_words = (uint32_t)((w >> 48) & 0xFFFF);
            if (!in_bounds(tseg, content, (uint64_t)out.data_words + out.ptr_words)) { err = "struct out of bounds"; return false; }
        } else {
            out.elem_code = (uint32_t)((w >> 32) & 7);
            out.count = (w >> 35) & 0x1FFFFFFFull;
            static const uint32_t bits[8] = {0, 1, 8, 16, 32, 64, 64, 0};
            uint64_t nwords = out.elem_code == 7 ? out.count + 1 : (out.count * bits[out.elem_code] + 63) / 64;
            if (!in_bounds(tseg, content, nwords)) { err = "list out of bounds"; return false; }
        }
        return true;
    }

    uint64_t struct_data(const ObjRef &s, uint32_t w) const { return w < s.data_words ? word(s.seg, s.start + w) : 0; }
    bool struct_ptr(const ObjRef &s, uint32_t p, ObjRef &out)
    {
        out = ObjRef();
        if (s.kind != 0 || p >= s.ptr_words) return true;          // absent field = null
        return resolve(s.seg, s.start + s.data_words + p, out);
    }
    bool text(const ObjRef &l, std::string &out)
    {
        out.clear();
        if (l.kind < 0) return true;
        if (l.kind != 1 || l.elem_code != 2) { err = "text field is not a byte list"; return false; }
        if (l.count == 0) return true;
        const char *p = reinterpret_cast<const char *>(segs.base[l.seg] + l.start);
        out.assign(p, p + l.count - 1);                            // drop the NUL
        return true;
    }
};

}  // namespace

std::string parse_msh(const uint8_t *data, size_t size, File &out, bool header_only, uint64_t max_hashes)
{
    out = File();
    if (size < 8 || (size & 7)) return "not a Cap'n Proto message (size)";
    const uint64_t total_words = size / 8;
    const uint64_t *w = reinterpret_cast<const uint64_t *>(data);
    const uint32_t nseg = (uint32_t)(w[0] & 0xFFFFFFFFu) + 1;
    if (nseg > (1u << 20)) return "too many segments";
    const uint64_t hdr_words = ((uint64_t)nseg + 1 + 1) / 2;       // (1 + nseg) u32, padded to 8 bytes
    if (hdr_words > total_words) return "truncated segment table";
    const uint32_t *u = reinterpret_cast<const uint32_t *>(data);
    Reader rd;
    uint64_t pos = hdr_words;
    for (uint32_t i = 0; i < nseg; i++) {
        const uint64_t sz = u[1 + i];
        if (sz > total_words - pos) return "truncated segment";
        rd.segs.base.push_back(w + pos);
        rd.segs.words.push_back(sz);
        pos += sz;
    }
    ObjRef root;
    if (!rd.resolve(0, 0, root) || root.kind != 0) return "bad root pointer: " + rd.err;

    Header &h = out.header;
    const uint64_t w0 = rd.struct_data(root, 0), w1 = rd.struct_data(root, 1), w2 = rd.struct_data(root, 2);
    h.kmer_size = (uint32_t)w0;
    h.window_size = (uint32_t)(w0 >> 32);
    h.sketch_size = (uint32_t)w1;
    h.concatenated = (w1 >> 32) & 1;
    h.noncanonical = (w1 >> 33) & 1;
    h.preserve_case = (w1 >> 34) & 1;
    const uint32_t ebits = (uint32_t)w2;
    memcpy(&h.error, &ebits, 4);
    h.seed = (uint32_t)(w2 >> 32) ^ 42u;                           // default 42 is XORed away on the wire
    ObjRef p_old, p_alpha, p_new;
    if (!rd.struct_ptr(root, 0, p_old) || !rd.struct_ptr(root, 2, p_alpha) || !rd.struct_ptr(root, 3, p_new))
        return "bad root pointers: " + rd.err;
    h.has_alphabet = p_alpha.kind == 1;
    if (!rd.text(p_alpha, h.alphabet)) return rd.err;
    if (!h.has_alphabet) h.alphabet = "ACGT";

    // referenceList if it has references, else referenceListOld (Sketch.cpp:932)
    auto refs_of = [&](const ObjRef &rl, ObjRef &lst) -> bool {
        lst = ObjRef();
        if (rl.kind != 0) return true;
        return rd.struct_ptr(rl, 0, lst);
    };
    ObjRef lst_new, lst_old;
    if (!refs_of(p_new, lst_new) || !refs_of(p_old, lst_old)) return "bad reference list: " + rd.err;
    auto list_elems = [&](const ObjRef &l, uint64_t &n, uint32_t &dw, uint32_t &pw) -> bool {
        n = 0; dw = pw = 0;
        if (l.kind < 0) return true;
        if (l.kind != 1 || l.elem_code != 7) { rd.err = "references is not a struct list"; return false; }
        const uint64_t tag = rd.word(l.seg, l.start);
        n = (tag >> 2) & 0x3FFFFFFFull;
        dw = (uint32_t)((tag >> 32) & 0xFFFF);
        pw = (uint32_t)((tag >> 48) & 0xFFFF);
        if (n * ((uint64_t)dw + pw) > l.count) { rd.err = "struct list overruns"; return false; }
        return true;
    };
    uint64_t n_new = 0, n_old = 0;
    uint32_t dw = 0, pw = 0, dwo = 0, pwo = 0;
    if (!list_elems(lst_new, n_new, dw, pw) || !list_elems(lst_old, n_old, dwo, pwo)) return rd.err;
    ObjRef lst = lst_new;
    uint64_t n = n_new;
    if (n_new == 0) { lst = lst_old; n = n_old; dw = dwo; pw = pwo; }
    h.reference_count = n;

    uint32_t alphabet_size = 0;
    const bool use64 = use64_for(h.alphabet, h.preserve_case, h.kmer_size, &alphabet_size);
    if (!header_only) out.references.resize(n);
    for (uint64_t i = 0; i < n; i++) {
        ObjRef rs;
        rs.kind = 0;
        rs.seg = lst.seg;
        rs.start = lst.start + 1 + i * ((uint64_t)dw + pw);
        rs.data_words = dw;
        rs.ptr_words = pw;
        ObjRef p_counts;
        if (!rd.struct_ptr(rs, 6, p_counts)) return rd.err;
        if (i == 0) h.has_counts = p_counts.kind == 1;             // Sketch.cpp:304
        if (header_only) break;
        Reference &r = out.references[i];
        ObjRef p_name, p_comment, p_h32, p_h64;
        if (!rd.struct_ptr(rs, 2, p_name) || !rd.struct_ptr(rs, 3, p_comment) || !rd.struct_ptr(rs, 4, p_h32) ||
            !rd.struct_ptr(rs, 5, p_h64))
            return rd.err;
        if (!rd.text(p_name, r.name) || !rd.text(p_comment, r.comment)) return rd.err;
        const uint64_t d0 = rd.struct_data(rs, 0), d1 = rd.struct_data(rs, 1);
        r.length = d1 ? d1 : (uint64_t)(uint32_t)d0;               // length64 else length (Sketch.cpp:947-954)
        r.counts_sorted = (d0 >> 32) & 1;
        const ObjRef &hl = use64 ? p_h64 : p_h32;
        uint64_t hn = 0;
        if (hl.kind == 1) {
            const uint32_t want = use64 ? 5u : 4u;
            if (hl.elem_code != want) return "hash list has unexpected element size";
            hn = hl.count;
            if (max_hashes && hn > max_hashes) hn = max_hashes;    // Sketch.cpp:965-968
            r.hashes.resize(hn);
            if (use64) {
                memcpy(r.hashes.data(), rd.segs.base[hl.seg] + hl.start, hn * 8);
            } else {
                const uint32_t *p = reinterpret_cast<const uint32_t *>(rd.segs.base[hl.seg] + hl.start);
                for (uint64_t k = 0; k < hn; k++) r.hashes[k] = p[k];
            }
        }
        if (p_counts.kind == 1) {
            if (p_counts.elem_code != 4) return "counts32 has unexpected element size";
            uint64_t cn = p_counts.count < hn ? p_counts.count : hn;   // reference reads hashCount entries
            r.counts.resize(hn, 0);
            const uint32_t *p = reinterpret_cast<const uint32_t *>(rd.segs.base[p_counts.seg] + p_counts.start);
            for (uint64_t k = 0; k < cn; k++) r.counts[k] = p[k];
        }
    }
    return "";
}

std::string load_file(const std::string &path, std::vector<uint8_t> &buf)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return "could not open \"" + path + "\" for reading.";
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 0) { fclose(f); return "could not get file stats for \"" + path + "\"."; }
    buf.resize((size_t)sz);
    const size_t got = fread(buf.data(), 1, (size_t)sz, f);
    fclose(f);
    if (got != (size_t)sz) return "short read on \"" + path + "\".";
    return "";
}

std::string read_msh(const std::string &path, File &out, bool header_only, uint64_t max_hashes)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return "could not open \"" + path + "\" for reading.";
    std::vector<uint8_t> buf;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz < 0) { fclose(f); return "could not get file stats for \"" + path + "\"."; }
    buf.resize((size_t)sz);
    const size_t got = fread(buf.data(), 1, (size_t)sz, f);
    fclose(f);
    if (got != (size_t)sz) return "short read on \"" + path + "\".";
    return parse_msh(buf.data(), buf.size(), out, header_only, max_hashes);
}

// ------------------------------------------------------------------------------------------
// writer: one segment (objects laid out in the order writeToCapnp creates them) while the
// message fits, several segments with far pointers beyond

namespace {

struct Builder {
    std::vector<uint64_t> w;      // segment words
    bool overflow = false;

    uint64_t alloc(uint64_t n)
    {
        const uint64_t at = w.size();
        w.resize(at + n, 0);
        return at;
    }
    uint64_t off30(uint64_t ptr_at, uint64_t target)
    {
        const int64_t off = (int64_t)target - (int64_t)(ptr_at + 1);
        if (off < -(1ll << 29) || off >= (1ll << 29)) overflow = true;
        return ((uint64_t)off & 0x3FFFFFFFull) << 2;
    }
    void set_struct_ptr(uint64_t at, uint64_t target, uint32_t dw, uint32_t pw)
    {
        w[at] = off30(at, target) | 0ull | ((uint64_t)dw << 32) | ((uint64_t)pw << 48);
    }
    void set_list_ptr(uint64_t at, uint64_t target, uint32_t code, uint64_t count)
    {
        if (count >= (1ull << 29)) overflow = true;
        w[at] = off30(at, target) | 1ull | ((uint64_t)code << 32) | (count << 35);
    }
    void set_text(uint64_t at, const std::string &s)
    {
        if (multi) { far_list(at, 2, s.size() + 1, s.data(), s.size()); return; }
        const uint64_t n = s.size() + 1;                           // with NUL
        const uint64_t t = alloc((n + 7) / 8);
        memcpy(reinterpret_cast<char *>(&w[t]), s.data(), s.size());
        set_list_ptr(at, t, 2, n);
    }

    // ---- multi-segment mode: segment 0 (`w`) keeps the structs, every out-of-line list goes to
    // a data segment and is reached through a far pointer whose landing pad sits right in
    // front of the list (Cap'n Proto encoding: far pointer = {2, pad offset, segment id};
    // the pad is an ordinary list pointer with offset 0)
    bool multi = false;
    uint64_t seg_limit = 0;                                        // words per data segment
    std::vector<std::vector<uint64_t>> data;

    // list of `count` elements of size code `code` whose payload is `bytes` bytes at `src`
    void far_list(uint64_t at, uint32_t code, uint64_t count, const void *src, uint64_t bytes)
    {
        const uint64_t bits = code == 2 ? 8 : code == 4 ? 32 : 64;
        const uint64_t words = (count * bits + 63) / 64;
        if (count >= (1ull << 29) || words + 1 > seg_limit) { overflow = true; return; }
        if (data.empty() || data.back().size() + words + 1 > seg_limit) data.emplace_back();
        std::vector<uint64_t> &d = data.back();
        const uint64_t pad = d.size();
        d.resize(pad + 1 + words, 0);
        d[pad] = 1ull | ((uint64_t)code << 32) | (count << 35);    // list pointer, offset 0
        if (bytes) memcpy(&d[pad + 1], src, bytes);
        if (pad >= (1ull << 29)) { overflow = true; return; }
        w[at] = 2ull | (pad << 3) | ((uint64_t)data.size() << 32); // far pointer -> segment id data.size()
    }
};

}  // namespace

static std::string serialize_impl(const File &in, std::vector<uint64_t> &out, bool multi, uint64_t seg_limit)
{
    const Header &h = in.header;
    const bool use64 = use64_for(h.alphabet, h.preserve_case, h.kmer_size);
    Builder b;
    b.multi = multi;
    b.seg_limit = seg_limit;
    // One allocation for the whole segment (growing by doubling re-copies a 100 MB message twice); in
    // single-segment mode word 0 is the stream frame, so the segment IS the output (pointer offsets are
    // relative: a prefix moves nothing).
    uint64_t est = 64 + in.references.size() * 10;
    for (const Reference &r : in.references)
        est += (r.name.size() + 8) / 8 + (r.comment.size() + 8) / 8 + (multi ? 0 : r.hashes.size() + (r.counts.size() + 1) / 2);
    b.w.reserve(est);
    const uint64_t prefix = multi ? 0 : b.alloc(1);
    const uint64_t rootp = b.alloc(1);
    const uint64_t root = b.alloc(3 + 4);
    b.set_struct_ptr(rootp, root, 3, 4);
    // ReferenceList: referenceListOld (p0) iff seed == 42, else referenceList (p3)  (Sketch.cpp:397)
    const uint64_t rl = b.alloc(1);
    b.set_struct_ptr(root + 3 + (h.seed == 42 ? 0 : 3), rl, 0, 1);
    const uint64_t n = in.references.size();
    const uint64_t lst = b.alloc(1 + n * 9);
    b.set_list_ptr(rl, lst, 7, n * 9);
    b.w[lst] = ((n & 0x3FFFFFFFull) << 2) | (2ull << 32) | (7ull << 48);      // tag: n elements of (2 data, 7 ptr)
    if (n >= (1ull << 29) / 9) return "too many sketches for one .msh segment";
    for (uint64_t i = 0; i < n; i++) {
        const Reference &r = in.references[i];
        const uint64_t s = lst + 1 + i * 9;
        b.set_text(s + 2 + 2, r.name);
        b.set_text(s + 2 + 3, r.comment);
        b.w[s + 1] = r.length;                                     // length64; legacy length stays 0 (Sketch.cpp:407)
        if (!r.hashes.empty()) {
            if (use64) {
                if (multi) {
                    b.far_list(s + 2 + 5, 5, r.hashes.size(), r.hashes.data(), r.hashes.size() * 8);
                } else {
                    const uint64_t t = b.w.size();
                    b.w.insert(b.w.end(), r.hashes.begin(), r.hashes.end());      // (no zero fill first)
                    b.set_list_ptr(s + 2 + 5, t, 5, r.hashes.size());
                }
            } else {
                std::vector<uint32_t> h32(r.hashes.size());
                for (size_t k = 0; k < r.hashes.size(); k++) h32[k] = (uint32_t)r.hashes[k];
                if (multi) {
                    b.far_list(s + 2 + 4, 4, h32.size(), h32.data(), h32.size() * 4);
                } else {
                    const uint64_t t = b.alloc((h32.size() + 1) / 2);
                    memcpy(&b.w[t], h32.data(), h32.size() * 4);
                    b.set_list_ptr(s + 2 + 4, t, 4, h32.size());
                }
            }
            if (!r.counts.empty() && h.has_counts) {               // Sketch.cpp:432-444
                if (multi) {
                    b.far_list(s + 2 + 6, 4, r.counts.size(), r.counts.data(), r.counts.size() * 4);
                } else {
                    const uint64_t t = b.alloc((r.counts.size() + 1) / 2);
                    memcpy(&b.w[t], r.counts.data(), r.counts.size() * 4);
                    b.set_list_ptr(s + 2 + 6, t, 4, r.counts.size());
                }
                b.w[s] |= 1ull << 32;                              // counts32Sorted
            }
        }
    }
    // LocusList with an empty loci list (Sketch.cpp:448-471)
    const uint64_t ll = b.alloc(1);
    b.set_struct_ptr(root + 3 + 1, ll, 0, 1);
    const uint64_t loci = b.alloc(1);
    b.w[loci] = (0ull << 2) | (3ull << 32) | (0ull << 48);         // tag: 0 elements of (3 data, 0 ptr)
    b.set_list_ptr(ll, loci, 7, 0);
    // scalars (Sketch.cpp:473-480)
    uint32_t ebits;
    memcpy(&ebits, &h.error, 4);
    b.w[root + 0] = (uint64_t)h.kmer_size | ((uint64_t)h.window_size << 32);
    b.w[root + 1] = (uint64_t)h.sketch_size | ((uint64_t)(h.concatenated ? 1 : 0) << 32) |
                    ((uint64_t)(h.noncanonical ? 1 : 0) << 33) | ((uint64_t)(h.preserve_case ? 1 : 0) << 34);
    b.w[root + 2] = (uint64_t)ebits | ((uint64_t)(h.seed ^ 42u) << 32);
    b.set_text(root + 3 + 2, h.alphabet);
    if (b.overflow) return multi ? "sketch file too large (a list or the sketch index exceeds a segment)"
                                 : "needs more than one segment";
    if (!multi && b.w.size() - prefix > seg_limit) return "needs more than one segment";
    if (b.w.size() >= (1ull << 29)) return "too many sketches for one .msh file";
    if (!multi) {                                                  // frame: 0 further segments, then this one's size
        b.w[0] = (uint64_t)(b.w.size() - 1) << 32;
        out = std::move(b.w);
        return "";
    }
    // stream framing: u32 (segments - 1), u32 words of every segment, padded to 8 bytes, then the segments
    const uint64_t nseg = 1 + b.data.size();
    std::vector<uint32_t> frame;
    frame.push_back((uint32_t)(nseg - 1));
    frame.push_back((uint32_t)b.w.size());
    for (const auto &d : b.data) frame.push_back((uint32_t)d.size());
    if (frame.size() & 1) frame.push_back(0);
    out.clear();
    out.resize(frame.size() / 2);
    memcpy(out.data(), frame.data(), frame.size() * 4);
    out.insert(out.end(), b.w.begin(), b.w.end());
    for (const auto &d : b.data) out.insert(out.end(), d.begin(), d.end());
    return "";
}

// One segment whenever the message fits (objects in the order writeToCapnp creates them);
// beyond `max_segment_words` (default 2^28 words = 2 GiB, below the 30-bit offset limit of
// in-segment pointers) the lists move to further segments behind far pointers.
std::string serialize_msh(const File &in, std::vector<uint64_t> &out, uint64_t max_segment_words)
{
    if (max_segment_words == 0) max_segment_words = 1ull << 28;
    std::string e = serialize_impl(in, out, false, max_segment_words);
    if (e == "needs more than one segment") e = serialize_impl(in, out, true, max_segment_words);
    return e;
}

std::string write_msh(const std::string &path, const File &in)
{
    std::vector<uint64_t> words;
    std::string e = serialize_msh(in, words);
    if (!e.empty()) return e;
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return "could not open " + path + " for writing.";
    const size_t put = fwrite(words.data(), 8, words.size(), f);
    const int rc = fclose(f);
    if (put != words.size() || rc != 0) return "short write on " + path;
    return "";
}

}  // namespace mshio
