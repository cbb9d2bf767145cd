// mash_main.cpp — `mash sketch | dist | triangle | info | paste | screen` on the MI355X hot path.
//
// Host C++ above the C ABI (include/mashgpu.h).  Option letters, defaults, file naming,
// stdout/stderr text and output order follow the reference commands
// (CommandSketch.cpp:36-169, CommandDistance.cpp:44-304, CommandTriangle.cpp:45-198,
// CommandInfo.cpp:40-299, CommandPaste.cpp:30-89, Command.cpp:165-200,311-347,
// sketchParameterSetup.cpp:15-125, Sketch.cpp:105-253).  All hashing / selection /
// comparison runs on the GPU through libmashgpu; there is no CPU fallback.
// Reads mode: -r, -m <copies>, -c <coverage>, -b <bytes>, -g, -M (exact, see mg_params::min_copies,
// ::bloom_bytes and mg_sketch_reads_host; the Bloom filter of -b has the geometry x86-64 builds of
// the reference use).
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <future>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mashgpu.h"
#include "fastx.h"
#include "msh_file.h"
#include "parse_pool.h"

using std::cerr;
using std::cout;
using std::endl;
using std::string;
using std::vector;

namespace {

using hostpool::ParsedFile;
using hostpool::ParsePool;
using hostpool::Ref;
using hostpool::parse_file_concatenated;

const char *kSuffix = ".msh";
const char *kAlphabetNucleotide = "ACGT";
const char *kAlphabetProtein = "ACDEFGHIKLMNPQRSTVWY";

bool has_suffix(const string &s, const string &suf)
{
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

// ------------------------------------------------------------------------------- options
// Command::Option (Command.h:27-66): numbers are parsed with stof and kept as float.
struct Opt {
    enum Type { Boolean, Number, Integer, Size, File, String } type = Boolean;
    string id, def;
    float lo = 0, hi = 0;
    bool active = false;
    string arg;
    float num = 0;
};

struct Cmd {
    string name;
    std::map<string, Opt> opts;              // by long name
    std::map<string, string> by_id;
    vector<string> args;

    void add(const string &name_, Opt::Type t, const string &id, const string &def = "", float lo = 0, float hi = 0)
    {
        Opt o;
        o.type = t; o.id = id; o.def = def; o.lo = lo; o.hi = hi;
        opts[name_] = o;
        by_id[id] = name_;
        set_arg(opts[name_], def);
    }
    static void set_arg(Opt &o, string a)
    {
        o.arg = a;
        if (o.type == Opt::Number || o.type == Opt::Integer) {
            if (a.empty()) { o.num = 0; return; }
            bool failed = false;
            try {
                o.num = std::stof(a);
                if (o.lo != o.hi && (o.num < o.lo || o.num > o.hi)) failed = true;
                else if (o.type == Opt::Integer && (float)(uint64_t)o.num != o.num) failed = true;
            } catch (const std::exception &) { failed = true; }
            if (failed) {
                cerr << "ERROR: Argument to -" << o.id << " must be a" << (o.type == Opt::Integer ? "n integer" : " number");
                if (o.lo != o.hi) cerr << " between " << o.lo << " and " << o.hi;
                cerr << " (" << a << " given)" << endl;
                exit(1);
            }
        } else if (o.type == Opt::Size) {
            if (a.empty()) { o.num = 0; return; }
            char suffix = a[a.size() - 1];
            double factor = 1;
            if (suffix < '0' || suffix > '9') {
                switch (suffix) {
                    case 'k': case 'K': factor = 1e3; break;
                    case 'm': case 'M': factor = 1e6; break;
                    case 'g': case 'G': factor = 1e9; break;
                    case 't': case 'T': factor = 1e12; break;
                    default:
                        cerr << "ERROR: Unrecognized unit (\"" << suffix << "\") in argument to -" << o.id
                             << ". If specified, unit must be one of [kKmMgGtT]." << endl;
                        exit(1);
                }
                a.resize(a.size() - 1);
            }
            bool fail = false;
            try { o.num = std::stof(a); } catch (const std::exception &) { fail = true; }
            if (o.num <= 0 || (float)(uint64_t)o.num != o.num) fail = true;
            if (fail) {
                cerr << "ERROR: Argument to -" << o.id << " must be a whole number, optionally followed by one of [kKmMgGtT]." << endl;
                exit(1);
            }
            o.num = (float)(o.num * factor);
        }
    }
    // Command::run(argc, argv), Command.cpp:311-347
    int parse(int argc, const char **argv)
    {
        for (int i = 0; i < argc; i++) {
            if (argv[i][0] == '-' && argv[i][1] != 0) {
                if (!by_id.count(argv[i] + 1)) { cerr << "ERROR: Unrecognized option: " << argv[i] << endl; return 1; }
                Opt &o = opts.at(by_id.at(argv[i] + 1));
                o.active = true;
                if (o.type != Opt::Boolean) {
                    i++;
                    if (i == argc) { cerr << "ERROR: -" << o.id << " requires an argument" << endl; return 1; }
                    set_arg(o, argv[i]);
                }
            } else {
                args.push_back(argv[i]);
            }
        }
        return 0;
    }
    const Opt &o(const string &n) const { return opts.at(n); }
    bool has(const string &n) const { return opts.count(n) != 0; }
    void use_sketch_options()                  // Command::useSketchOptions, Command.cpp:354-379
    {
        add("threads", Opt::Integer, "p", "1");
        add("kmer", Opt::Integer, "k", "21", 1, 32);
        add("noncanonical", Opt::Boolean, "n");
        add("protein", Opt::Boolean, "a");
        add("alphabet", Opt::String, "z");
        add("case", Opt::Boolean, "Z");
        add("sketchSize", Opt::Integer, "s", "1000");
        add("individual", Opt::Boolean, "i");
        add("seed", Opt::Integer, "S", "42", 0, (float)0xFFFFFFFF);
        add("warning", Opt::Number, "w", "0.01", 0, 1);
        add("reads", Opt::Boolean, "r");
        add("memory", Opt::Size, "b");
        add("minCov", Opt::Integer, "m", "1");
        add("targetCov", Opt::Number, "c");
        add("genome", Opt::Size, "g");
    }
};

// ------------------------------------------------------------------------------- parameters
struct Params {                               // Sketch::Parameters (Sketch.h:34-106), fields used here
    int kmer = 0;
    uint64_t sketch_size = 0;
    uint32_t seed = 0;
    bool concatenated = false, noncanonical = false, preserve_case = false, reads = false, counts = false;
    float warning = 0;
    uint64_t genome_size = 0;
    uint32_t min_copies = 1;                  // minCov (-m)
    bool never_admit = false;                 // -m 0: the reference's heap admits no hash at all (an empty sketch)
    double target_cov = 0;                    // targetCov (-c)
    uint64_t bloom_bytes = 0;                 // memoryBound (-b)
    int threads = 1;                          // -p: files parsed concurrently (the GPU does the sketching)
    string alphabet;                          // normalised (uppercased unless preserve_case), sorted
    uint32_t alphabet_size = 0;
    bool use64 = false;
};

string normalise_alphabet(const string &chars, bool preserve_case)
{
    bool m[256] = {false};
    for (char c : chars) {
        char u = c;
        if (!preserve_case && u > 96 && u < 123) u -= 32;
        m[(unsigned char)u] = true;
    }
    string out;                               // Sketch::getAlphabetAsString: ascending byte order
    for (int i = 0; i < 256; i++) if (m[i]) out.push_back((char)i);
    return out;
}

void set_alphabet(Params &p, const string &chars)
{
    p.alphabet = normalise_alphabet(chars, p.preserve_case);
    p.use64 = mshio::use64_for(p.alphabet, p.preserve_case, (uint32_t)p.kmer, &p.alphabet_size);
}

// sketchParameterSetup, sketchParameterSetup.cpp:15-105
int sketch_parameter_setup(Params &p, const Cmd &c)
{
    p.kmer = (int)c.o("kmer").num;
    p.sketch_size = (uint64_t)c.o("sketchSize").num;
    p.concatenated = !c.o("individual").active;
    p.noncanonical = c.o("noncanonical").active;
    p.seed = (uint32_t)c.o("seed").num;
    p.reads = c.o("reads").active;
    p.preserve_case = c.o("case").active;
    p.threads = std::max(1, (int)c.o("threads").num);
    if (c.has("warning")) p.warning = c.o("warning").num;
    if (c.o("memory").active || c.o("minCov").active || c.o("targetCov").active) {
        if (c.o("memory").active && c.o("minCov").active) {          // sketchParameterSetup.cpp:44-48
            cerr << "ERROR: The option " << c.o("minCov").id << " cannot be used with " << c.o("memory").id << "." << endl;
            return 1;
        }
        if (c.o("memory").active) p.bloom_bytes = (uint64_t)c.o("memory").num;      // sketchParameterSetup.cpp:42
        p.reads = true;
        if (c.o("targetCov").active) p.target_cov = c.o("targetCov").num;          // sketchParameterSetup.cpp:24
        if (c.o("minCov").num >= 1) p.min_copies = (uint32_t)c.o("minCov").num;   // Sketch.cpp:1156 (reads mode only)
        // -m 0: multiplicityMinimum - 1 wraps (uint64_t), no pending count ever equals it, so the
        // reference's heap admits nothing (MinHashHeap.cpp:96-118): an empty sketch
        else if (c.o("minCov").active) p.never_admit = true;
    }
    if (c.o("genome").active) { p.reads = true; p.genome_size = (uint64_t)c.o("genome").num; }
    if (p.reads) p.counts = true;
    if (p.reads && c.o("threads").active)
        cerr << "WARNING: The option " << c.o("threads").id << " will be ignored with " << c.o("reads").id << "." << endl;
    if (p.reads && !p.concatenated) {
        cerr << "ERROR: The option " << c.o("individual").id << " cannot be used with " << c.o("reads").id << "." << endl;
        return 1;
    }
    if (c.o("protein").active) {
        p.noncanonical = true;
        if (!c.o("kmer").active) p.kmer = 9;
        set_alphabet(p, kAlphabetProtein);
    } else if (c.o("alphabet").active) {
        p.noncanonical = true;
        set_alphabet(p, c.o("alphabet").arg);
    } else {
        set_alphabet(p, kAlphabetNucleotide);
    }
    return 0;
}

void split_file(const string &file, vector<string> &lines)      // Command.cpp:398-414
{
    std::ifstream in(file);
    if (in.fail()) { cerr << "ERROR: Could not open " << file << ".\n"; exit(1); }
    string line;
    while (getline(in, line)) lines.push_back(line);
}

// ------------------------------------------------------------------------------- sketch set
struct SketchSet {                            // the part of class Sketch the commands use
    Params p;
    vector<Ref> refs;
    double kmer_space() const { return std::pow((double)p.alphabet_size, (double)p.kmer); }   // Sketch.cpp:509
};

// One device by default; more on request (MASH_GPU_DEVICES=all or "0,1,...", MASH_GPU_DEVICE=<n> picks the one):
// a local communicator -- one context per device, RCCL between them (MASH_GPU_DEVICES=all or a list;
// one device, device 0 or MASH_GPU_DEVICE, without it).  `ctx` is the first
// device's context: sketching and screening run there, `dist` and `triangle` shard their row
// blocks over all of them (mg_compare_*_sharded_host; the reference fans the same loops out to
// its -p threads, CommandTriangle.cpp:129-139, CommandDistance.cpp:195-232).
struct Gpu {
    mg_comm *comm = nullptr;
    mg_ctx *ctx = nullptr;
    Gpu()
    {
        vector<int> devs;
        if (const char *e = getenv("MASH_GPU_DEVICE")) devs.push_back(atoi(e));
        else if (const char *l = getenv("MASH_GPU_DEVICES")) {
            // "all": every visible GPU; else a comma separated list.  Without the variable ONE device is
            // used: more GPUs mean a communicator (RCCL start-up: seconds) that small jobs never earn
            // back, so the caller asks for them.
            if (strcmp(l, "all") == 0) {
                const int n = mg_device_count();
                for (int i = 0; i < n; i++) devs.push_back(i);
                l = "";
            }
            for (const char *q = l; *q;) {
                char *end;
                const long v = strtol(q, &end, 10);
                if (end == q) break;
                devs.push_back((int)v);
                q = *end == ',' ? end + 1 : end;
            }
        }
        if (devs.empty()) devs.push_back(0);
        if (mg_comm_create_local(devs.data(), (int)devs.size(), &comm) != MG_OK) {
            cerr << "ERROR: no usable GPU: " << mg_comm_last_error(nullptr) << endl;
            exit(1);
        }
        ctx = mg_comm_ctx(comm, 0);
    }
    ~Gpu() { mg_comm_destroy(comm); }
};

void params_from_header(Params &p, const mshio::Header &h)     // initParametersFromCapnp, Sketch.cpp:255-324
{
    p.kmer = (int)h.kmer_size;
    p.sketch_size = h.sketch_size;
    p.concatenated = h.concatenated;
    p.noncanonical = h.noncanonical;
    p.preserve_case = h.preserve_case;
    p.counts = h.has_counts;
    p.seed = h.seed;
    set_alphabet(p, h.has_alphabet ? h.alphabet : string(kAlphabetNucleotide));
}

// "Sketching <file>..." lines: one write(2) per file is what the consumer of 12 000 small genomes would
// spend a tenth of its time in.  Lines are gathered and written when 4 KiB have accumulated, before the
// consumer sleeps, and before anything else goes to stderr.
struct ProgressLog {
    string buf;
    void line(const string &l)
    {
        buf += l;
        if (buf.size() >= 4096) flush();
    }
    void flush()
    {
        if (buf.empty()) return;
        cerr << buf << std::flush;
        buf.clear();
    }
} g_progress;

// One batch of inputs -> GPU -> hash lists appended to `set`.
// Two modes: `stream` (files -> sketches): the bytes go to a mg_sketch_session as they are parsed --
// packed into pinned staging buffers and copied to the device while parsing goes on, no
// concatenated copy on the host; host mode (reads mode, whose -c replay needs the bytes): `bases`.
struct PendingBatch {
    vector<uint8_t> bases;
    vector<uint64_t> off{0};
    vector<Ref> refs;
    bool stream = false;
    mg_sketch_session *sess = nullptr;        // stream mode, created at the first byte (parameters are final by then)
    uint64_t nbytes = 0;
    // flush threshold in bytes: with parse workers the batches are SMALL, so that sketching, copying back and
    // handing out the hashes of one batch run while the workers parse the next (init_from_files)
    uint64_t max_bytes = 2ull << 30;
    // result buffers of mg_sketch_*: kept over the batches of a run (a fresh 100 MB vector is 25 000 page faults)
    std::unique_ptr<uint64_t[]> out_hashes;
    std::unique_ptr<uint32_t[]> out_nhash, out_counts;
    uint64_t out_cap = 0, out_cap_n = 0;
    // fn(0..n-1), possibly on several threads (the parse workers, when there are any)
    std::function<void(size_t, const std::function<void(size_t)> &)> parallel_for =
        [](size_t n, const std::function<void(size_t)> &fn) { for (size_t i = 0; i < n; i++) fn(i); };
    void append(const uint8_t *p, size_t n)
    {
        if (sess) {
            if (mg_sketch_add(sess, p, n) != MG_OK) { cerr << "ERROR: could not stage input for the GPU" << endl; exit(1); }
        } else {
            bases.insert(bases.end(), p, p + n);
        }
        nbytes += n;
    }
    void add_record(const string &seq)
    {
        append(reinterpret_cast<const uint8_t *>(seq.data()), seq.size());
        const uint8_t sep = (uint8_t)MG_RECORD_SEP;
        append(&sep, 1);
    }
    void end_sketch(Ref &&r)
    {
        if (sess) mg_sketch_end_sketch(sess);
        off.push_back(nbytes);
        refs.push_back(std::move(r));
    }
    void reset()
    {
        bases.clear();
        off.assign(1, 0);
        refs.clear();
        nbytes = 0;
    }
};

// MASH_AMD_TIMING=1: wall time of the stages of a run, to stderr at exit
struct StageClock {
    const bool on = getenv("MASH_AMD_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    vector<std::pair<string, double>> acc;
    void lap(const char *name)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        const double dt = std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
        for (auto &a : acc) if (a.first == name) { a.second += dt; return; }
        acc.emplace_back(name, dt);
    }
    ~StageClock()
    {
        if (!on) return;
        lap("teardown");                             // (declared before the Gpu: its destruction is in here)
        cerr << "timing:";
        for (auto &a : acc) cerr << ' ' << a.first << ' ' << a.second << " s;";
        cerr << endl;
    }
};

double g_gpu_sketch_seconds = 0;            // time inside mg_sketch_host (MASH_AMD_TIMING)

static mg_params batch_params(const SketchSet &set)
{
    mg_params mp;
    mg_params_init(&mp, set.p.kmer, set.p.sketch_size, set.p.seed, set.p.alphabet.c_str(), set.p.noncanonical,
                   set.p.preserve_case);
    if (set.p.reads) mp.min_copies = set.p.min_copies;
    return mp;
}

// stream mode: the session is opened when the first sketch of a batch arrives
void ensure_session(Gpu &gpu, const SketchSet &set, PendingBatch &b)
{
    if (!b.stream || b.sess) return;
    const mg_params mp = batch_params(set);
    if (mg_sketch_begin(gpu.ctx, &mp, &b.sess) != MG_OK) { cerr << "ERROR: " << mg_last_error(gpu.ctx) << endl; exit(1); }
}

void flush_batch(Gpu &gpu, SketchSet &set, PendingBatch &b)
{
    if (b.refs.empty()) return;
    const mg_params mp = batch_params(set);
    const uint64_t n = b.refs.size(), s = set.p.sketch_size;
    if (n * s > b.out_cap || n > b.out_cap_n) {
        b.out_cap = std::max<uint64_t>(n * s, b.out_cap);
        b.out_cap_n = std::max<uint64_t>(n, b.out_cap_n);
        b.out_hashes.reset(new uint64_t[b.out_cap]);
        b.out_nhash.reset(new uint32_t[b.out_cap_n]);
        if (set.p.counts) b.out_counts.reset(new uint32_t[b.out_cap]);
    }
    uint64_t *hashes = b.out_hashes.get();
    uint32_t *nhash = b.out_nhash.get(), *counts = set.p.counts ? b.out_counts.get() : nullptr;
    if (!b.sess && b.bases.empty()) b.bases.push_back((uint8_t)MG_RECORD_SEP);
    const auto t_gpu = std::chrono::steady_clock::now();
    // (several GPUs, MASH_GPU_DEVICES: the batch is cut into byte-balanced blocks of sketches, one per device)
    const int sk_rc = b.sess ? mg_sketch_finish(b.sess, hashes, nhash, counts)
                      : mg_comm_size(gpu.comm) > 1 && mp.min_copies <= 1
                          ? mg_sketch_sharded_host(gpu.comm, &mp, b.bases.data(), b.bases.size(), b.off.data(), n, hashes, nhash, counts)
                          : mg_sketch_host(gpu.ctx, &mp, b.bases.data(), b.bases.size(), b.off.data(), n, hashes, nhash, counts);
    g_gpu_sketch_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_gpu).count();
    if (sk_rc != MG_OK) {
        g_progress.flush();
        const char *m = mg_comm_last_error(gpu.comm);
        cerr << "ERROR: " << ((m && *m) ? m : mg_last_error(gpu.ctx)) << endl;
        exit(1);
    }
    const size_t piece = 64;                                 // sketches per job
    b.parallel_for((n + piece - 1) / piece, [&](size_t j) {
        for (uint64_t i = j * piece; i < std::min<uint64_t>(n, (j + 1) * piece); i++) {
            b.refs[i].hashes.assign(hashes + i * s, hashes + i * s + nhash[i]);
            if (counts) b.refs[i].counts.assign(counts + i * s, counts + i * s + nhash[i]);
        }
    });
    for (uint64_t i = 0; i < n; i++) set.refs.push_back(std::move(b.refs[i]));
    b.reset();
}

const uint64_t kBatchBytes = 2ull << 30;
// ... and once it holds this many hash slots (sketches x sketch size): the hash, count and
// first-position arrays of a batch are n*s*8 + n*s*4 + n*s*8 bytes on the host and again on the
// device, so a file of a million short records at -s 10000 must not become one batch (the
// reference streams such a file in constant memory)
const uint64_t kBatchHashes = 1ull << 27;
static bool batch_full(const PendingBatch &b, uint64_t sketch_size)
{
    static const uint64_t cap = [] {                       // MASH_AMD_BATCH_HASHES: test knob
        const char *e = getenv("MASH_AMD_BATCH_HASHES");
        return e ? std::max<uint64_t>(1, strtoull(e, nullptr, 10)) : kBatchHashes;
    }();
    return b.nbytes > std::min(kBatchBytes, b.max_bytes) || (uint64_t)b.refs.size() * sketch_size > cap;
}

void queue_parsed_file(Gpu &gpu, SketchSet &set, PendingBatch &b, ParsedFile &&pf)
{
    if (!pf.error.empty()) { g_progress.flush(); cerr << pf.error << endl; exit(1); }
    ensure_session(gpu, set, b);
    b.append(pf.bases.data(), pf.bases.size());
    b.end_sketch(std::move(pf.ref));
    if (batch_full(b, set.p.sketch_size)) flush_batch(gpu, set, b);
}

// sketchFileBySequence (Sketch.cpp:326-370): one sketch per record
void queue_file_by_sequence(Gpu &gpu, SketchSet &set, PendingBatch &b, const string &file)
{
    fastx::Reader rd;
    if (!rd.open(file)) { cerr << "ERROR: could not open " << file << " for reading." << endl; exit(1); }
    fastx::Record rec;
    long l;
    while ((l = rd.next(rec)) >= 0) {
        if (l < set.p.kmer) continue;
        Ref ref;
        ref.name = rec.name;
        ref.comment = rec.comment;
        ref.length = (uint64_t)l;
        ensure_session(gpu, set, b);
        b.add_record(rec.seq);
        b.end_sketch(std::move(ref));
        if (batch_full(b, set.p.sketch_size)) flush_batch(gpu, set, b);
    }
    if (l != -1) { cerr << "\nERROR: reading " << file << "." << endl; exit(1); }
}

// Sketch::initFromReads -> sketchFile over all files, round robin (Sketch.cpp:96-103, :1147-1336)
// `shared`: a streamed-ingest session kept by the caller across calls (one reads-mode query file after the
// other in `mash dist -r`): its pinned staging buffers are allocated once, not per file
void sketch_reads(Gpu &gpu, SketchSet &set, const vector<string> &files, mg_sketch_session **shared = nullptr,
                  mg_reads_session **shared_reads = nullptr)
{
    vector<fastx::Reader *> readers;
    Ref ref;
    for (size_t f = 0; f < files.size(); f++) {
        if (files[f] == "-" && f > 1) { cerr << "ERROR: '-' for stdin must be first input" << endl; exit(1); }
        if (ref.name.empty() && files[f] != "-") ref.name = files[f];
        fastx::Reader *r = new fastx::Reader;
        if (!r->open(files[f])) { cerr << "ERROR: could not open " << files[f] << endl; exit(1); }
        readers.push_back(r);
    }
    PendingBatch b;
    // Every reads option goes through a reads session, chunk by chunk: the device hashes a chunk and hands
    // back the k-mers that can still change the heap, the host replays MinHashHeap::tryInsert over that
    // thinned stream (-m pending set, -b Bloom filter, -c stop test included) -- neither host nor device ever
    // holds more than a chunk or two, as the reference's reader loop holds one record and its heap
    // (Sketch.cpp:1196-1270).  With -c the reading STOPS with the chunk that reaches the target coverage
    // (Sketch.cpp:1258).  A chunk is on the device while the next one is being parsed.
    // MASH_AMD_READS_RESIDENT=1 (tests; plain -r / -m only): the round-2 route -- the reads leave for the device as
    // they are parsed and ONE sketch call runs over the whole read set in HBM at the end; same bytes out.
    const bool none = set.p.never_admit;          // -m 0: every record is read and counted, no hash is kept
    const bool must_replay = set.p.target_cov > 0 || set.p.bloom_bytes > 0;      // order-dependent by definition
    const bool cov_mode = !none && (must_replay || !getenv("MASH_AMD_READS_RESIDENT"));
    b.stream = !none && !cov_mode && !getenv("MASH_AMD_NO_STREAM");
    if (b.stream && shared) b.sess = *shared;
    ensure_session(gpu, set, b);
    mg_reads_session *rs = nullptr;
    size_t reads_chunk = 64u << 20;
    if (const char *e = getenv("MASH_AMD_READS_CHUNK")) reads_chunk = std::max<size_t>(1, strtoull(e, nullptr, 10));   // test knob
    int cov_stopped = 0;                          // written by this thread only, in wait_chunk
    int task_stopped = 0;                         // written by the task; read after its future.get()
    vector<uint8_t> in_flight;                    // the chunk the device is working on
    std::future<int> pending;
    auto wait_chunk = [&]() {
        if (!pending.valid()) return;
        if (pending.get() != MG_OK) { cerr << "ERROR: " << mg_last_error(gpu.ctx) << endl; exit(1); }
        cov_stopped = task_stopped;
    };
    auto feed_chunk = [&]() {
        wait_chunk();                             // (cov_stopped now tells about the chunk before this one)
        if (b.bases.empty() || cov_stopped) { b.bases.clear(); return; }
        in_flight.swap(b.bases);
        b.bases.clear();
        pending = std::async(std::launch::async, [&]() { return mg_reads_add_host(rs, in_flight.data(), in_flight.size(), &task_stopped); });
    };
    if (cov_mode) {
        mg_params mp = batch_params(set);
        mp.min_copies = set.p.min_copies;
        mp.target_cov = set.p.target_cov;
        mp.bloom_bytes = set.p.bloom_bytes;
        // (MASH_AMD_SHARED_READS=1: one session for all reads-mode files of a run, emptied by mg_reads_reset between
        //  them -- its device buffers are then allocated once; opt-in until it has run on the GPU)
        if (shared_reads && *shared_reads) rs = *shared_reads;
        else if (mg_reads_begin(gpu.ctx, &mp, &rs) != MG_OK) { cerr << "ERROR: " << mg_last_error(gpu.ctx) << endl; exit(1); }
        b.bases.reserve(std::min<size_t>(reads_chunk, 64u << 20) + (1u << 16));
    }
    fastx::Record rec;
    size_t it = 0;
    long l = -1;
    int count = 0;
    bool skipped = false;
    while (!readers.empty()) {
        l = readers[it]->next(rec);
        if (l < -1) break;
        if (l == -1) {
            delete readers[it];
            readers.erase(readers.begin() + it);
            if (it == readers.size()) it = 0;
            continue;
        }
        if (l >= set.p.kmer) {
            if (count == 0) {
                if (files[0] == "-") { ref.name = rec.name; ref.comment = rec.comment; }
                else ref.comment = rec.name + " " + rec.comment;
            }
            count++;
            if (!none) b.add_record(rec.seq);
            it++;
            if (it == readers.size()) it = 0;
            if (cov_mode && b.bases.size() >= reads_chunk) {
                feed_chunk();
                if (cov_stopped) { l = -1; break; }            // target coverage reached: the rest is not read
            }
        }
        else skipped = true;                                   // skipped without advancing to the next file (Sketch.cpp:1222-1226)
    }
    for (auto *r : readers) delete r;
    if (l != -1) { cerr << "\nERROR: reading input files." << endl; exit(1); }
    // (an empty result is refused below, once the length is known: Sketch.cpp:1302-1314)
    auto refuse_empty = [&]() {
        if (skipped) cerr << "\nWARNING: All fasta records in input files were shorter than the k-mer size (" << set.p.kmer << ")." << endl;
        else cerr << "\nERROR: Did not find fasta records in \"input files\"." << endl;
        exit(1);
    };
    if (count == 0 && set.p.genome_size == 0) refuse_empty();
    uint64_t reads_used = (uint64_t)count;
    auto wrap_comment = [&](uint64_t n) {                   // "[N seqs] first [...]" (Sketch.cpp:1284-1292)
        if (n > 1) ref.comment = "[" + std::to_string(n) + " seqs] " + ref.comment + " [...]";
    };
    if (cov_mode) {
        // the sequential heap, replayed exactly; with -c it also decided where reading stopped (Sketch.cpp:1258)
        feed_chunk();
        wait_chunk();
        const uint64_t s = set.p.sketch_size;
        vector<uint64_t> hashes(s);
        vector<uint32_t> counts(s);
        uint32_t nh = 0;
        if (mg_reads_finish(rs, hashes.data(), &nh, counts.data(), &reads_used) != MG_OK) {
            cerr << "ERROR: " << mg_last_error(gpu.ctx) << endl;
            exit(1);
        }
        if (shared_reads && getenv("MASH_AMD_SHARED_READS")) { mg_reads_reset(rs); *shared_reads = rs; }
        else mg_reads_free(rs);
        ref.hashes.assign(hashes.begin(), hashes.begin() + nh);
        ref.counts.assign(counts.begin(), counts.begin() + nh);
        wrap_comment(reads_used);                              // the reference counts the reads it consumed
        set.refs.push_back(std::move(ref));
    } else if (none) {
        wrap_comment(reads_used);
        set.refs.push_back(std::move(ref));
    } else {
        wrap_comment(reads_used);
        b.end_sketch(std::move(ref));
        flush_batch(gpu, set, b);
    }
    if (b.sess && shared) *shared = b.sess;               // empty again after its finish: the next file fills it
    else if (b.sess) mg_sketch_session_free(b.sess);
    Ref &r = set.refs.back();
    // estimateSetSize (MinHashHeap.h:45): 2^bits * n / max kept hash
    double est = 0;
    if (!r.hashes.empty())
        est = std::pow(2.0, set.p.use64 ? 64.0 : 32.0) * (double)r.hashes.size() / (double)r.hashes.back();
    r.length = set.p.genome_size ? set.p.genome_size : (uint64_t)est;
    // nothing kept (e.g. -m 2 on a genome without repeats) and no -g: length 0 is "no records" (Sketch.cpp:1302-1314)
    if (r.length == 0) refuse_empty();
    cerr << "Estimated genome size: " << est << endl;
    double msum = 0;                                       // estimateMultiplicity (MinHashHeap.h:44)
    for (uint32_t c : r.counts) msum += c;
    cerr << "Estimated coverage:    " << (r.counts.empty() ? 0.0 : msum / (double)r.counts.size()) << endl;
    if (set.p.target_cov > 0) cerr << "Reads used:            " << reads_used << endl;      // Sketch.cpp:1324-1327
}

// the .msh branch of Sketch::initFromFiles (Sketch.cpp:120-172) + loadCapnp
bool load_msh_into(SketchSet &set, const string &file, bool first_sets_params, bool contain = false)
{
    mshio::File hdr;
    vector<uint8_t> image;                                       // read once, parsed twice (header, then lists)
    string e = mshio::load_file(file, image);
    if (e.empty()) e = mshio::parse_msh(image.data(), image.size(), hdr, true, 0);
    if (!e.empty()) { cerr << "ERROR: " << e << endl; exit(1); }
    if (first_sets_params) params_from_header(set.p, hdr.header);
    Params t;
    params_from_header(t, hdr.header);
    if (set.p.alphabet != t.alphabet) {
        cerr << "\nWARNING: The sketch file " << file << " has different alphabet (" << t.alphabet << ") than the current alphabet (" << set.p.alphabet << "). This file will be skipped." << endl << endl;
        return false;
    }
    if (t.seed != set.p.seed) {
        cerr << "\nWARNING: The sketch " << file << " has a seed size (" << t.seed << ") that does not match the current seed (" << set.p.seed << "). This file will be skipped." << endl << endl;
        return false;
    }
    if (t.kmer != set.p.kmer) {
        cerr << "\nWARNING: The sketch " << file << " has a kmer size (" << t.kmer << ") that does not match the current kmer size (" << set.p.kmer << "). This file will be skipped." << endl << endl;
        return false;
    }
    if (!contain && t.sketch_size < set.p.sketch_size) {
        cerr << "\nWARNING: The sketch file " << file << " has a target sketch size (" << t.sketch_size << ") that is smaller than the current sketch size (" << set.p.sketch_size << "). This file will be skipped." << endl << endl;
        return false;
    }
    if (t.noncanonical != set.p.noncanonical) {
        cerr << "\nWARNING: The sketch file " << file << " is " << (t.noncanonical ? "noncanonical" : "canonical") << ", which is incompatible with the current setting. This file will be skipped." << endl << endl;
        return false;
    }
    if (t.sketch_size > set.p.sketch_size)
        cerr << "\nWARNING: The sketch file " << file << " has a target sketch size (" << t.sketch_size << ") that is larger than the current sketch size (" << set.p.sketch_size << "). Its sketches will be reduced." << endl << endl;
    mshio::File f;
    e = mshio::parse_msh(image.data(), image.size(), f, false, set.p.sketch_size);
    if (!e.empty()) { cerr << "ERROR: " << e << endl; exit(1); }
    for (auto &r : f.references) {
        Ref x;
        x.name = std::move(r.name);
        x.comment = std::move(r.comment);
        x.length = r.length;
        x.hashes = std::move(r.hashes);
        x.counts = std::move(r.counts);
        set.refs.push_back(std::move(x));
    }
    return true;
}

bool parseable_by_pool(const vector<string> &files, size_t i, const Params &p)
{
    return p.concatenated && !(p.reads && p.concatenated) && !has_suffix(files[i], kSuffix) && files[i] != "-";
}

// A run of parsed files goes to the device as ONE window of the session's pinned staging buffer
// (mg_sketch_stage): the copies into it are dealt to the parse workers, the consumer only keeps the books.
void queue_parsed_group(Gpu &gpu, SketchSet &set, PendingBatch &b, vector<ParsedFile> &group, ParsePool &pool)
{
    ensure_session(gpu, set, b);
    uint64_t total = 0;
    for (const ParsedFile &pf : group) total += pf.bases.size();
    uint8_t *win = nullptr;
    if (mg_sketch_stage(b.sess, total, &win) != MG_OK) { g_progress.flush(); cerr << "ERROR: " << mg_last_error(gpu.ctx) << endl; exit(1); }
    vector<ParsePool::Copy> jobs;
    uint64_t at = 0;
    for (const ParsedFile &pf : group) {
        // (pieces of <= 1 MiB, so that one large genome is copied by several workers too)
        for (uint64_t o = 0; o < pf.bases.size(); o += 1 << 20)
            jobs.push_back({win + at + o, pf.bases.data() + o, (size_t)std::min<uint64_t>(1 << 20, pf.bases.size() - o)});
        at += pf.bases.size();
    }
    pool.copy_all(jobs);
    for (ParsedFile &pf : group) {
        if (mg_sketch_commit(b.sess, pf.bases.size()) != MG_OK) { g_progress.flush(); cerr << "ERROR: " << mg_last_error(gpu.ctx) << endl; exit(1); }
        b.nbytes += pf.bases.size();
        b.end_sketch(std::move(pf.ref));
    }
    group.clear();
    if (batch_full(b, set.p.sketch_size)) flush_batch(gpu, set, b);
}

void init_from_files(Gpu &gpu, SketchSet &set, const vector<string> &files, const Params &p, int verbosity = 1,
                     bool enforce_parameters = false, ParsePool *early = nullptr)
{
    set.p = p;
    PendingBatch b;
    mg_sketch_session *reads_sess = nullptr;               // shared by the reads-mode query files
    mg_reads_session *reads_rs = nullptr;
    // (env: the concatenate-then-copy path, for tests; several GPUs: whole batches, cut over the devices)
    b.stream = !getenv("MASH_AMD_NO_STREAM") && mg_comm_size(gpu.comm) <= 1;
    // concatenated mode with -p > 1: files are parsed ahead by a pool of workers (ParsePool)
    // Reads options (-r -m -c -b -g) reach sketchFile through here too (`mash dist -r ref.msh reads.fq`,
    // `mash triangle -r ...`): ONE sketch per file with the reads-mode heap, the estimated (or -g)
    // genome size as its length and the two "Estimated ..." lines (Sketch.cpp:1156, :1272-1282,
    // :1320-1330) -- the same code path as `mash sketch -r` over that one file.
    const bool reads_files = set.p.reads && set.p.concatenated;
    auto parseable = [&](size_t i) { return parseable_by_pool(files, i, set.p); };
    std::unique_ptr<ParsePool> own;
    ParsePool *pool = early;
    if (!pool && p.threads > 1 && !reads_files) { own.reset(new ParsePool(files, (size_t)p.threads, parseable)); pool = own.get(); }
    const bool grouped = pool && b.stream && !getenv("MASH_AMD_NO_GROUPS");
    if (pool) {
        b.parallel_for = [pool](size_t n, const std::function<void(size_t)> &fn) { pool->run_jobs(n, fn); };
        if (b.stream) {
            const char *e = getenv("MASH_AMD_BATCH_BYTES");
            b.max_bytes = e ? std::max<uint64_t>(1, strtoull(e, nullptr, 10)) : 64ull << 20;
        }
    }
    vector<ParsedFile> group;
    auto announce = [&](size_t i) {
        if (verbosity <= 0) return;
        if (files[i] == "-") g_progress.line("Sketching from stdin...\n");
        else g_progress.line("Sketching " + files[i] + "...\n");
    };
    for (size_t i = 0; i < files.size(); i++) {
        if (has_suffix(files[i], kSuffix)) {
            g_progress.flush();
            flush_batch(gpu, set, b);                      // keep input order
            load_msh_into(set, files[i], i == 0 && !enforce_parameters);
            if (pool) pool->skip(i);
            continue;
        }
        announce(i);
        if (pool && parseable(i)) {
            // (a worker opens the file; a failure comes back in input order as the file's error)
            if (pool->ready_bytes(i) < 0) g_progress.flush();          // about to wait: show where we are
            ParsedFile pf = pool->take(i, set.p.kmer);
            if (grouped) ensure_session(gpu, set, b);
            const uint64_t cap = grouped ? mg_sketch_stage_capacity(b.sess) : 0;
            if (!grouped || !pf.error.empty() || pf.bases.size() > cap) {
                queue_parsed_file(gpu, set, b, std::move(pf));
                continue;
            }
            // this file and those behind it that are parsed already, while they fit one staging window
            uint64_t bytes = pf.bases.size();
            group.push_back(std::move(pf));
            while (i + 1 < files.size() && group.size() < 4096 && parseable(i + 1)) {
                const long long nb = pool->ready_bytes(i + 1);
                if (nb < 0 || bytes + (uint64_t)nb > cap) break;
                ParsedFile nx = pool->take(i + 1, set.p.kmer);
                announce(++i);
                if (!nx.error.empty()) {                   // everything before it first, then its message (and exit)
                    queue_parsed_group(gpu, set, b, group, *pool);
                    queue_parsed_file(gpu, set, b, std::move(nx));
                }
                bytes += (uint64_t)nb;
                group.push_back(std::move(nx));
            }
            queue_parsed_group(gpu, set, b, group, *pool);
            continue;
        }
        g_progress.flush();
        if (files[i] != "-") {
            FILE *t = fopen(files[i].c_str(), "r");
            if (!t) { cerr << "ERROR: could not open " << files[i] << " for reading." << endl; exit(1); }
            fclose(t);
        }
        if (pool) pool->skip(i);
        if (reads_files) {
            flush_batch(gpu, set, b);                      // keep input order
            sketch_reads(gpu, set, {files[i]}, &reads_sess, &reads_rs);
        } else if (set.p.concatenated) {
            queue_parsed_file(gpu, set, b, parse_file_concatenated(files[i], set.p.kmer));
        } else {
            queue_file_by_sequence(gpu, set, b, files[i]);
        }
    }
    g_progress.flush();
    flush_batch(gpu, set, b);
    if (b.sess) mg_sketch_session_free(b.sess);
    if (reads_sess) mg_sketch_session_free(reads_sess);
    if (reads_rs) mg_reads_free(reads_rs);
}

string write_set(SketchSet &set, const string &path, bool consume = false)
{
    mshio::File f;
    f.header.kmer_size = (uint32_t)set.p.kmer;
    f.header.sketch_size = (uint32_t)set.p.sketch_size;
    f.header.seed = set.p.seed;
    f.header.concatenated = set.p.concatenated;
    f.header.noncanonical = set.p.noncanonical;
    f.header.preserve_case = set.p.preserve_case;
    f.header.has_alphabet = true;
    f.header.alphabet = set.p.alphabet;
    f.header.has_counts = set.p.counts;
    f.references.reserve(set.refs.size());
    for (Ref &r : set.refs) {
        mshio::Reference x;
        x.name = r.name; x.comment = r.comment; x.length = r.length;
        if (consume) { x.hashes = std::move(r.hashes); x.counts = std::move(r.counts); }     // (the caller is done with the set)
        else { x.hashes = r.hashes; x.counts = r.counts; }
        f.references.push_back(std::move(x));
    }
    return mshio::write_msh(path, f);
}

// warnKmerSize machinery (CommandSketch.cpp:110-131, sketchParameterSetup.cpp:107-125, Sketch.cpp:53-61)
struct KmerWarning {
    uint64_t length_max = 0;
    string name;
    double random_chance = 0;
    int k_min = 0;
    int count = 0;
};

KmerWarning scan_kmer_warning(const SketchSet &set)
{
    KmerWarning w;
    const double warning = set.p.warning;
    const double threshold = (warning * set.kmer_space()) / (1. - warning);
    for (const Ref &r : set.refs) {
        if ((double)r.length > threshold) {
            if (w.count == 0 || r.length > w.length_max) {
                w.length_max = r.length;
                w.name = r.name;
                w.random_chance = 1. / (set.kmer_space() / r.length + 1.);
                w.k_min = (int)std::ceil(std::log(r.length * (1 - warning) / warning) / std::log((double)set.p.alphabet_size));
            }
            w.count++;
        }
    }
    return w;
}

void warn_kmer_size(const SketchSet &set, const KmerWarning &w)
{
    cerr << "\nWARNING: For the k-mer size used (" << set.p.kmer << "), the random match probability (" << w.random_chance
         << ") is above the specified warning threshold (" << set.p.warning << ") for the sequence \"" << w.name
         << "\" of size " << w.length_max;
    if (w.count > 1) cerr << " (and " << (w.count - 1) << " others)";
    cerr << ". Distances to " << (w.count == 1 ? "this sequence" : "these sequences")
         << " may be underestimated as a result. To meet the threshold of " << set.p.warning << ", a k-mer size of at least "
         << w.k_min << " is required. See: -k, -w." << endl << endl;
}

// dense table upload
// the same table on EVERY device of the communicator (host -> GPU 0 -> RCCL broadcast)
mg_dtable *upload_all(Gpu &gpu, const SketchSet &set, uint64_t s, vector<uint64_t> *lengths_out = nullptr, bool by_rows = false)
{
    const uint64_t n = set.refs.size();
    vector<uint64_t> h(std::max<uint64_t>(n * s, 1), MG_HASH_PAD), len(std::max<uint64_t>(n, 1));
    vector<uint32_t> nh(std::max<uint64_t>(n, 1));
    for (uint64_t i = 0; i < n; i++) {
        const Ref &r = set.refs[i];
        const uint64_t k = std::min<uint64_t>(r.hashes.size(), s);
        nh[i] = (uint32_t)k;
        len[i] = r.length;
        std::copy(r.hashes.begin(), r.hashes.begin() + k, h.begin() + i * s);
    }
    mg_dtable *d = nullptr;
    // by_rows: the larger side of a rect job -- every GPU gets a block of its rows instead of a replica
    if ((by_rows ? mg_dtable_upload_rows(gpu.comm, h.data(), nh.data(), len.data(), n, s, &d)
                 : mg_dtable_upload(gpu.comm, h.data(), nh.data(), len.data(), n, s, &d)) != MG_OK) {
        cerr << "ERROR: " << mg_comm_last_error(gpu.comm) << endl;
        exit(1);
    }
    if (lengths_out) *lengths_out = len;
    return d;
}

// ------------------------------------------------------------------------------- commands
int cmd_sketch(int argc, const char **argv)
{
    Cmd c;
    c.name = "sketch";
    c.add("help", Opt::Boolean, "h");
    c.add("list", Opt::Boolean, "l");
    c.add("prefix", Opt::File, "o");
    c.add("id", Opt::File, "I");
    c.add("comment", Opt::File, "C");
    c.add("counts", Opt::Boolean, "M");
    c.use_sketch_options();
    if (c.parse(argc, argv)) return 1;
    if (c.args.empty() || c.o("help").active) {
        cout << "\nUsage:\n\n  mash sketch [options] <input> [<input>] ...\n\n"
                "Create a sketch file (.msh) from fasta/fastq inputs (gzipped or not) on the GPU.\n"
                "Options: -l -o <prefix> -I <id> -C <comment> -p <threads> -k <1-32> -s <size> -S <seed> -i -n -a -z <alphabet> -Z -w <p>\n"
                "Reads:   -r  -m <min copies> (0 keeps no hash, as in the reference: refused unless -g gives a length)  -c <target coverage>\n"
                "         -b <Bloom filter bytes, K/M/G/T>  -g <genome size>  -M (store multiplicities)\n\n";
        return 0;
    }
    Params p;
    p.counts = c.o("counts").active;
    if (sketch_parameter_setup(p, c)) return 1;
    vector<string> files;
    for (const string &a : c.args) { if (c.o("list").active) split_file(a, files); else files.push_back(a); }
    if ((c.o("id").active || c.o("comment").active) && files.size() > 1 && !p.reads)
        cerr << "WARNING: -I and -C will only apply to first sketch" << endl;
    StageClock clk;
    // MASH_AMD_EARLY_PARSE=1: the parse workers start BEFORE the device context exists, so that its creation
    // (0.07-0.2 s) is spent parsing (bounded: MASH_AMD_PARSE_AHEAD bytes, default 1 GiB).  Off by default:
    // measured (tools/sketch_e2e.py, 12 000 x 50 kbp at -p 16: 0.26 s with, 0.21 s without) -- once the
    // consumer's chain (staging copies, device tail, hand-out) is what bounds the run, parsing earlier buys
    // nothing and a gigabyte of parse buffers allocated up front slows context creation and the writer.
    // Not when a leading .msh may still change k (Sketch.cpp:140-160).
    std::unique_ptr<ParsePool> early;
    if (!p.reads && p.concatenated && p.threads > 1 && !files.empty() && !has_suffix(files[0], kSuffix) && getenv("MASH_AMD_EARLY_PARSE")) {
        early.reset(new ParsePool(files, (size_t)p.threads, [&files, &p](size_t i) { return parseable_by_pool(files, i, p); }));
        early->start(0, p.kmer);
    }
    Gpu gpu;
    clk.lap("device");
    SketchSet set;
    if (p.reads) { set.p = p; sketch_reads(gpu, set, files); }
    else init_from_files(gpu, set, files, p, 1, false, early.get());
    early.reset();
    clk.lap("ingest+sketch");
    if (clk.on) cerr << "timing: of which mg_sketch_host " << g_gpu_sketch_seconds << " s" << endl;
    if (c.o("id").active && !set.refs.empty()) set.refs[0].name = c.o("id").arg;
    if (c.o("comment").active && !set.refs.empty()) set.refs[0].comment = c.o("comment").arg;
    const KmerWarning w = scan_kmer_warning(set);
    string prefix = !c.o("prefix").arg.empty() ? c.o("prefix").arg : (c.args[0] == "-" ? string("stdin") : c.args[0]);
    if (!has_suffix(prefix, kSuffix)) prefix += kSuffix;
    cerr << "Writing to " << prefix << "..." << endl;
    const string e = write_set(set, prefix, true);
    clk.lap("write");
    if (!e.empty()) { cerr << "ERROR: " << e << endl; return 1; }
    if (w.count > 0 && !p.reads) warn_kmer_size(set, w);
    return 0;
}

// Bulk result lines: same bytes `cout << double` would print (default precision 6 == "%g" ==
// to_chars general/6), but buffered and without a flush per line -- at 10^8 lines the
// reference's `<< endl` formatting is what a run waits for once the kernels take milliseconds.
struct FastOut {
    string buf;
    bool sink;                               // false: a worker's piece, appended to the real one in order
    explicit FastOut(bool to_stdout = true) : sink(to_stdout) { if (sink) buf.reserve(1u << 22); }
    ~FastOut() { if (sink) flush(); }
    void flush()
    {
        cout.flush();
        if (!buf.empty()) { fwrite(buf.data(), 1, buf.size(), stdout); fflush(stdout); buf.clear(); }
    }
    void room() { if (sink && buf.size() > (1u << 22) - 4096) flush(); }
    FastOut &operator<<(const string &x) { buf += x; return *this; }
    FastOut &operator<<(const char *x) { buf += x; return *this; }
    FastOut &operator<<(char x) { buf += x; return *this; }
    FastOut &operator<<(double v)
    {
        char t[40];
        auto r = std::to_chars(t, t + sizeof t, v, std::chars_format::general, 6);
        buf.append(t, r.ptr);
        return *this;
    }
    FastOut &operator<<(uint32_t v)
    {
        char t[16];
        auto r = std::to_chars(t, t + sizeof t, v);
        buf.append(t, r.ptr);
        return *this;
    }
    void eol() { buf += '\n'; room(); }
};

// Text of a block of result rows, formatted on worker threads and written in row order (number
// formatting, not the GPU, bounds a full matrix: ~50 ns per value on one core).  Rows are cut
// into chunks of >= 2^17 pairs, a wave of chunks is formatted concurrently, the pieces are
// appended in order; fn(out, row, slot) must only touch slot-private state besides `out`.
unsigned emit_threads()
{
    static const unsigned nt = []() {
        if (const char *e = getenv("MASH_AMD_EMIT_THREADS")) return (unsigned)std::min(64, std::max(1, atoi(e)));
        return std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    }();
    return nt;
}

template <class CostFn, class RowFn>
void emit_rows(FastOut &out, uint64_t r0, uint64_t r1, CostFn cost, RowFn fn)
{
    const unsigned nt = emit_threads();
    vector<uint64_t> cut{r0};
    uint64_t acc = 0;
    for (uint64_t r = r0; r < r1; r++) {
        acc += cost(r);
        if (acc >= (1u << 17)) { cut.push_back(r + 1); acc = 0; }
    }
    if (cut.back() != r1) cut.push_back(r1);
    const size_t nchunks = cut.size() - 1;
    if (nt < 2 || nchunks < 2) {
        for (uint64_t r = r0; r < r1; r++) fn(out, r, 0u);
        return;
    }
    vector<FastOut> part;
    part.reserve(nt);
    for (unsigned t = 0; t < nt; t++) part.emplace_back(false);
    for (size_t c0 = 0; c0 < nchunks; c0 += nt) {
        const unsigned m = (unsigned)std::min<size_t>(nt, nchunks - c0);
        vector<std::thread> th;
        for (unsigned t = 0; t < m; t++)
            th.emplace_back([&, t]() {
                part[t].buf.clear();
                for (uint64_t r = cut[c0 + t]; r < cut[c0 + t + 1]; r++) fn(part[t], r, t);
            });
        for (auto &x : th) x.join();
        for (unsigned t = 0; t < m; t++) { out.buf += part[t].buf; out.room(); }
    }
}

void print_pair_line(FastOut &out, const Ref &ref, const Ref &qry, bool comment, const mg_pair &pr)
{
    out << ref.name;
    if (comment) out << ':' << ref.comment;
    out << '\t' << qry.name;
    if (comment) out << ':' << qry.comment;
    out << '\t' << pr.distance << '\t' << pr.p_value << '\t' << pr.numer << '/' << pr.denom;
    out.eol();
}

// Thresholded runs (-d < 1 or -v < 1): both filters of compareSketches, the distances and p-values
// of the survivors and their compaction run on the device (mg_compare_*_results_host), so only
// surviving pairs cross PCIe, as finished records; bit-identical to the host arithmetic
// (MASH_AMD_NO_FILTER=1 takes the full-matrix path instead, MASH_AMD_HOST_FINISH=1 the host tail).
bool edge_filter_wanted(double d_max, double p_max) { return (d_max < 1.0 || p_max < 1.0) && !getenv("MASH_AMD_NO_FILTER"); }
bool host_finish_wanted() { return getenv("MASH_AMD_HOST_FINISH") != nullptr; }

template <class Call>
bool fetch_results(Gpu &gpu, vector<mg_result> &res, Call call)
{
    uint64_t n = 0;
    if (res.size() < (1u << 16)) res.resize(1u << 16);
    int rc = call(res.data(), (uint64_t)res.size(), &n);
    if (rc == MG_ERR_NOMEM && n > res.size()) {
        res.resize(n);
        rc = call(res.data(), (uint64_t)res.size(), &n);
    }
    if (rc != MG_OK) {
        // the sharded calls record their error in the communicator (or a shard's context)
        const char *m = mg_comm_last_error(gpu.comm);
        cerr << "ERROR: " << ((m && *m) ? m : mg_last_error(gpu.ctx)) << endl;
        return false;
    }
    res.resize(n);
    return true;
}

template <class Call>
bool fetch_edges(Gpu &gpu, vector<mg_edge> &edges, Call call)
{
    uint64_t n = 0;
    if (edges.size() < (1u << 16)) edges.resize(1u << 16);
    int rc = call(edges.data(), (uint64_t)edges.size(), &n);
    if (rc == MG_ERR_NOMEM && n > edges.size()) {
        edges.resize(n);
        rc = call(edges.data(), (uint64_t)edges.size(), &n);
    }
    if (rc != MG_OK) { cerr << "ERROR: " << mg_last_error(gpu.ctx) << endl; return false; }
    edges.resize(n);
    return true;
}

bool finish_edge(const mg_edge &e, uint64_t len_ref, uint64_t len_qry, int k, double kspace, double p_max, mg_pair &pr)
{
    pr.numer = e.numer;
    pr.denom = e.denom;
    pr.distance = mg_distance(e.numer, e.denom, k);
    pr.p_value = mg_p_value(e.numer, len_ref, len_qry, kspace, e.denom);
    pr.pass = pr.p_value <= p_max;                           // CommandDistance.cpp:419-422
    return pr.pass;
}

int cmd_dist(int argc, const char **argv)
{
    Cmd c;
    c.name = "dist";
    c.add("help", Opt::Boolean, "h");
    c.add("list", Opt::Boolean, "l");
    c.add("table", Opt::Boolean, "t");
    c.add("pvalue", Opt::Number, "v", "1.0", 0., 1.);
    c.add("distance", Opt::Number, "d", "1.0", 0., 1.);
    c.add("comment", Opt::Boolean, "C");
    c.use_sketch_options();
    if (c.parse(argc, argv)) return 1;
    if (c.args.size() < 2 || c.o("help").active) {
        cout << "\nUsage:\n\n  mash dist [options] <reference> <query> [<query>] ...\n\n"
                "Output fields: [reference-ID, query-ID, distance, p-value, shared-hashes].\n"
                "Options: -l -t -v <max p> -d <max dist> -C and the sketch options of `mash sketch`.\n\n";
        return 0;
    }
    const bool table = c.o("table").active, comment = c.o("comment").active;
    const double p_max = c.o("pvalue").num, d_max = c.o("distance").num;
    Params p;
    if (sketch_parameter_setup(p, c)) return 1;
    const string &file_ref = c.args[0];
    const bool is_sketch = has_suffix(file_ref, kSuffix);
    if (is_sketch) {
        for (const char *o : {"kmer", "noncanonical", "protein", "alphabet"})
            if (c.o(o).active) {
                cerr << "ERROR: The option -" << c.o(o).id << " cannot be used when a sketch is provided; it is inherited from the sketch." << endl;
                return 1;
            }
    } else {
        cerr << "Sketching " << file_ref << " (provide sketch file made with \"mash sketch\" to skip)...";
    }
    Gpu gpu;
    SketchSet ref;
    init_from_files(gpu, ref, {file_ref}, p, is_sketch ? 1 : 0);
    KmerWarning w;
    if (is_sketch) {
        if (c.o("sketchSize").active && p.reads && p.sketch_size != ref.p.sketch_size) {        // CommandDistance.cpp:121-128
            cerr << "ERROR: The sketch size must match the reference when using a bloom filter (leave this option out to inherit from the reference sketch)." << endl;
            return 1;
        }
        p.sketch_size = ref.p.sketch_size;
        p.kmer = ref.p.kmer;
        p.noncanonical = ref.p.noncanonical;
        p.preserve_case = ref.p.preserve_case;
        p.seed = ref.p.seed;
        set_alphabet(p, ref.p.alphabet);
    } else {
        w = scan_kmer_warning(ref);
        cerr << "done.\n";
    }
    if (table) {
        cout << "#query";
        for (const Ref &r : ref.refs) cout << '\t' << r.name;
        cout << endl;
    }
    vector<string> qfiles;
    for (size_t i = 1; i < c.args.size(); i++) { if (c.o("list").active) split_file(c.args[i], qfiles); else qfiles.push_back(c.args[i]); }
    SketchSet qry;
    init_from_files(gpu, qry, qfiles, p, 0, true);
    const uint64_t nref = ref.refs.size(), nq = qry.refs.size();
    if (nref == 0 || nq == 0) return 0;
    vector<uint64_t> len_ref, len_qry;
    // several GPUs: the larger side is cut into row blocks; references larger than the queries
    // are then not even replicated (the host-tail test paths address device 0's table as a whole)
    const bool ref_by_rows = mg_comm_size(gpu.comm) > 1 && nref > nq && !host_finish_wanted() && !getenv("MASH_AMD_NO_FILTER");
    mg_dtable *dr = upload_all(gpu, ref, ref.p.sketch_size, &len_ref, ref_by_rows);
    mg_dtable *dq = upload_all(gpu, qry, qry.p.sketch_size, &len_qry);
    mg_table *tr = mg_dtable_local(dr, 0), *tq = mg_dtable_local(dq, 0);
    const double kspace = ref.kmer_space();
    const uint64_t qblock = std::max<uint64_t>(1, (1ull << 24) / nref);
    vector<mg_counts> counts;
    vector<mg_pair> pairs;
    FastOut out;
    if (!table && edge_filter_wanted(d_max, p_max)) {
        vector<mg_edge> edges;
        vector<mg_result> res;
        const uint64_t fblock = std::max<uint64_t>(1, (1ull << 30) / nref);
        for (uint64_t q0 = 0; q0 < nq; q0 += fblock) {
            const uint64_t q1 = std::min(nq, q0 + fblock);
            if (!host_finish_wanted()) {
                if (!fetch_results(gpu, res, [&](mg_result *o, uint64_t cap, uint64_t *n) {
                        return mg_compare_rect_results_sharded_host(gpu.comm, dr, dq, q0, q1, ref.p.kmer, kspace, d_max, p_max, o, cap, n); }))
                    return 1;
                emit_rows(out, 0, res.size(), [](uint64_t) { return 1; }, [&](FastOut &o, uint64_t x, unsigned) {
                    const mg_result &e = res[x];
                    mg_pair pr;
                    pr.numer = e.numer; pr.denom = e.denom; pr.distance = e.distance; pr.p_value = e.p_value; pr.pass = 1;
                    print_pair_line(o, ref.refs[e.col], qry.refs[e.row], comment, pr);
                });
                continue;
            }
            if (!fetch_edges(gpu, edges, [&](mg_edge *o, uint64_t cap, uint64_t *n) {
                    return mg_compare_rect_filter_host(gpu.ctx, tr, tq, q0, q1, ref.p.kmer, d_max, o, cap, n); }))
                return 1;
            emit_rows(out, 0, edges.size(), [](uint64_t) { return 1; }, [&](FastOut &o, uint64_t x, unsigned) {
                const mg_edge &e = edges[x];
                mg_pair pr;
                if (finish_edge(e, len_ref[e.col], len_qry[e.row], ref.p.kmer, kspace, p_max, pr))
                    print_pair_line(o, ref.refs[e.col], qry.refs[e.row], comment, pr);
            });
        }
        mg_dtable_free(dr);
        mg_dtable_free(dq);
        if (w.count > 0 && !p.reads) warn_kmer_size(ref, w);
        return 0;
    }
    // -t without a p-value filter prints distances only: they come from one table per run
    // (a distance depends on numer / denom alone) and no p-value is evaluated
    const uint64_t s_tab = std::min(ref.p.sketch_size, qry.p.sketch_size);
    vector<double> dist_lut;
    vector<string> dist_txt;
    if (table && p_max >= 1.0 && s_tab <= (1u << 20) && !getenv("MASH_AMD_FULL_FINISH")) {
        dist_lut.resize(s_tab + 1);
        for (uint64_t x = 0; x <= s_tab; x++) dist_lut[x] = mg_distance((uint32_t)x, (uint32_t)s_tab, ref.p.kmer);
        dist_txt.resize(s_tab + 1);                                // the cell of the table, formatted once
        for (uint64_t x = 0; x <= s_tab; x++) {
            FastOut t(false);
            t << '\t';
            if (!(dist_lut[x] > d_max)) t << dist_lut[x];          // CommandDistance.cpp:409-412
            dist_txt[x] = t.buf;
        }
    }
    for (uint64_t q0 = 0; q0 < nq; q0 += qblock) {
        const uint64_t q1 = std::min(nq, q0 + qblock);
        if (dist_lut.empty() && !host_finish_wanted()) {
            // compare + distance + p-value + filters on the device, finished records back
            pairs.resize((q1 - q0) * nref);
            if (mg_compare_rect_pairs_sharded_host(gpu.comm, dr, dq, q0, q1, ref.p.kmer, kspace, d_max, p_max, pairs.data()) != MG_OK) {
                cerr << "ERROR: " << mg_comm_last_error(gpu.comm) << endl;
                return 1;
            }
        } else {
            counts.resize((q1 - q0) * nref);
            if (mg_compare_rect_sharded_host(gpu.comm, dr, dq, q0, q1, counts.data()) != MG_OK) { cerr << "ERROR: " << mg_comm_last_error(gpu.comm) << endl; return 1; }
        }
        if (!dist_lut.empty()) {
            emit_rows(out, q0, q1, [&](uint64_t) { return nref; }, [&](FastOut &o, uint64_t q, unsigned) {
                o << qry.refs[q].name;                            // writeOutput, CommandDistance.cpp:247-304
                for (uint64_t r = 0; r < nref; r++) {
                    const mg_counts &c = counts[(q - q0) * nref + r];
                    if (c.denom == s_tab) o << dist_txt[c.numer];
                    else {
                        const double d = mg_distance(c.numer, c.denom, ref.p.kmer);
                        o << '\t';
                        if (!(d > d_max)) o << d;                  // CommandDistance.cpp:409-412
                    }
                    o.room();
                }
                o.eol();
            });
            continue;
        }
        if (host_finish_wanted()) {
            pairs.resize(counts.size());
            mg_finish_rect_host(counts.data(), len_ref.data(), nref, len_qry.data() + q0, q1 - q0, ref.p.kmer, kspace, d_max, p_max, pairs.data());
        }
        emit_rows(out, q0, q1, [&](uint64_t) { return nref; }, [&](FastOut &o, uint64_t q, unsigned) {
            if (table) o << qry.refs[q].name;                // writeOutput, CommandDistance.cpp:247-304
            for (uint64_t r = 0; r < nref; r++) {
                const mg_pair &pr = pairs[(q - q0) * nref + r];
                if (table) { o << '\t'; if (pr.pass) o << pr.distance; o.room(); }
                else if (pr.pass) print_pair_line(o, ref.refs[r], qry.refs[q], comment, pr);
            }
            if (table) o.eol();
        });
    }
    mg_dtable_free(dr);
    mg_dtable_free(dq);
    if (w.count > 0 && !p.reads) warn_kmer_size(ref, w);
    return 0;
}

int cmd_triangle(int argc, const char **argv)
{
    Cmd c;
    c.name = "triangle";
    c.add("help", Opt::Boolean, "h");
    c.add("list", Opt::Boolean, "l");
    c.add("comment", Opt::Boolean, "C");
    c.add("edge", Opt::Boolean, "E");
    c.add("pvalue", Opt::Number, "v", "1.0", 0., 1.);
    c.add("distance", Opt::Number, "d", "1.0", 0., 1.);
    c.use_sketch_options();
    if (c.parse(argc, argv)) return 1;
    if (c.args.empty() || c.o("help").active) {
        cout << "\nUsage:\n\n  mash triangle [options] <seq1> [<seq2>] ...\n\n"
                "Lower-triangular distance matrix in relaxed Phylip format (or -E edge list).\n"
                "Options: -l -C -E -v <max p> -d <max dist> and the sketch options of `mash sketch`.\n\n";
        return 0;
    }
    const bool comment = c.o("comment").active;
    bool edge = c.o("edge").active;
    const double p_max = c.o("pvalue").num, d_max = c.o("distance").num;
    if (c.o("pvalue").active || c.o("distance").active) edge = true;
    Params p;
    if (sketch_parameter_setup(p, c)) return 1;
    if (c.args.size() == 1 && !c.o("list").active) p.concatenated = false;   // CommandTriangle.cpp:74-77
    vector<string> files;
    for (const string &a : c.args) { if (c.o("list").active) split_file(a, files); else files.push_back(a); }
    Gpu gpu;
    SketchSet set;
    init_from_files(gpu, set, files, p, 1);
    const KmerWarning w = scan_kmer_warning(set);
    const uint64_t n = set.refs.size();
    if (n == 0) return 0;
    auto label = [&](const Ref &r) -> const string & { return comment ? r.comment : r.name; };
    if (!edge) {
        cout << '\t' << n << endl;
        cout << label(set.refs[0]) << endl;
    }
    vector<uint64_t> lengths;
    mg_dtable *dt = upload_all(gpu, set, set.p.sketch_size, &lengths);
    mg_table *t = mg_dtable_local(dt, 0);
    const double kspace = set.kmer_space();
    double p_peak = 0;
    vector<mg_counts> counts;
    vector<mg_pair> pairs;
    uint64_t r0 = 1;
    StageClock clk;
    FastOut out;
    if (edge && edge_filter_wanted(d_max, p_max)) {
        vector<mg_edge> edges;
        vector<mg_result> res;
        while (r0 < n) {
            uint64_t r1 = r0, npairs = 0;
            while (r1 < n && (npairs == 0 || npairs + r1 <= (1ull << 31))) { npairs += r1; r1++; }
            if (!host_finish_wanted()) {
                if (!fetch_results(gpu, res, [&](mg_result *o, uint64_t cap, uint64_t *cnt) {
                        return mg_compare_tri_results_sharded_host(gpu.comm, dt, r0, r1, set.p.kmer, kspace, d_max, p_max, o, cap, cnt); }))
                    return 1;
                clk.lap("compare+finish+filter+copy");
                emit_rows(out, 0, res.size(), [](uint64_t) { return 1; }, [&](FastOut &o, uint64_t x, unsigned) {
                    const mg_result &e = res[x];
                    o << label(set.refs[e.row]) << '\t' << label(set.refs[e.col]) << '\t' << e.distance << '\t'
                      << e.p_value << '\t' << e.numer << '/' << e.denom;
                    o.eol();
                });
                clk.lap("format+write");
                r0 = r1;
                continue;
            }
            if (!fetch_edges(gpu, edges, [&](mg_edge *o, uint64_t cap, uint64_t *cnt) {
                    return mg_compare_tri_filter_host(gpu.ctx, t, r0, r1, set.p.kmer, d_max, o, cap, cnt); }))
                return 1;
            clk.lap("compare+filter+copy");
            emit_rows(out, 0, edges.size(), [](uint64_t) { return 1; }, [&](FastOut &o, uint64_t x, unsigned) {
                const mg_edge &e = edges[x];
                mg_pair pr;
                if (finish_edge(e, lengths[e.row], lengths[e.col], set.p.kmer, kspace, p_max, pr)) {
                    o << label(set.refs[e.row]) << '\t' << label(set.refs[e.col]) << '\t' << pr.distance << '\t'
                      << pr.p_value << '\t' << pr.numer << '/' << pr.denom;
                    o.eol();
                }
            });
            clk.lap("finish+format+write");
            r0 = r1;
        }
    }
    // The Phylip matrix prints distances only, and of the p-values just the largest (to stderr).
    // A distance depends on (numer, denom) alone -> one table per run; and a p-value can only raise
    // the peak if even the pair of the two longest genomes sharing that many hashes would: the
    // p-value grows with both lengths (larger r in CommandDistance.cpp:437-441, the binomial tail
    // grows with r), so pairs whose bound stays below the running peak skip the incomplete-beta
    // evaluation.  The bound test has a 1e-9 relative margin for the rounding of that evaluation.
    const uint64_t s_tab = set.p.sketch_size;
    vector<double> dist_lut, p_bound;
    uint64_t bound_l1 = 0, bound_l2 = 0;
    // p-value of the two longest genomes sharing x hashes, with a 1e-9 margin; memoised (workers may both
    // evaluate an entry: they store the same bits)
    auto bound_of = [&](uint32_t x) -> double {
        uint64_t bits = __atomic_load_n(reinterpret_cast<const uint64_t *>(&p_bound[x]), __ATOMIC_RELAXED);
        double v;
        memcpy(&v, &bits, 8);
        if (v != v) {                                              // not evaluated yet
            v = mg_p_value(x, bound_l1, bound_l2, kspace, s_tab) * (1.0 + 1e-9);
            memcpy(&bits, &v, 8);
            __atomic_store_n(reinterpret_cast<uint64_t *>(&p_bound[x]), bits, __ATOMIC_RELAXED);
        }
        return v;
    };
    vector<string> dist_txt;                                       // "\t" + the text of dist_lut[x]: formatted once, not per pair
    if (!edge && s_tab <= (1u << 20) && !getenv("MASH_AMD_FULL_FINISH")) {          // (env: the plain path, for tests)
        uint64_t l1 = 0, l2 = 0;                                   // the two largest lengths
        for (uint64_t v : lengths) { if (v > l1) { l2 = l1; l1 = v; } else if (v > l2) l2 = v; }
        dist_lut.resize(s_tab + 1);
        p_bound.resize(s_tab + 1);
        // (the bounds are evaluated on first use: a run meets a few hundred distinct numerators, and an
        //  exact tail costs O(x) -- all s of them up front were minutes at s = 10^5 and a large r)
        for (uint64_t x = 0; x <= s_tab; x++) {
            dist_lut[x] = mg_distance((uint32_t)x, (uint32_t)s_tab, set.p.kmer);
            p_bound[x] = std::nan("");
        }
        bound_l1 = l1;
        bound_l2 = l2;
        dist_txt.resize(s_tab + 1);
        for (uint64_t x = 0; x <= s_tab; x++) {
            FastOut t(false);
            t << '\t' << dist_lut[x];
            dist_txt[x] = t.buf;
        }
    }
    const bool lean = !dist_lut.empty();
    // The Phylip matrix from the SPARSE result (mg_compare_tri_sparse_host): a pair that shares no hash is {0, min(s, |A| + |B|)}
    // -- distance 1, p-value 1 -- and in a collection nearly every pair is such a pair: what crosses PCIe and is looked at are
    // the exceptions, a row's text is runs of "\t1" between them (CommandTriangle.cpp:159-198 writes the same cells).  One
    // device; a block whose exceptions outnumber an eighth of its pairs (one species: every pair shares hashes) sends the
    // rest of the run down the dense path below, which has the engines for that.  (MASH_AMD_DENSE_MATRIX=1: dense from the start.)
    bool sparse_rows = lean && mg_comm_size(gpu.comm) == 1 && !getenv("MASH_AMD_DENSE_MATRIX");
    vector<mg_edge> exc;
    vector<uint64_t> exc_at;
    string ones;
    if (sparse_rows) for (int k = 0; k < 4096; k++) ones += "\t1";
    while (r0 < n && sparse_rows) {
        uint64_t r1 = r0, npairs = 0;
        while (r1 < n && (npairs == 0 || npairs + r1 <= (1ull << 26))) { npairs += r1; r1++; }
        clk.lap("setup");
        const uint64_t cap = npairs / 8 + 4096;
        if (exc.size() < cap) exc.resize(cap);
        uint64_t cnt = 0;
        const int rc = mg_compare_tri_sparse_host(gpu.ctx, t, r0, r1, exc.data(), cap, &cnt);
        if (rc == MG_ERR_NOMEM) { sparse_rows = false; break; }
        if (rc != MG_OK) { cerr << "ERROR: " << mg_last_error(gpu.ctx) << endl; return 1; }
        clk.lap("compare+sparse result+copy");
        exc_at.assign(r1 - r0 + 1, cnt);                      // exceptions of row i: exc[exc_at[i - r0] .. exc_at[i - r0 + 1])
        for (uint64_t x = cnt; x-- > 0;) exc_at[exc[x].row - r0] = x;
        for (uint64_t i = r1 - r0; i-- > 0;) if (exc_at[i] > exc_at[i + 1]) exc_at[i] = exc_at[i + 1];
        vector<double> peak(emit_threads(), p_peak);
        emit_rows(out, r0, r1, [](uint64_t i) { return i / 16 + 64; }, [&](FastOut &o, uint64_t i, unsigned slot) {
            double pk = peak[slot];
            o << label(set.refs[i]);
            const bool empty_row = set.refs[i].hashes.empty();
            auto defaults = [&](uint64_t c0, uint64_t c1) {       // columns [c0, c1) share nothing with row i
                if (c0 >= c1) return;
                if (pk < 1.0) pk = 1.0;                          // pValue(0, ...) = 1 (CommandDistance.cpp:431-434)
                if (!empty_row) {
                    for (uint64_t m = c1 - c0; m;) { const uint64_t k = std::min<uint64_t>(m, 4096); o.buf.append(ones, 0, 2 * k); m -= k; }
                } else {
                    // (two empty sketches: 0 / 0 union elements, common == denom: distance 0, CommandDistance.cpp:390-393)
                    for (uint64_t j = c0; j < c1; j++) o << (set.refs[j].hashes.empty() ? "\t0" : "\t1");
                }
                o.room();
            };
            uint64_t c = 0;
            for (uint64_t x = exc_at[i - r0]; x < exc_at[i - r0 + 1]; x++) {
                const mg_edge &e = exc[x];
                defaults(c, e.col);
                const bool full = e.denom == s_tab;
                if (full) o << dist_txt[e.numer];
                else o << '\t' << mg_distance(e.numer, e.denom, set.p.kmer);
                if (!full || bound_of(e.numer) >= pk) {
                    const double pv = mg_p_value(e.numer, lengths[i], lengths[e.col], kspace, e.denom);
                    if (pv > pk) pk = pv;
                }
                c = (uint64_t)e.col + 1;
            }
            defaults(c, i);
            peak[slot] = pk;
            o.eol();
        });
        for (double v : peak) if (v > p_peak) p_peak = v;
        clk.lap("format+write");
        r0 = r1;
    }
    while (r0 < n) {
        uint64_t r1 = r0, npairs = 0;
        while (r1 < n && (npairs == 0 || npairs + r1 <= (1ull << 24))) { npairs += r1; r1++; }
        const bool dev_finish = !lean && !host_finish_wanted();
        if (!dev_finish) counts.resize(npairs);
        if (!lean) pairs.resize(npairs);
        clk.lap("setup");
        if (dev_finish) {
            if (mg_compare_tri_pairs_sharded_host(gpu.comm, dt, r0, r1, set.p.kmer, kspace, d_max, p_max, pairs.data()) != MG_OK) {
                cerr << "ERROR: " << mg_comm_last_error(gpu.comm) << endl;
                return 1;
            }
            clk.lap("compare+finish+copy");
        } else {
            if (mg_compare_tri_sharded_host(gpu.comm, dt, r0, r1, counts.data()) != MG_OK) { cerr << "ERROR: " << mg_comm_last_error(gpu.comm) << endl; return 1; }
            clk.lap("compare+copy");
            if (!lean) mg_finish_tri_host(counts.data(), lengths.data(), r0, r1, set.p.kmer, kspace, d_max, p_max, pairs.data());
            clk.lap("finish");
        }
        const uint64_t base = r0 * (r0 - 1) / 2;
        vector<double> peak(emit_threads(), 0.0);
        if (lean) {
            for (double &v : peak) v = p_peak;                     // what earlier blocks established
            emit_rows(out, r0, r1, [](uint64_t i) { return i; }, [&](FastOut &o, uint64_t i, unsigned slot) {
                uint64_t idx = i * (i - 1) / 2 - base;             // writeOutput, CommandTriangle.cpp:159-198
                double pk = peak[slot];
                o << label(set.refs[i]);
                for (uint64_t j = 0; j < i; j++, idx++) {
                    const mg_counts &c = counts[idx];
                    const bool full = c.denom == s_tab;
                    if (full) o << dist_txt[c.numer];
                    else o << '\t' << mg_distance(c.numer, c.denom, set.p.kmer);
                    o.room();
                    if (!full || bound_of(c.numer) >= pk) {
                        const double pv = mg_p_value(c.numer, lengths[i], lengths[j], kspace, c.denom);
                        if (pv > pk) pk = pv;
                    }
                }
                peak[slot] = pk;
                o.eol();
            });
            for (double v : peak) if (v > p_peak) p_peak = v;
            clk.lap("format+write");
            r0 = r1;
            continue;
        }
        emit_rows(out, r0, r1, [](uint64_t i) { return i; }, [&](FastOut &o, uint64_t i, unsigned slot) {
            const Ref &ref = set.refs[i];                     // writeOutput, CommandTriangle.cpp:159-198
            uint64_t idx = i * (i - 1) / 2 - base;
            double pk = peak[slot];
            if (!edge) o << label(ref);
            for (uint64_t j = 0; j < i; j++, idx++) {
                const mg_pair &pr = pairs[idx];
                if (edge) {
                    if (pr.pass) {
                        o << label(ref) << '\t' << label(set.refs[j]) << '\t' << pr.distance << '\t' << pr.p_value << '\t'
                          << pr.numer << '/' << pr.denom;
                        o.eol();
                    }
                } else {
                    o << '\t' << pr.distance;
                    o.room();
                }
                if (pr.p_value > pk) pk = pr.p_value;
            }
            peak[slot] = pk;
            if (!edge) o.eol();
        });
        for (double v : peak) if (v > p_peak) p_peak = v;
        clk.lap("format+write");
        r0 = r1;
    }
    mg_dtable_free(dt);
    if (!edge) cerr << "Max p-value: " << p_peak << endl;
    if (w.count > 0 && !p.reads) warn_kmer_size(set, w);
    return 0;
}

void print_columns(const vector<vector<string>> &cols, int indent, int spacing, const char *missing)
{
    vector<size_t> width(cols.size(), 0);
    for (size_t i = 0; i < cols.size(); i++)
        for (const string &s : cols[i]) width[i] = std::max(width[i], std::max<size_t>(s.size(), 1));
    for (size_t r = 0; r < cols[0].size(); r++) {
        size_t offset = 0, target = indent;
        for (size_t j = 0; j < cols.size(); j++) {
            for (size_t k = offset; k < target; k++) cout << ' ';
            const string text = cols[j][r].empty() ? string(missing) : cols[j][r];
            cout << text;
            offset = target + text.size();
            target += width[j] + spacing;
        }
        cout << endl;
    }
}

int cmd_info(int argc, const char **argv)
{
    Cmd c;
    c.name = "info";
    c.add("help", Opt::Boolean, "h");
    c.add("header", Opt::Boolean, "H");
    c.add("tabular", Opt::Boolean, "t");
    c.add("counts", Opt::Boolean, "c");
    c.add("dump", Opt::Boolean, "d");
    if (c.parse(argc, argv)) return 1;
    if (c.args.size() != 1 || c.o("help").active) {
        cout << "\nUsage:\n\n  mash info [options] <sketch>\n\nOptions: -H (header only) -t (tabular) -c (count histograms) -d (JSON dump)\n\n";
        return 0;
    }
    const bool header = c.o("header").active, tabular = c.o("tabular").active, counts = c.o("counts").active, dump = c.o("dump").active;
    auto incompatible = [](const char *a, const char *b) { cerr << "ERROR: The options " << a << " and " << b << " are incompatible." << endl; return 1; };
    if (header && tabular) return incompatible("-H", "-t");
    if (header && counts) return incompatible("-H", "-c");
    if (tabular && counts) return incompatible("-t", "-c");
    if (dump && tabular) return incompatible("-d", "-t");
    if (dump && header) return incompatible("-d", "-H");
    if (dump && counts) return incompatible("-d", "-c");
    const string &file = c.args[0];
    if (!has_suffix(file, kSuffix)) { cerr << "ERROR: The file \"" << file << "\" does not look like a sketch." << endl; return 1; }
    mshio::File f;
    const string e = mshio::read_msh(file, f, header);
    if (!e.empty()) { cerr << "ERROR: " << e << endl; return 1; }
    Params p;
    params_from_header(p, f.header);
    const uint64_t nref = header ? f.header.reference_count : f.references.size();
    if (counts) {
        if (f.references.empty()) { cerr << "ERROR: Sketch file contains no sketches" << endl; return 1; }
        if (!f.header.has_counts) { cerr << "ERROR: Sketch file does not have hash counts. Re-sketch with -M to use this feature." << endl; return 1; }
        cout << "#Sketch\tBin\tFrequency" << endl;
        for (const auto &r : f.references) {
            std::map<uint32_t, uint64_t> hist;
            for (uint32_t v : r.counts) hist[v]++;
            for (const auto &kv : hist) cout << r.name << '\t' << kv.first << '\t' << kv.second << endl;
        }
        return 0;
    }
    if (dump) {                                              // writeJson, CommandInfo.cpp:222-299 (byte for byte)
        cout << "{" << endl;
        cout << "	\"kmer\" : " << p.kmer << ',' << endl;
        cout << "	\"alphabet\" : \"" << p.alphabet << "\"," << endl;
        cout << "	\"preserveCase\" : " << (p.preserve_case ? "true" : "false") << ',' << endl;
        cout << "	\"canonical\" : " << (p.noncanonical ? "false" : "true") << ',' << endl;
        cout << "	\"sketchSize\" : " << p.sketch_size << ',' << endl;
        cout << "	\"hashType\" : \"MurmurHash3_x64_128\"," << endl;
        cout << "	\"hashBits\" : " << (p.use64 ? 64 : 32) << ',' << endl;
        cout << "	\"hashSeed\" : " << p.seed << ',' << endl;
        cout << " 	\"sketches\" :" << endl;
        cout << "	[" << endl;
        for (size_t i = 0; i < f.references.size(); i++) {
            const auto &r = f.references[i];
            cout << "		{" << endl;
            cout << "			\"name\" : \"" << r.name << "\"," << endl;
            cout << "			\"length\" : " << r.length << ',' << endl;
            cout << "			\"comment\" : \"" << r.comment << "\"," << endl;
            cout << "			\"hashes\" :" << endl;
            cout << "			[" << endl;
            for (size_t j = 0; j < r.hashes.size(); j++) {
                cout << "				" << r.hashes[j];
                if (j + 1 < r.hashes.size()) cout << ',';
                cout << endl;
            }
            cout << "			]" << endl;
            if (r.counts_sorted) {
                cout << "			\"counts\" :" << endl;
                cout << "			[" << endl;
                for (size_t j = 0; j < r.counts.size(); j++) {
                    cout << "				" << r.counts[j];
                    if (j + 1 < r.hashes.size()) cout << ',';
                    cout << endl;
                }
                cout << "			]" << endl;
            }
            cout << (i + 1 < f.references.size() ? "		}," : "		}") << endl;
        }
        cout << "	]" << endl;
        cout << "}" << endl;
        return 0;
    }
    if (tabular) {
        cout << "#Hashes\tLength\tID\tComment" << endl;
    } else {
        cout << "Header:" << endl;
        cout << "  Hash function (seed):          MurmurHash3_x64_128 (" << p.seed << ")" << endl;
        cout << "  K-mer size:                    " << p.kmer << " (" << (p.use64 ? "64" : "32") << "-bit hashes)" << endl;
        cout << "  Alphabet:                      " << p.alphabet << (p.noncanonical ? "" : " (canonical)") << (p.preserve_case ? " (case-sensitive)" : "") << endl;
        cout << "  Target min-hashes per sketch:  " << p.sketch_size << endl;
        cout << "  Sketches:                      " << nref << endl;
    }
    if (!header) {
        vector<vector<string>> cols(4);
        if (!tabular) {
            cout << endl << "Sketches:" << endl;
            cols[0].push_back("[Hashes]"); cols[1].push_back("[Length]"); cols[2].push_back("[ID]"); cols[3].push_back("[Comment]");
        }
        for (const auto &r : f.references) {
            if (tabular) cout << r.hashes.size() << '\t' << r.length << '\t' << r.name << '\t' << r.comment << endl;
            else {
                cols[0].push_back(std::to_string(r.hashes.size()));
                cols[1].push_back(std::to_string(r.length));
                cols[2].push_back(r.name);
                cols[3].push_back(r.comment);
            }
        }
        if (!tabular) print_columns(cols, 2, 2, "-");
    }
    return 0;
}

int cmd_paste(int argc, const char **argv)
{
    Cmd c;
    c.name = "paste";
    c.add("help", Opt::Boolean, "h");
    c.add("list", Opt::Boolean, "l");
    if (c.parse(argc, argv)) return 1;
    if (c.args.size() < 2 || c.o("help").active) {
        cout << "\nUsage:\n\n  mash paste [options] <out_prefix> <sketch> [<sketch>] ...\n\nOptions: -l (inputs are lists of file names)\n\n";
        return 0;
    }
    vector<string> files;
    for (size_t i = 1; i < c.args.size(); i++) { if (c.o("list").active) split_file(c.args[i], files); else files.push_back(c.args[i]); }
    for (const string &f : files)
        if (!has_suffix(f, kSuffix)) { cerr << "ERROR: The file \"" << f << "\" does not look like a sketch." << endl; return 1; }
    SketchSet set;
    for (size_t i = 0; i < files.size(); i++) load_msh_into(set, files[i], i == 0);
    string out = c.args[0];
    if (!has_suffix(out, kSuffix)) out += kSuffix;
    if (access(out.c_str(), F_OK) != -1) { cerr << "ERROR: \"" << out << "\" exists; remove to write." << endl; exit(1); }
    cerr << "Writing " << out << "..." << endl;
    const string e = write_set(set, out);
    if (!e.empty()) { cerr << "ERROR: " << e << endl; return 1; }
    return 0;
}

// mash screen (CommandScreen.cpp:54-461), nucleotide query sketches
int cmd_screen(int argc, const char **argv)
{
    Cmd c;
    c.name = "screen";
    c.add("help", Opt::Boolean, "h");
    c.add("threads", Opt::Integer, "p", "1");
    c.add("winning!", Opt::Boolean, "w");
    c.add("identity", Opt::Number, "i", "0", -1., 1.);
    c.add("pvalue", Opt::Number, "v", "1.0", 0., 1.);
    if (c.parse(argc, argv)) return 1;
    if (c.args.size() < 2 || c.o("help").active) {
        cout << "\nUsage:\n\n  mash screen [options] <queries>.msh <mixture> [<mixture>] ...\n\n"
                "Output fields: [identity, shared-hashes, median-multiplicity, p-value, query-ID, query-comment].\n"
                "Options: -w (winner-takes-all) -i <min identity> -v <max p-value>\n\n";
        return 0;
    }
    if (!has_suffix(c.args[0], kSuffix)) { cerr << "ERROR: " << c.args[0] << " does not look like a sketch (.msh)" << endl; exit(1); }
    const double p_max = c.o("pvalue").num, identity_min = c.o("identity").num;
    SketchSet set;
    load_msh_into(set, c.args[0], true);
    const bool trans = set.p.alphabet == normalise_alphabet(kAlphabetProtein, false);   // CommandScreen.cpp:120
    Gpu gpu;
    cerr << "Loading " << c.args[0] << "..." << endl;
    const uint64_t n = set.refs.size(), s = set.p.sketch_size;
    // the query table on every GPU; the mixture is sharded by batch over them (mg_dscreen)
    mg_dtable *t = upload_all(gpu, set, s);
    mg_params mp;
    mg_params_init(&mp, set.p.kmer, s, set.p.seed, set.p.alphabet.c_str(), set.p.noncanonical, set.p.preserve_case);
    mg_dscreen *sc = nullptr;
    if (mg_dscreen_create(gpu.comm, &mp, t, trans ? 1 : 0, &sc) != MG_OK) {
        cerr << "ERROR: " << mg_comm_last_error(gpu.comm) << endl;
        return 1;
    }
    const int nq = (int)c.args.size() - 1;
    vector<fastx::Reader *> readers;
    for (int f = 1; f <= nq; f++) {
        if (c.args[f] == "-" && f > 1) { cerr << "ERROR: '-' for stdin must be first query" << endl; exit(1); }
        fastx::Reader *r = new fastx::Reader;
        if (!r->open(c.args[f])) { cerr << "ERROR: could not open " << c.args[f] << endl; exit(1); }
        readers.push_back(r);
    }
    // mixture records, round robin over the inputs (CommandScreen.cpp:197-270); records < k are dropped
    vector<uint8_t> batch;
    batch.reserve(256u << 20);
    size_t screen_batch_bytes = 255u << 20;
    if (const char *e = getenv("MASH_AMD_SCREEN_BATCH")) screen_batch_bytes = std::max<size_t>(64, strtoull(e, nullptr, 10));   // test knob
    auto flush = [&]() {
        if (batch.empty()) return;
        if (mg_dscreen_add_host(sc, batch.data(), batch.size()) != MG_OK) { cerr << "ERROR: " << mg_comm_last_error(gpu.comm) << endl; exit(1); }
        batch.clear();
    };
    fastx::Record rec;
    size_t it = 0;
    long l = -1;
    uint64_t count = 0;
    while (!readers.empty()) {
        l = readers[it]->next(rec);
        if (l < -1) break;
        if (l == -1) {
            delete readers[it];
            readers.erase(readers.begin() + it);
            if (it == readers.size()) it = 0;
            continue;
        }
        count++;
        if (l >= set.p.kmer) {
            batch.insert(batch.end(), rec.seq.begin(), rec.seq.end());
            batch.push_back((uint8_t)MG_RECORD_SEP);
            if (batch.size() > screen_batch_bytes) flush();
        }
        it++;
        if (it == readers.size()) it = 0;
    }
    for (auto *r : readers) delete r;
    if (l != -1) { cerr << "\nERROR: reading inputs" << endl; exit(1); }
    flush();
    // what the mixture touched: one hit per (sketch, hash) that was observed -- not the n x s matrix of
    // counters, of which a mixture leaves all but a fraction of a per cent at zero
    vector<uint64_t> mix(s);
    uint32_t mix_n = 0;
    uint64_t distinct = 0, nhits = 0;
    if (mg_dscreen_finish_sparse_host(sc, nullptr, 0, &nhits, mix.data(), &mix_n, &distinct) != MG_OK) { cerr << "ERROR: " << mg_comm_last_error(gpu.comm) << endl; return 1; }
    vector<mg_screen_hit> hits(nhits);
    if (nhits && mg_dscreen_finish_sparse_host(sc, hits.data(), nhits, &nhits, nullptr, nullptr, nullptr) != MG_OK) { cerr << "ERROR: " << mg_comm_last_error(gpu.comm) << endl; return 1; }
    mg_dscreen_free(sc);
    mg_dtable_free(t);
    cerr << "   " << distinct << " distinct hashes." << endl;
    cerr << (trans ? "Translating from " : "Streaming from ");
    if (nq == 1) cerr << c.args[1]; else cerr << nq << " inputs";
    cerr << "..." << endl;
    if (count == 0) { cerr << "\nERROR: Did not find sequence records in inputs" << endl; exit(1); }
    // estimateSetSize of the mixture's bottom-s (MinHashHeap.h:45, CommandScreen.cpp:322)
    double est = 0;
    if (mix_n) est = std::pow(2.0, set.p.use64 ? 64.0 : 32.0) * (double)mix_n / (double)mix[mix_n - 1];
    const uint64_t set_size = (uint64_t)est;
    cerr << "   Estimated distinct" << (trans ? " (translated)" : "") << " k-mers in mixture: " << set_size << endl;
    if (set_size == 0) cerr << "WARNING: no valid k-mers in input." << endl;
    cerr << "Summing shared..." << endl;
    vector<uint64_t> shared(n, 0);
    vector<vector<uint64_t>> depths(n);
    for (const mg_screen_hit &h : hits) { shared[h.row]++; depths[h.row].push_back(h.count); }       // (ordered by sketch, then hash)
    const double kspace = set.kmer_space();
    if (c.o("winning!").active) {
        cerr << "Reallocating to winners..." << endl;
        vector<double> scores(n);
        for (uint64_t i = 0; i < n; i++) scores[i] = mg_identity(shared[i], set.refs[i].hashes.size(), set.p.kmer);
        // each observed hash goes to the best-scoring sketch containing it (ties: larger length, then first seen)
        std::map<uint64_t, vector<uint32_t>> owners;
        std::map<uint64_t, uint32_t> obs;
        for (const mg_screen_hit &h : hits) { owners[h.hash].push_back(h.row); obs[h.hash] = h.count; }
        std::fill(shared.begin(), shared.end(), 0);
        for (auto &d : depths) d.clear();
        for (const auto &kv : owners) {
            double max_score = 0;
            uint64_t max_len = 0, max_idx = kv.second[0];
            for (uint32_t k : kv.second) {
                if (scores[k] > max_score) { max_score = scores[k]; max_idx = k; max_len = set.refs[k].length; }
                else if (scores[k] == max_score && set.refs[k].length > max_len) { max_idx = k; max_len = set.refs[k].length; }
            }
            shared[max_idx]++;
            depths[max_idx].push_back(obs[kv.first]);
        }
    }
    cerr << "Computing coverage medians..." << endl;
    for (auto &d : depths) std::sort(d.begin(), d.end());
    cerr << "Writing output..." << endl;
    for (uint64_t i = 0; i < n; i++) {
        if (shared[i] != 0 || identity_min < 0.0) {
            const uint64_t denom = set.refs[i].hashes.size();
            const double identity = mg_identity(shared[i], denom, set.p.kmer);
            if (identity < identity_min) continue;
            const double pv = mg_p_value_within(shared[i], set_size, kspace, denom);
            if (pv > p_max) continue;
            cout << identity << '\t' << shared[i] << '/' << denom << '\t' << (shared[i] > 0 ? depths[i].at(shared[i] / 2) : 0)
                 << '\t' << pv << '\t' << set.refs[i].name << '\t' << set.refs[i].comment << endl;
        }
    }
    return 0;
}

// test/interchange helper: rebuild a .msh from the JSON `mash info -d` prints (no GPU)
int cmd_json2msh(int argc, const char **argv)
{
    if (argc != 2) { cerr << "usage: mash json2msh <info-dump.json> <out.msh>" << endl; return 1; }
    std::ifstream in(argv[0]);
    if (in.fail()) { cerr << "ERROR: Could not open " << argv[0] << endl; return 1; }
    string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    auto find_value = [&](size_t from, const string &key, size_t *at) -> string {
        const string pat = "\"" + key + "\" :";
        size_t p = text.find(pat, from);
        if (p == string::npos) { *at = string::npos; return ""; }
        p += pat.size();
        while (p < text.size() && (text[p] == ' ' || text[p] == '\n' || text[p] == '\t')) p++;
        size_t e;
        string v;
        if (text[p] == '"') { e = text.find("\",\n", p + 1); if (e == string::npos) e = text.find("\"\n", p + 1); v = text.substr(p + 1, e - p - 1); }
        else { e = text.find_first_of(",\n", p); v = text.substr(p, e - p); }
        *at = e;
        return v;
    };
    size_t at;
    mshio::File f;
    f.header.kmer_size = (uint32_t)std::stoul(find_value(0, "kmer", &at));
    f.header.alphabet = find_value(0, "alphabet", &at);
    f.header.has_alphabet = true;
    f.header.preserve_case = find_value(0, "preserveCase", &at) == "true";
    f.header.noncanonical = find_value(0, "canonical", &at) != "true";
    f.header.sketch_size = (uint32_t)std::stoul(find_value(0, "sketchSize", &at));
    f.header.seed = (uint32_t)std::stoul(find_value(0, "hashSeed", &at));
    f.header.concatenated = true;
    size_t pos = text.find("\"sketches\"");
    while (true) {
        size_t a1;
        const string name = find_value(pos, "name", &a1);
        if (a1 == string::npos) break;
        mshio::Reference r;
        r.name = name;
        r.length = std::stoull(find_value(a1, "length", &a1));
        r.comment = find_value(a1, "comment", &a1);
        size_t hb = text.find('[', text.find("\"hashes\"", a1)), he = text.find(']', hb);
        const char *q = text.c_str() + hb + 1, *qe = text.c_str() + he;
        while (q < qe) {
            while (q < qe && (*q < '0' || *q > '9')) q++;
            if (q >= qe) break;
            char *end;
            r.hashes.push_back(strtoull(q, &end, 10));
            q = end;
        }
        f.references.push_back(std::move(r));
        pos = he;
    }
    const string e = mshio::write_msh(argv[1], f);
    if (!e.empty()) { cerr << "ERROR: " << e << endl; return 1; }
    return 0;
}

}  // namespace

int main(int argc, const char **argv)
{
    const string usage =
        "\nMash (MI355X hot path), commands:\n\n  sketch    Create sketches (reduced representations for fast operations).\n"
        "  dist      Estimate the distance of query sequences to references.\n"
        "  triangle  Estimate a lower-triangular distance matrix.\n  info      Display information about sketch files.\n"
        "  paste     Create a single sketch file from multiple sketch files.\n"
        "  screen    Determine whether query sequences are within a larger mixture of sequences.\n\n";
    if (argc < 2) { cout << usage; return 0; }
    const string cmd = argv[1];
    const bool timing = getenv("MASH_AMD_TIMING") != nullptr;
    auto stamp = [&](const char *what) {             // CLOCK_MONOTONIC, comparable with the caller's clock (tools/cli_e2e.py)
        if (timing) cerr << "timing: " << what << ' ' << std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() << endl;
    };
    cerr.precision(12);
    stamp("main_begin");
    cerr.precision(6);
    int rc;
    if (cmd == "sketch") rc = cmd_sketch(argc - 2, argv + 2);
    else if (cmd == "dist") rc = cmd_dist(argc - 2, argv + 2);
    else if (cmd == "triangle") rc = cmd_triangle(argc - 2, argv + 2);
    else if (cmd == "info") rc = cmd_info(argc - 2, argv + 2);
    else if (cmd == "paste") rc = cmd_paste(argc - 2, argv + 2);
    else if (cmd == "screen") rc = cmd_screen(argc - 2, argv + 2);
    else if (cmd == "json2msh") rc = cmd_json2msh(argc - 2, argv + 2);
    else if (cmd == "--version") { cout << "2.3-mi355x" << endl; rc = 0; }
    else {
        cerr << "ERROR: Unknown command: " << cmd << endl;
        cout << usage;
        rc = 1;
    }
    cerr.precision(12);
    stamp("main_end");
    // Everything this process owns is released by the kernel; unloading the HIP runtime and its code
    // objects politely takes tens of milliseconds that a caller of a 0.4 s command waits for.
    if (!getenv("MASH_AMD_SLOW_EXIT")) {
        cout.flush();
        cerr.flush();
        fflush(nullptr);
        _exit(rc);
    }
    return rc;
}
