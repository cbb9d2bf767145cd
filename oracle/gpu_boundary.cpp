// oracle/gpu_boundary.cpp -- TEST INFRASTRUCTURE (never shipped): the REFERENCE CLI with its hot path bound
// to libmashgpu.so, i.e. INTEGRATION.md sections 1-2 as a patch that compiles.
//
// `make -C oracle refcli-gpu` links all 21 of the reference's translation units, unmodified, as
// `refcli` does -- but three of its symbols are made weak in copies of their object files (objcopy; an
// alias keeps the original reachable as ref_*) and defined again HERE, calling the C ABI of include/mashgpu.h:
//
//   Sketch::initFromFiles                      (Sketch.cpp:105-253)          -> batch sketching, mg_sketch_host
//   mash::compare(CommandDistance::CompareInput*)   (CommandDistance.cpp:306-334) -> mg_compare_rect_pairs_host
//   mash::compare(CommandTriangle::TriangleInput*)  (CommandTriangle.cpp:200-214) -> mg_compare_tri_pairs_host
//
// Everything else -- option parsing, ThreadPool, kseq, the writers, the .msh codec behind the shim --
// is the reference's own code.  What the replacement does not cover (sketch files among the inputs,
// stdin, reads mode, -i, windowed sketches) goes to the original through its alias.  tests/test_refcli_gpu.py
// runs the reference's three `make test` recipes and ten CLI fixtures through the resulting binary.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#define private public                       // parameters / createIndex() of Sketch (a maintainer edits the class itself)
#include "mash/Sketch.h"
#undef private
#include "mash/CommandDistance.h"
#include "mash/CommandTriangle.h"
#include "mash/kseq.h"
#include "mashgpu.h"

KSEQ_INIT(gzFile, gzread)

using namespace std;

// the original under its alias (Itanium ABI: a member function takes `this` first)
extern "C" int ref_Sketch_initFromFiles(Sketch *self, const vector<string> &files, const Sketch::Parameters &p, int verbosity,
                                         bool enforceParameters, bool contain);

namespace {

mg_ctx *gpu()
{
    static mg_ctx *ctx = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        if (mg_ctx_create(0, &ctx) != MG_OK) {
            cerr << "ERROR: no usable GPU: " << mg_last_error(nullptr) << endl;
            exit(1);
        }
    });
    return ctx;
}

// once per loaded Sketch: vector<Reference> -> dense table, kept for the run
mg_table *table_of(const Sketch &sk)
{
    static std::mutex mu;
    static std::map<const Sketch *, mg_table *> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(&sk);
    if (it != cache.end()) return it->second;
    const uint64_t n = sk.getReferenceCount(), s = (uint64_t)sk.getMinHashesPerWindow();
    vector<uint64_t> h(n * s, MG_HASH_PAD), len(n);
    vector<uint32_t> nh(n);
    for (uint64_t i = 0; i < n; i++) {
        const Sketch::Reference &r = sk.getReference(i);
        nh[i] = (uint32_t)std::min<uint64_t>(r.hashesSorted.size(), s);
        len[i] = r.length;
        for (uint32_t k = 0; k < nh[i]; k++)
            h[i * s + k] = r.hashesSorted.get64() ? r.hashesSorted.at(k).hash64 : r.hashesSorted.at(k).hash32;
    }
    mg_table *t = nullptr;
    if (mg_table_upload(gpu(), h.data(), nh.data(), len.data(), n, s, &t) != MG_OK) {
        cerr << "ERROR: " << mg_last_error(gpu()) << endl;
        exit(1);
    }
    cache[&sk] = t;
    return t;
}

void to_pair(mash::CommandDistance::CompareOutput::PairOutput &o, const mg_pair &p)
{
    o.numer = p.numer;
    o.denom = p.denom;
    o.distance = p.distance;
    o.pValue = p.p_value;
    o.pass = p.pass != 0;
}

}  // namespace

// ---- sketching ------------------------------------------------------------
int Sketch::initFromFiles(const vector<string> &files, const Parameters &parametersNew, int verbosity, bool enforceParameters, bool contain)
{
    bool batch = !parametersNew.reads && parametersNew.concatenated && !parametersNew.windowed && !files.empty();
    for (const string &f : files)
        if (f == "-" || hasSuffix(f, suffixSketch) || hasSuffix(f, suffixSketchWindowed)) batch = false;
    if (!batch) return ref_Sketch_initFromFiles(this, files, parametersNew, verbosity, enforceParameters, contain);

    parameters = parametersNew;
    mg_params p;
    string alpha;
    getAlphabetAsString(alpha);
    if (mg_params_init(&p, parameters.kmerSize, parameters.minHashesPerWindow, parameters.seed, alpha.c_str(),
                       parameters.noncanonical, parameters.preserveCase) != MG_OK) {
        cerr << "ERROR: parameters outside the device path" << endl;
        exit(1);
    }
    p.counts = parameters.counts;
    // host ingest stays kseq: one byte range per sketch, records joined with MG_RECORD_SEP
    vector<uint8_t> bases;
    vector<uint64_t> off(1, 0);
    Sketch::SketchOutput *out = new Sketch::SketchOutput();
    for (const string &f : files) {                              // concatenated mode (Sketch.cpp:1147-1336)
        if (verbosity > 0) cerr << "Sketching " << f << "..." << endl;
        Reference ref;
        ref.name = f;
        ref.length = 0;
        ref.countsSorted = false;
        int count = 0;
        bool skipped = false;
        gzFile fp = gzopen(f.c_str(), "r");
        if (fp == 0) {
            cerr << "ERROR: could not open " << f << " for reading." << endl;
            exit(1);
        }
        kseq_t *ks = kseq_init(fp);
        int l;
        while ((l = kseq_read(ks)) >= 0) {
            if (l < parameters.kmerSize) { skipped = true; continue; }          // :1222-1226: skipped, not counted
            if (count == 0) ref.comment = string(ks->name.s) + " " + (ks->comment.s ? ks->comment.s : "");      // :1228-1242
            count++;
            ref.length += l;                                     // :1253
            bases.insert(bases.end(), ks->seq.s, ks->seq.s + l);
            bases.push_back(MG_RECORD_SEP);
        }
        kseq_destroy(ks);
        gzclose(fp);
        if (count > 1) ref.comment = "[" + to_string(count) + " seqs] " + ref.comment + " [...]";      // :1284-1292
        if (l != -1) { cerr << "\nERROR: reading input files." << endl; exit(1); }                 // :1294-1298
        if (ref.length == 0) {                                                                     // :1300-1312
            if (skipped) cerr << "\nWARNING: All fasta records in input files were shorter than the k-mer size (" << parameters.kmerSize << ")." << endl;
            else cerr << "\nERROR: Did not find fasta records in \"input files\"." << endl;
            exit(1);
        }
        off.push_back(bases.size());
        out->references.push_back(ref);
    }
    const uint64_t n = out->references.size(), s = parameters.minHashesPerWindow;
    vector<uint64_t> hashes(n * s);
    vector<uint32_t> nhash(n), counts(parameters.counts ? n * s : 0);
    if (mg_sketch_host(gpu(), &p, bases.data(), bases.size(), off.data(), n, hashes.data(), nhash.data(),
                       parameters.counts ? counts.data() : nullptr) != MG_OK) {
        cerr << "ERROR: " << mg_last_error(gpu()) << endl;
        exit(1);
    }
    for (uint64_t i = 0; i < n; i++) {                           // setMinHashesForReference (:1139-1145)
        Reference &r = out->references[i];
        r.hashesSorted.setUse64(parameters.use64);
        for (uint32_t h = 0; h < nhash[i]; h++) {
            if (parameters.use64) r.hashesSorted.push_back64(hashes[i * s + h]);
            else r.hashesSorted.push_back32((uint32_t)hashes[i * s + h]);
        }
        if (parameters.counts) {
            r.counts.assign(counts.begin() + i * s, counts.begin() + i * s + nhash[i]);
            r.countsSorted = true;
        }
    }
    useThreadOutput(out);                                        // appends the references (and deletes `out`)
    createIndex();
    return 0;
}

namespace mash {

// ---- mash dist -- one job = pairCount cells of the query-major grid from
// (indexQuery, indexRef) (CommandDistance.cpp:306-334)
CommandDistance::CompareOutput *compare(CommandDistance::CompareInput *input)
{
    const Sketch &sketchRef = input->sketchRef, &sketchQuery = input->sketchQuery;
    CommandDistance::CompareOutput *output = new CommandDistance::CompareOutput(sketchRef, sketchQuery, input->indexRef, input->indexQuery, input->pairCount);
    const uint64_t nref = sketchRef.getReferenceCount();
    const uint64_t first = input->indexQuery * nref + input->indexRef, last = first + input->pairCount;   // flat cells [first, last)
    const uint64_t q0 = first / nref, q1 = (last + nref - 1) / nref;
    vector<mg_pair> pairs((q1 - q0) * nref);
    if (mg_compare_rect_pairs_host(gpu(), table_of(sketchRef), table_of(sketchQuery), q0, q1, sketchRef.getKmerSize(), sketchRef.getKmerSpace(),
                                   input->maxDistance, input->maxPValue, pairs.data()) != MG_OK) {
        cerr << "ERROR: " << mg_last_error(gpu()) << endl;
        exit(1);
    }
    for (uint64_t c = first; c < last; c++) to_pair(output->pairs[c - first], pairs[c - q0 * nref]);
    return output;
}

// mash triangle -- one job = row `index` (CommandTriangle.cpp:200-214)
CommandTriangle::TriangleOutput *compare(CommandTriangle::TriangleInput *input)
{
    const Sketch &sketch = input->sketch;
    CommandTriangle::TriangleOutput *output = new CommandTriangle::TriangleOutput(sketch, input->index);
    if (input->index == 0) return output;
    vector<mg_pair> pairs(input->index);
    if (mg_compare_tri_pairs_host(gpu(), table_of(sketch), input->index, input->index + 1, sketch.getKmerSize(), sketch.getKmerSpace(),
                                  input->maxDistance, input->maxPValue, pairs.data()) != MG_OK) {
        cerr << "ERROR: " << mg_last_error(gpu()) << endl;
        exit(1);
    }
    for (uint64_t j = 0; j < input->index; j++) to_pair(output->pairs[j], pairs[j]);
    return output;
}

}  // namespace mash
