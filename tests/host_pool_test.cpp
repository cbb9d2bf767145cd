// tests/host_pool_test.cpp -- CPU test of mash_amd/host/parse_pool.h (built and run by tests/test_host_pool.py):
// files parsed ahead by N worker threads come back strictly in input order and equal to the sequential parse --
// whatever the thread count, the look-ahead limit, and the way the consumer gathers them (one by one, or in runs of
// files that are ready, as init_from_files does); unreadable and empty files come back as that file's error; files
// the consumer handles itself are skipped; run_jobs spreads work over the same threads while they parse.
#include <zlib.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../mash_amd/host/parse_pool.h"

using namespace hostpool;

static bool same(const ParsedFile &a, const ParsedFile &b)
{
    return a.ref.name == b.ref.name && a.ref.comment == b.ref.comment && a.ref.length == b.ref.length && a.bases == b.bases &&
           a.error == b.error;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: host_pool_test <scratch dir>\n"); return 2; }
    const std::string dir = argv[1];
    std::mt19937_64 rng(7);
    const int kmer = 21;
    std::vector<std::string> files;
    const char acgt[] = "ACGTNacgt";
    for (int i = 0; i < 160; i++) {
        std::string path = dir + "/f" + std::to_string(i) + (i % 23 == 5 ? ".fa.gz" : ".fa");
        if (i == 40) { files.push_back(dir + "/does_not_exist.fa"); continue; }
        std::string body;
        if (i != 41) {                                      // 41: an empty file
            const int nrec = 1 + (int)(rng() % 4);
            for (int r = 0; r < nrec; r++) {
                size_t len = (i % 9 == 0) ? rng() % 30 : rng() % (i % 7 == 0 ? 300000 : 20000);   // some records shorter than k
                if (i == 42) len = 5;                       // 42: nothing but short records
                body += ">rec" + std::to_string(r) + " file " + std::to_string(i) + "\n";
                for (size_t q = 0; q < len; q++) {
                    body.push_back(acgt[rng() % 9]);
                    if (q % 70 == 69) body.push_back('\n');
                }
                body.push_back('\n');
            }
        }
        if (i % 23 == 5) {
            gzFile g = gzopen(path.c_str(), "wb");
            gzwrite(g, body.data(), (unsigned)body.size());
            gzclose(g);
        } else {
            FILE *f = fopen(path.c_str(), "wb");
            fwrite(body.data(), 1, body.size(), f);
            fclose(f);
        }
        files.push_back(path);
    }
    std::vector<ParsedFile> want;
    for (const std::string &f : files) want.push_back(parse_file_concatenated(f, kmer));
    if (want[40].error.find("could not open") == std::string::npos) { fprintf(stderr, "missing file not reported\n"); return 1; }
    if (want[41].error.find("Did not find fasta records") == std::string::npos) { fprintf(stderr, "empty file not reported\n"); return 1; }
    if (want[42].error.find("shorter than the k-mer size") == std::string::npos) { fprintf(stderr, "short records not reported\n"); return 1; }
    const size_t skipped = 7;                               // a file the consumer handles itself (a .msh among the inputs)
    for (const char *ahead : {"", "1", "100000"}) {
        if (*ahead) setenv("MASH_AMD_PARSE_AHEAD", ahead, 1); else unsetenv("MASH_AMD_PARSE_AHEAD");
        for (size_t threads : {1u, 3u, 8u, 32u}) {
            for (int gather = 0; gather < 2; gather++) {
                ParsePool pool(files, threads, [&](size_t i) { return i != skipped; });
                std::atomic<uint64_t> acc{0};
                size_t got = 0;
                for (size_t i = 0; i < files.size(); i++) {
                    if (i == skipped) { pool.skip(i); got++; continue; }
                    std::vector<ParsedFile> run;
                    run.push_back(pool.take(i, kmer));
                    while (gather && i + 1 < files.size() && i + 1 != skipped && pool.ready_bytes(i + 1) >= 0 && run.size() < 9) run.push_back(pool.take(++i, kmer));
                    for (size_t k = 0; k < run.size(); k++) {
                        const size_t idx = i + 1 - run.size() + k;
                        if (!same(run[k], want[idx])) { fprintf(stderr, "file %zu differs (threads %zu, ahead '%s', gather %d)\n", idx, threads, ahead, gather); return 1; }
                        got++;
                    }
                    if (i % 13 == 0) {                      // work dealt to the pool between two files
                        pool.run_jobs(50, [&](size_t j) { acc.fetch_add((j + 1) * (j + 1)); });
                    }
                }
                if (got != files.size()) { fprintf(stderr, "%zu of %zu files\n", got, files.size()); return 1; }
                uint64_t expect = 0, rounds = 0;
                for (size_t i = 0; i < files.size(); i++) rounds += 0;
                (void)rounds;
                for (uint64_t j = 1; j <= 50; j++) expect += j * j;
                if (acc.load() % expect != 0 || acc.load() == 0) { fprintf(stderr, "run_jobs lost work: %llu\n", (unsigned long long)acc.load()); return 1; }
            }
        }
    }
    printf("OK %zu files\n", files.size());
    return 0;
}
