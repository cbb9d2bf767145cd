// parse_pool.h -- the host half of the reference's sketch workers (Sketch.cpp:1147-1336, ThreadPool.hxx:127-167):
// kseq-compatible parsing of ONE input file in concatenated mode, and the pool of worker threads that parses files
// ahead of the consumer, hands them over strictly in input order and lends its threads for other jobs in between.
// Host only (no GPU call): tests/host_pool_test.cpp exercises it on the CPU; mash_main.cpp is its user.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mashgpu.h"
#include "fastx.h"

namespace hostpool {

using std::string;
using std::vector;

struct Ref {
    string name, comment;
    uint64_t length = 0;
    vector<uint64_t> hashes;
    vector<uint32_t> counts;
};

// sketchFile in concatenated mode for ONE file (Sketch.cpp:1147-1336), non-reads: the host half
// (kseq parse, name/comment/length) -- runs on a worker thread with -p > 1, like the reference's
// ThreadPool workers, while the k-mer work of earlier files is on the GPU.
struct ParsedFile {
    Ref ref;
    vector<uint8_t> bases;                    // records >= k, each followed by MG_RECORD_SEP
    string error;                             // fatal message (printed by the consumer, in input order)
    bool warn_only = false;
};

inline ParsedFile parse_file_concatenated(const string &file, int kmer)
{
    ParsedFile out;
    fastx::Reader rd;
    if (!rd.open(file)) { out.error = "ERROR: could not open " + file + " for reading."; return out; }
    Ref &ref = out.ref;
    if (file != "-") ref.name = file;
    fastx::Record rec;
    long l;
    int count = 0;
    bool skipped = false;
    while ((l = rd.next(rec)) >= 0) {
        if (l < kmer) { skipped = true; continue; }
        if (count == 0) {
            if (file == "-") { ref.name = rec.name; ref.comment = rec.comment; }
            else ref.comment = rec.name + " " + rec.comment;
        }
        count++;
        ref.length += (uint64_t)l;
        out.bases.insert(out.bases.end(), rec.seq.begin(), rec.seq.end());
        out.bases.push_back((uint8_t)MG_RECORD_SEP);
    }
    if (count > 1) ref.comment = "[" + std::to_string(count) + " seqs] " + ref.comment + " [...]";
    if (l != -1) { out.error = "\nERROR: reading input files."; return out; }
    if (ref.length == 0) {
        if (skipped) out.error = "\nWARNING: All fasta records in input files were shorter than the k-mer size (" + std::to_string(kmer) + ").";
        else out.error = "\nERROR: Did not find fasta records in \"input files\".";
    }
    return out;
}

// Sketch::initFromFiles (Sketch.cpp:105-253)
// -p N: N worker threads parse input files ahead of the consumer (decompress + kseq parse, the
// host half of the reference's ThreadPool workers).  Files are claimed in input order, at most
// `window` positions ahead of the file the consumer is waiting for, and handed over strictly in
// input order, so the output does not depend on N.  The workers live as long as the pool: one
// thread per FILE (std::async) costs more than parsing a small genome.
class ParsePool {
public:
    struct Copy { uint8_t *dst; const uint8_t *src; size_t n; };

    ParsePool(const vector<string> &files, size_t threads, std::function<bool(size_t)> parseable)
        : files_(files), parseable_(std::move(parseable)), nthreads_(threads),
          window_(std::max<size_t>(64, std::min<size_t>(files.size(), 1 << 14))), ring_(window_)
    {
        if (const char *e = getenv("MASH_AMD_PARSE_AHEAD")) ahead_limit_ = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
    }
    ~ParsePool()
    {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_work_.notify_all();
        for (auto &t : workers_) t.join();
    }
    // Start the workers on files[first..] (k must be final).  `mash sketch` calls this BEFORE the device
    // context exists: its creation (0.15-0.25 s) is then spent parsing, up to ahead_limit_ bytes.
    void start(size_t first, int kmer)
    {
        std::lock_guard<std::mutex> g(m_);
        if (!workers_.empty()) return;
        kmer_ = kmer;
        next_.store(first);
        pos_.store(first);
        for (size_t t = 0; t < nthreads_; t++) workers_.emplace_back([this]() { work(); });
    }
    // file i, which the consumer handles itself (.msh, stdin), is behind us
    void skip(size_t i) { advance(i + 1); }
    // bytes of file i if a worker has finished it, else -1 (never blocks)
    long long ready_bytes(size_t i)
    {
        Slot &sl = ring_[i % window_];
        if (sl.state.load(std::memory_order_acquire) != 1 || sl.index != i) return -1;
        return (long long)sl.pf.bases.size();
    }
    // result for file i (parsed by a worker, or here if no worker got to it); indices must ascend
    ParsedFile take(size_t i, int kmer)
    {
        if (workers_.empty()) start(i, kmer);
        if (pos_.load() < i) advance(i);
        Slot &sl = ring_[i % window_];
        bool mine = false;
        {
            std::lock_guard<std::mutex> lk(m_);
            if (next_.load() < i) next_.store(i);
            if (next_.load() == i) { next_.store(i + 1); mine = true; }      // nobody has claimed it: parse on this thread
        }
        ParsedFile pf;
        if (mine) {
            pf = parse_file_concatenated(files_[i], kmer);
        } else {
            for (int spin = 0; !(sl.state.load(std::memory_order_acquire) == 1 && sl.index == i); spin++) {
                if (spin < 2000) { std::this_thread::yield(); continue; }
                std::unique_lock<std::mutex> lk(m_);             // a slow file (gzip, a large genome): sleep
                consumer_waits_.store(true);
#ifdef HOSTPOOL_UNTIMED_WAIT                                    // (ThreadSanitizer builds: gcc 11's libtsan does not model timed waits)
                cv_done_.wait(lk, [&]() { return sl.state.load(std::memory_order_acquire) == 1 && sl.index == i; });
#else
                cv_done_.wait_for(lk, std::chrono::milliseconds(2),
                                  [&]() { return sl.state.load(std::memory_order_acquire) == 1 && sl.index == i; });
#endif
                consumer_waits_.store(false);
            }
            pf = std::move(sl.pf);
            sl.pf = ParsedFile();
            held_.fetch_sub(pf.bases.size());
            sl.state.store(0, std::memory_order_release);
        }
        advance(i + 1);
        return pf;
    }
    // fn(0) ... fn(n - 1) spread over the workers and the caller; returns when all have been carried out
    // (workers in the middle of a file join when they are through with it)
    void run_jobs(size_t n, const std::function<void(size_t)> &fn)
    {
        if (n == 0) return;
        std::unique_lock<std::mutex> lk(m_);
        if (workers_.empty() || n == 1) {
            lk.unlock();
            for (size_t j = 0; j < n; j++) fn(j);
            return;
        }
        job_fn_ = &fn;
        job_count_ = n;
        job_next_ = 0;
        job_left_ = n;
        cv_work_.notify_all();
        while (job_next_ < n) {
            const size_t j = job_next_++;
            lk.unlock();
            fn(j);
            lk.lock();
            job_left_--;
        }
        cv_copy_.wait(lk, [&]() { return job_left_ == 0; });
        job_fn_ = nullptr;
    }
    void copy_all(const vector<Copy> &jobs)
    {
        run_jobs(jobs.size(), [&jobs](size_t j) { memcpy(jobs[j].dst, jobs[j].src, jobs[j].n); });
    }

private:
    struct Slot {
        std::atomic<int> state{0};                       // 0: free, 1: pf holds file `index`
        size_t index = 0;
        ParsedFile pf;
    };
    void advance(size_t pos)
    {
        pos_.store(pos);
        // Only workers held back by the look-ahead limits can use this news (a worker raises limit_waiters_
        // under m_ BEFORE it tests pos_).  Workers that are idle because every file has been claimed must
        // NOT be woken here: 16 of them, 12 000 times, is what the consumer then spends its time on.
        if (limit_waiters_.load() > 0 && next_.load() < files_.size()) {
            std::lock_guard<std::mutex> g(m_);
            cv_work_.notify_all();
        }
    }
    bool can_claim() const { return next_.load() < files_.size() && next_.load() < pos_.load() + window_ && held_.load() < ahead_limit_; }
    void work()
    {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            const bool limited = next_.load() < files_.size();          // if it has to wait, then for the consumer to move on
            if (limited) limit_waiters_.fetch_add(1);
            cv_work_.wait(lk, [&]() { return stop_ || (job_fn_ && job_next_ < job_count_) || can_claim(); });
            if (limited) limit_waiters_.fetch_sub(1);
            if (stop_) return;
            if (job_fn_ && job_next_ < job_count_) {
                const size_t j = job_next_++;
                const std::function<void(size_t)> &fn = *job_fn_;
                lk.unlock();
                fn(j);
                lk.lock();
                if (--job_left_ == 0) cv_copy_.notify_all();
                continue;
            }
            const size_t i = next_.fetch_add(1);
            if (!parseable_(i)) continue;                // .msh / stdin: the consumer handles those itself
            lk.unlock();
            ParsedFile pf = parse_file_concatenated(files_[i], kmer_);
            Slot &sl = ring_[i % window_];
            held_.fetch_add(pf.bases.size());
            sl.pf = std::move(pf);
            sl.index = i;
            sl.state.store(1, std::memory_order_release);
            lk.lock();
            if (consumer_waits_.load()) cv_done_.notify_all();
        }
    }
    const vector<string> &files_;
    std::function<bool(size_t)> parseable_;
    size_t nthreads_, window_;
    vector<Slot> ring_;
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_, cv_copy_;
    vector<std::thread> workers_;
    std::atomic<size_t> next_{0};                        // next file to claim (written under m_)
    std::atomic<size_t> pos_{0};                         // files below have been taken by the consumer
    std::atomic<uint64_t> held_{0};                      // bytes parsed and not yet taken
    uint64_t ahead_limit_ = 1ull << 30;
    std::atomic<int> limit_waiters_{0};
    std::atomic<bool> consumer_waits_{false};
    const std::function<void(size_t)> *job_fn_ = nullptr;
    size_t job_count_ = 0, job_next_ = 0, job_left_ = 0;
    int kmer_ = 0;
    bool stop_ = false;
};

}  // namespace hostpool
