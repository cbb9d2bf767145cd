// hipemu.h -- a few hundred lines of "HIP on host threads": enough of the kernel language to run the index-build kernels
// of mash_amd/csrc/index_build.hip on the CPU, one workgroup at a time, every work-item an OS thread.
//
// TEST INFRASTRUCTURE ONLY (tests/test_index_emu.py).  There is no GPU in the build container and GPU minutes are
// rationed, so the kernels' index arithmetic and -- under ThreadSanitizer -- their barriers are checked here first:
//   * __syncthreads is a pthread barrier over the workgroup, which TSan models: a missing barrier between an LDS write
//     and another work-item's read is reported as a data race;
//   * wave operations (__shfl_up, __shfl_xor, __ballot, ...) exchange through a per-wave slot array between two wave
//     barriers: all 64 lanes of a wave must reach them, as on the hardware;
//   * __shared__ variables are `static` (workgroups run one after the other), dynamic shared memory is one buffer per launch;
//   * device memory is host memory; streams are synchronous.
// What it does NOT model: the lock step of a wave (code that relies on it without a barrier is reported as a race, which
// is the conservative side), bank conflicts, occupancy, performance of any kind.
// Two back ends: HIPEMU_FIBERS (default for the functional test) runs the work-items of a workgroup as ucontext fibers on ONE
// OS thread -- a barrier is a round of context switches, no system scheduler involved, deterministic, ~50x faster than 512
// threads on 8 cores; without it every work-item is an OS thread (what ThreadSanitizer needs to see).
#pragma once
#include <pthread.h>
#ifdef HIPEMU_FIBERS
#include <ucontext.h>
#endif
#include <stdint.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r; r.x = x; r.y = y; return r; }
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__

namespace hipemu {

#ifdef HIPEMU_FIBERS
struct FBar { unsigned count = 0, arrived = 0, gen = 0; };
struct Wave {
    uint64_t xch[64];
    FBar bar;
};
struct Block {
    FBar bar;
    std::vector<Wave> waves;
    int acc = 0;
    unsigned char *dyn = nullptr;
    unsigned nthreads = 0;
};
#else
struct Wave {
    uint64_t xch[64];
    pthread_barrier_t bar;
};
struct Block {
    pthread_barrier_t bar;
    std::vector<Wave> waves;
    std::atomic<int> acc{0};
    unsigned char *dyn = nullptr;
    unsigned nthreads = 0;
};
#endif
struct ThreadCtx {
    dim3 tid, bid, bdim, gdim;
    unsigned lin = 0;
    Block *blk = nullptr;
};

#ifdef HIPEMU_FIBERS
inline ThreadCtx *cur = nullptr;
struct Fiber {
    ucontext_t ctx;
    ThreadCtx tc;
    bool done = false;
    std::vector<unsigned char> stack;
};
inline std::vector<Fiber> fibers;
inline unsigned cur_f = 0, idle_yields = 0;
inline ucontext_t main_ctx;
inline std::function<void()> *body = nullptr;

inline void switch_from(unsigned me)
{
    const unsigned n = (unsigned)fibers.size();
    unsigned nx = me;
    for (unsigned k = 0; k < n; k++) {
        nx = nx + 1 == n ? 0 : nx + 1;
        if (!fibers[nx].done) break;
    }
    if (fibers[nx].done) {                                  // nobody left: back to the launcher
        setcontext(&main_ctx);
    }
    if (nx == me) return;
    cur_f = nx;
    cur = &fibers[nx].tc;
    swapcontext(&fibers[me].ctx, &fibers[nx].ctx);
}
inline void yield()
{
    if (++idle_yields > 64u * (unsigned)fibers.size() + 4096u) {
        fprintf(stderr, "hipemu: deadlock -- work-items wait at a barrier others never reach (non-uniform control flow?)\n");
        abort();
    }
    switch_from(cur_f);
}
inline void fbar_wait(FBar &b)
{
    const unsigned g = b.gen;
    if (++b.arrived == b.count) {
        b.arrived = 0;
        b.gen++;
        idle_yields = 0;
    } else {
        while (b.gen == g) yield();
    }
}
inline void fiber_main()
{
    (*body)();
    fibers[cur_f].done = true;
    idle_yields = 0;
    switch_from(cur_f);
    abort();                                                // (a finished fiber is never resumed)
}
inline void wave_sync() { fbar_wait(cur->blk->waves[cur->lin >> 6].bar); }
inline void block_sync() { fbar_wait(cur->blk->bar); }
#else
inline thread_local ThreadCtx *cur = nullptr;
inline void wave_sync() { pthread_barrier_wait(&cur->blk->waves[cur->lin >> 6].bar); }
inline void block_sync() { pthread_barrier_wait(&cur->blk->bar); }
#endif
inline Wave &wave() { return cur->blk->waves[cur->lin >> 6]; }

template <class K, class... A>
void launch(K kernel, dim3 grid, dim3 block, size_t shmem, A... args)
{
    const unsigned nt = block.x * block.y * block.z;
    if (nt == 0 || grid.x == 0) return;
    if (nt % 64u) { fprintf(stderr, "hipemu: workgroup of %u work-items is not whole waves\n", nt); abort(); }
    Block B;
    B.nthreads = nt;
    B.waves.resize(nt / 64u);
    std::vector<unsigned char> dyn(shmem + 64);
    B.dyn = dyn.data();
    const unsigned nblocks = grid.x * grid.y * grid.z;
#ifdef HIPEMU_FIBERS
    B.bar.count = nt;
    for (auto &w : B.waves) w.bar.count = 64;
    if (fibers.size() != nt) {
        fibers.clear();
        fibers.resize(nt);
        for (auto &f : fibers) f.stack.resize(96 * 1024);
    }
    std::function<void()> fn = [&]() { kernel(args...); };
    body = &fn;
    for (unsigned b = 0; b < nblocks; b++) {
        B.bar.arrived = 0;
        for (auto &w : B.waves) w.bar.arrived = 0;
        for (unsigned t = 0; t < nt; t++) {
            Fiber &f = fibers[t];
            f.done = false;
            f.tc.blk = &B;
            f.tc.lin = t;
            f.tc.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            f.tc.bdim = block;
            f.tc.gdim = grid;
            f.tc.bid = dim3(b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y));
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack.data();
            f.ctx.uc_stack.ss_size = f.stack.size();
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, (void (*)())fiber_main, 0);
        }
        volatile bool started = false;
        getcontext(&main_ctx);
        if (!started) {
            started = true;
            cur_f = 0;
            cur = &fibers[0].tc;
            idle_yields = 0;
            setcontext(&fibers[0].ctx);
        }
    }
    cur = nullptr;
    body = nullptr;
#else
    pthread_barrier_init(&B.bar, nullptr, nt);
    for (auto &w : B.waves) pthread_barrier_init(&w.bar, nullptr, 64);
    std::vector<std::thread> th;
    th.reserve(nt);
    for (unsigned t = 0; t < nt; t++) {
        th.emplace_back([&, t]() {
            ThreadCtx c;
            c.blk = &B;
            c.lin = t;
            c.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            c.bdim = block;
            c.gdim = grid;
            cur = &c;
            for (unsigned b = 0; b < nblocks; b++) {
                c.bid = dim3(b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y));
                kernel(args...);
                pthread_barrier_wait(&B.bar);              // the next workgroup starts when this one is done (static __shared__)
            }
            cur = nullptr;
        });
    }
    for (auto &t : th) t.join();
    pthread_barrier_destroy(&B.bar);
    for (auto &w : B.waves) pthread_barrier_destroy(&w.bar);
#endif
}

}  // namespace hipemu

#define threadIdx (hipemu::cur->tid)
#define blockIdx (hipemu::cur->bid)
#define blockDim (hipemu::cur->bdim)
#define gridDim (hipemu::cur->gdim)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hipemu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__)
// dynamic shared memory: `extern __shared__ T name[];` is spelled MG_DYN_SHARED(T, name) in code that runs here too
#define MG_DYN_SHARED(T, name) T *name = reinterpret_cast<T *>(hipemu::cur->blk->dyn)

static inline void __syncthreads() { hipemu::block_sync(); }
static inline int __syncthreads_or(int x)
{
    hipemu::Block *b = hipemu::cur->blk;
    hipemu::block_sync();
    if (x) b->acc = 1;
    hipemu::block_sync();
    const int r = b->acc;
    hipemu::block_sync();
    if (hipemu::cur->lin == 0) b->acc = 0;
    return r;
}
static inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }

template <class T> static inline uint64_t hipemu_bits(T v) { uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <class T> static inline T hipemu_unbits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

template <class T> static inline T __shfl_up(T v, unsigned d, int = 64)
{
    hipemu::Wave &w = hipemu::wave();
    const unsigned lane = hipemu::cur->lin & 63u;
    w.xch[lane] = hipemu_bits(v);
    hipemu::wave_sync();
    const T r = lane >= d ? hipemu_unbits<T>(w.xch[lane - d]) : v;
    hipemu::wave_sync();
    return r;
}
template <class T> static inline T __shfl_xor(T v, unsigned d, int = 64)
{
    hipemu::Wave &w = hipemu::wave();
    const unsigned lane = hipemu::cur->lin & 63u;
    w.xch[lane] = hipemu_bits(v);
    hipemu::wave_sync();
    const T r = hipemu_unbits<T>(w.xch[(lane ^ d) & 63u]);
    hipemu::wave_sync();
    return r;
}
template <class T> static inline T __shfl(T v, unsigned src, int = 64)
{
    hipemu::Wave &w = hipemu::wave();
    const unsigned lane = hipemu::cur->lin & 63u;
    w.xch[lane] = hipemu_bits(v);
    hipemu::wave_sync();
    const T r = hipemu_unbits<T>(w.xch[src & 63u]);
    hipemu::wave_sync();
    return r;
}
static inline uint64_t __ballot(int pred)
{
    hipemu::Wave &w = hipemu::wave();
    const unsigned lane = hipemu::cur->lin & 63u;
    w.xch[lane] = pred ? 1u : 0u;
    hipemu::wave_sync();
    uint64_t m = 0;
    for (unsigned l = 0; l < 64; l++) m |= (uint64_t)(w.xch[l] & 1u) << l;
    hipemu::wave_sync();
    return m;
}
static inline int __popcll(uint64_t x) { return __builtin_popcountll(x); }
static inline int __popc(uint32_t x) { return __builtin_popcount(x); }

// atomics on "device" and "shared" memory alike
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v)
{
    unsigned long long o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o > v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline uint32_t atomicMax(uint32_t *p, uint32_t v)
{
    uint32_t o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
