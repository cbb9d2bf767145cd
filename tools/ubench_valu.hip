// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the integer VALU
// ops the hot kernels are made of, on gfx950.  One wave per SIMD (256 threads/CU, 1 block/CU),
// 8 independent chains per lane so latency is hidden; s_memtime brackets the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITER 32768

#define DEF_KERNEL(NAME, BODY)                                                         \
    __global__ void NAME(uint64_t *out, uint32_t seed)                                 \
    {                                                                                  \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3; \
        uint32_t a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;   \
        uint64_t b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;    \
        const uint32_t c = seed | 0x9E3779B1u;                                         \
        const uint64_t c64 = 0x87c37b91114253d5ULL ^ seed;                             \
        (void)c64; (void)c;                                                            \
        uint64_t t0 = __builtin_readcyclecounter();                                    \
        for (int i = 0; i < ITER; i++) { BODY }                                        \
        uint64_t t1 = __builtin_readcyclecounter();                                    \
        uint64_t acc = (uint64_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7; \
        if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;                           \
        if (acc == 0x1234567) out[blockIdx.x * 2 + 1] = acc;                           \
    }

#define R8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define R8B(OP) OP(b0) OP(b1) OP(b2) OP(b3) OP(b4) OP(b5) OP(b6) OP(b7)

#define OP_ADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_MUL24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_MAD24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x) : "v"(c));
#define OP_MAD64(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"((uint32_t)x), "v"(c) : "vcc");
#define OP_LSHL64(x) asm volatile("v_lshlrev_b64 %0, 7, %0" : "+v"(x));
#define OP_LSHLADD64(x) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(x) : "v"(c64));
#define OP_ALIGNBIT(x) asm volatile("v_alignbit_b32 %0, %0, %1, 13" : "+v"(x) : "v"(c));
#define OP_CMPLT64(x) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(x), "v"(c64) : "vcc");
#define OP_CMPLT32(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x), "v"(c) : "vcc");
#define OP_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(c) : "vcc");
#define OP_MUL64(x) x *= c64;
#define OP_CNDMASK_S(x) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(x) : "v"(c) : "s10", "s11");
#define OP_CNDMASK_IND(x) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(c), "v"(seed) : "vcc");
#define OP_BFI(x) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x) : "v"(c), "v"(seed));
#define OP_SUBB(x) asm volatile("v_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(c) : "vcc");
#define OP_CMPCND(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(c) : "vcc");
#define OP_CMPCND_S(x) asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(x) : "v"(c) : "s10", "s11");
#define OP_MINU(x) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(c));
#define OP_MAX3(x) asm volatile("v_max3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
#define OP_MOV(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(c));
#define OP_READLANE(x) asm volatile("v_readlane_b32 s10, %0, 5" : : "v"(x) : "s10");
#define OP_MBCNT(x) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(x) : "v"(c));
#define OP_ANDOR(x) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
#define OP_XOR3(x) asm volatile("v_xor3_b32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
#define OP_LSHLOR(x) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x) : "v"(c));
#define OP_ADD3(x) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
#define OP_PERM(x) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
#define OP_BFE(x) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(x));
#define OP_ADDCO(x) asm volatile("v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(x) : : "vcc");

DEF_KERNEL(k_add, R8(OP_ADD))
DEF_KERNEL(k_xor, R8(OP_XOR))
DEF_KERNEL(k_mullo, R8(OP_MULLO))
DEF_KERNEL(k_mulhi, R8(OP_MULHI))
DEF_KERNEL(k_mul24, R8(OP_MUL24))
DEF_KERNEL(k_mad24, R8(OP_MAD24))
DEF_KERNEL(k_mad64, R8B(OP_MAD64))
DEF_KERNEL(k_lshl64, R8B(OP_LSHL64))
DEF_KERNEL(k_lshladd64, R8B(OP_LSHLADD64))
DEF_KERNEL(k_alignbit, R8(OP_ALIGNBIT))
DEF_KERNEL(k_cmplt64, R8B(OP_CMPLT64))
DEF_KERNEL(k_cmplt32, R8(OP_CMPLT32))
DEF_KERNEL(k_cndmask, R8(OP_CNDMASK))
DEF_KERNEL(k_mul64_cxx, R8B(OP_MUL64))
DEF_KERNEL(k_add3, R8(OP_ADD3))
DEF_KERNEL(k_perm, R8(OP_PERM))
DEF_KERNEL(k_bfe, R8(OP_BFE))
DEF_KERNEL(k_addco, R8(OP_ADDCO))
DEF_KERNEL(k_cnd_s, R8(OP_CNDMASK_S))
DEF_KERNEL(k_cnd_ind, R8(OP_CNDMASK_IND))
DEF_KERNEL(k_bfi, R8(OP_BFI))
DEF_KERNEL(k_subb, R8(OP_SUBB))
DEF_KERNEL(k_cmpcnd, R8(OP_CMPCND))
DEF_KERNEL(k_cmpcnd_s, R8(OP_CMPCND_S))
DEF_KERNEL(k_minu, R8(OP_MINU))
DEF_KERNEL(k_max3, R8(OP_MAX3))
DEF_KERNEL(k_mov, R8(OP_MOV))
DEF_KERNEL(k_readlane, R8(OP_READLANE))
DEF_KERNEL(k_mbcnt, R8(OP_MBCNT))
DEF_KERNEL(k_andor, R8(OP_ANDOR))
DEF_KERNEL(k_lshlor, R8(OP_LSHLOR))

typedef void (*kern_t)(uint64_t *, uint32_t);

int main()
{
    struct { const char *name; kern_t k; } tests[] = {
        {"v_add_u32", k_add}, {"v_xor_b32", k_xor}, {"v_add3_u32", k_add3}, {"v_bfe_u32", k_bfe},
        {"v_perm_b32", k_perm}, {"v_alignbit_b32", k_alignbit}, {"v_cndmask_b32", k_cndmask},
        {"v_addc_co_u32", k_addco}, {"v_cmp_lt_u32", k_cmplt32}, {"v_cmp_lt_u64", k_cmplt64},
        {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24}, {"v_mul_lo_u32", k_mullo},
        {"v_mul_hi_u32", k_mulhi}, {"v_mad_u64_u32", k_mad64}, {"v_lshlrev_b64", k_lshl64},
        {"v_lshl_add_u64", k_lshladd64}, {"u64 *= const (C++)", k_mul64_cxx},
        {"v_cndmask e64 sgpr", k_cnd_s}, {"v_cndmask indep dst", k_cnd_ind}, {"v_bfi_b32", k_bfi},
        {"v_subb_co_u32", k_subb}, {"v_cmp+v_cndmask vcc", k_cmpcnd}, {"v_cmp+nop+cndmask sgpr", k_cmpcnd_s},
        {"v_min_u32", k_minu}, {"v_max3_u32", k_max3}, {"v_mov_b32", k_mov}, {"v_readlane_b32", k_readlane},
    };
    uint64_t *d;
    hipMalloc(&d, 4096 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves : {1, 2, 4, 8}) {
        // 1/2/4 waves per SIMD: one block per CU of 256/512/1024 threads; 8: two 1024-thread blocks per CU
        const int bt = waves >= 4 ? 1024 : 256 * waves;
        const int grid = waves == 8 ? 512 : 256;
        printf("--- %d wave(s) per SIMD (block %d threads, grid %d): wall-clock throughput ---\n", waves, bt, grid);
        for (auto &t : tests) {
            hipLaunchKernelGGL(t.k, dim3(grid), dim3(bt), 0, 0, d, 12345u);   // warm
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(t.k, dim3(grid), dim3(bt), 0, 0, d, 12345u);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            std::vector<uint64_t> h(512);
            hipMemcpy(h.data(), d, 512 * 8, hipMemcpyDeviceToHost);
            const double winstr_per_simd = (double)ITER * 8 * waves;           // wave-instructions each SIMD executed
            const double ns = ms * 1e6 / winstr_per_simd;
            printf("%-22s %8.3f ms   %6.3f ns per wave-instr per SIMD (= %5.2f cycles @2.4GHz)  ticks/instr/wave %6.2f\n",
                   t.name, ms, ns, ns * 2.4, (double)h[0] / (ITER * 8.0));
        }
    }
    hipFree(d);
    return 0;
}
