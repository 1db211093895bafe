// dev microbenchmark: issue rate of the integer VALU instructions the JPEG kernels are made of (wave64, gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define OP_LOOP(NAME, BODY)                                                                   \
    __global__ __launch_bounds__(256) void NAME(int* out, int a, int b, int iters)           \
    {                                                                                         \
        int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < iters; i++) {                                                     \
            _Pragma("unroll") for (int k = 0; k < 8; k++) { BODY }                            \
        }                                                                                     \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;          \
    }
#define EACH(F) x0 = F(x0); x1 = F(x1); x2 = F(x2); x3 = F(x3); x4 = F(x4); x5 = F(x5); x6 = F(x6); x7 = F(x7);
#define F_ADD(x) ((x) + a)
#define F_MAD24(x) (__builtin_amdgcn_mul_i24((x), a) + b)
__device__ __forceinline__ int f_mad24(int x, int a, int b) { int r; asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(a), "v"(b)); return r; }
__device__ __forceinline__ int f_mullo(int x, int a) { int r; asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r) : "v"(x), "s"(a)); return r; }
__device__ __forceinline__ int f_bfe(int x) { int r; asm volatile("v_bfe_i32 %0, %1, 3, 10" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ int f_med3(int x, int b) { int r; asm volatile("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(b)); return r; }
__device__ __forceinline__ int f_lshlor(int x, int b) { int r; asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(r) : "v"(x), "v"(b)); return r; }
__device__ __forceinline__ int f_add3(int x, int a, int b) { int r; asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(a), "v"(b)); return r; }
__device__ __forceinline__ int f_addv(int x, int b) { int r; asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(b)); return r; }
__device__ __forceinline__ int f_cnd(int x, int b) { int r; asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r) : "v"(x), "v"(b)); return r; }
__device__ __forceinline__ int f_ashr(int x) { int r; asm volatile("v_ashrrev_i32 %0, 11, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ int f_pkadd(int x, int b) { int r; asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(b)); return r; }
__device__ __forceinline__ int f_pkmad(int x, int a, int b) { int r; asm volatile("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int f_dpp(int x) { return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ int f_mul24v2(int x, int a) { int r; asm volatile("v_mul_i32_i24_e32 %0, %1, %2" : "=v"(r) : "s"(a), "v"(x)); return r; }
__device__ __forceinline__ int f_and(int x, int b) { int r; asm volatile("v_and_b32_e32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(b)); return r; }
__device__ __forceinline__ int f_max(int x, int b) { int r; asm volatile("v_max_i32_e32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(b)); return r; }
__device__ __forceinline__ int f_perm(int x, int b) { int r; asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(b), "s"(0x05010400)); return r; }
__device__ __forceinline__ int f_satpk(int x) { int r; asm volatile("v_sat_pk_u8_i16_e32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ int f_addsdwa(int x, int b) { int r; asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(r) : "v"(x), "v"(b)); return r; }
__device__ __forceinline__ int f_mulsdwa(int x, int b) { int r; asm volatile("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(x), "v"(b)); return r; }
__device__ __forceinline__ int f_cnd2(int x, int b) { int r; asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(r) : "v"(x), "v"(b) : ); return r; }
__device__ __forceinline__ int f_lshladd(int x, int b) { int r; asm volatile("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(r) : "v"(x), "v"(b)); return r; }
__device__ __forceinline__ int f_cnd64(int x, int b, unsigned long long m) { int r; asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(b), "s"(m)); return r; }
__device__ __forceinline__ int f_cndc(int x, int b, bool c) { return c ? x + 1 : b; }
#define G_CND64(x) f_cnd64(x, b, msk)
#define G_MUL24V2(x) f_mul24v2(x, a)
#define G_AND(x) f_and(x, b)
#define G_MAX(x) f_max(x, b)
#define G_PERM(x) f_perm(x, b)
#define G_SATPK(x) f_satpk(x)
#define G_ADDSDWA(x) f_addsdwa(x, b)
#define G_MULSDWA(x) f_mulsdwa(x, b)
#define G_CND2(x) f_cnd2(x, b)
#define G_LSHLADD(x) f_lshladd(x, b)
#define G_MAD(x) f_mad24(x, a, b)
#define G_MULLO(x) f_mullo(x, a)
#define G_BFE(x) f_bfe(x)
#define G_MED3(x) f_med3(x, b)
#define G_LSHLOR(x) f_lshlor(x, b)
#define G_ADD3(x) f_add3(x, a, b)
#define G_ADDV(x) f_addv(x, b)
#define G_CND(x) f_cnd(x, b)
#define G_ASHR(x) f_ashr(x)
#define G_PKADD(x) f_pkadd(x, b)
#define G_PKMAD(x) f_pkmad(x, a, b)
#define G_DPP(x) f_dpp(x)
OP_LOOP(k_addv, EACH(G_ADDV))
OP_LOOP(k_mad24, EACH(G_MAD))
OP_LOOP(k_mullo, EACH(G_MULLO))
OP_LOOP(k_bfe, EACH(G_BFE))
OP_LOOP(k_med3, EACH(G_MED3))
OP_LOOP(k_lshlor, EACH(G_LSHLOR))
OP_LOOP(k_add3, EACH(G_ADD3))
OP_LOOP(k_cnd, EACH(G_CND))
OP_LOOP(k_ashr, EACH(G_ASHR))
OP_LOOP(k_pkadd, EACH(G_PKADD))
OP_LOOP(k_pkmad, EACH(G_PKMAD))
OP_LOOP(k_dpp, EACH(G_DPP))
OP_LOOP(k_mul24v2, EACH(G_MUL24V2))
OP_LOOP(k_and, EACH(G_AND))
OP_LOOP(k_max, EACH(G_MAX))
OP_LOOP(k_perm, EACH(G_PERM))
OP_LOOP(k_satpk, EACH(G_SATPK))
OP_LOOP(k_addsdwa, EACH(G_ADDSDWA))
OP_LOOP(k_mulsdwa, EACH(G_MULSDWA))
OP_LOOP(k_cnd2, EACH(G_CND2))
OP_LOOP(k_lshladd, EACH(G_LSHLADD))
__global__ __launch_bounds__(256) void k_cnd64(int* out, int a, int b, int iters)
{
    unsigned long long msk = __builtin_amdgcn_ballot_w64((threadIdx.x & 1) != 0);
    int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
        _Pragma("unroll") for (int k = 0; k < 8; k++) { EACH(G_CND64) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__device__ __forceinline__ int f_dot2(int x, int a, int b) { int r; asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int f_dot4(int x, int a, int b) { int r; asm volatile("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int f_pkmul(int x, int a) { int r; asm volatile("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(a)); return r; }
__device__ __forceinline__ int f_madi16(int x, int a, int b) { int r; asm volatile("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int f_pkashr(int x, int a) { int r; asm volatile("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(x)); return r; }
#define G_DOT2(x) f_dot2(x, a, b)
#define G_DOT4(x) f_dot4(x, a, b)
#define G_PKMUL(x) f_pkmul(x, a)
#define G_MADI16(x) f_madi16(x, a, b)
#define G_PKASHR(x) f_pkashr(x, a)
OP_LOOP(k_dot2, EACH(G_DOT2))
OP_LOOP(k_dot4, EACH(G_DOT4))
OP_LOOP(k_pkmul, EACH(G_PKMUL))
OP_LOOP(k_madi16, EACH(G_MADI16))
OP_LOOP(k_pkashr, EACH(G_PKASHR))
__device__ __forceinline__ int f_mulhi(int x, int a) { int r; asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(a)); return r; }
__device__ __forceinline__ int f_mulhi24(int x, int a) { int r; asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(x), "v"(a)); return r; }
__device__ __forceinline__ int f_xor(int x, int a) { int r; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(a)); return r; }
#define G_MULHI(x) f_mulhi(x, a)
#define G_MULHI24(x) f_mulhi24(x, a)
#define G_XOR(x) f_xor(x, a)
OP_LOOP(k_mulhi, EACH(G_MULHI))
OP_LOOP(k_mulhi24, EACH(G_MULHI24))
OP_LOOP(k_xor, EACH(G_XOR))
typedef void (*kern_t)(int*, int, int, int);
int main()
{
    int* out;
    hipMalloc(&out, 256 * 8 * 256 * 8 * sizeof(int));
    struct { const char* name; kern_t k; } ks[] = {{"v_add_u32", k_addv}, {"v_mad_i32_i24", k_mad24}, {"v_mul_lo_u32", k_mullo}, {"v_bfe_i32", k_bfe},
        {"v_med3_i32", k_med3}, {"v_lshl_or_b32", k_lshlor}, {"v_add3_u32", k_add3}, {"v_cndmask_b32", k_cnd}, {"v_ashrrev_i32", k_ashr},
        {"v_pk_add_u16", k_pkadd}, {"v_pk_mad_u16", k_pkmad}, {"v_mov_dpp", k_dpp}, {"v_mul_i32_i24_e32", k_mul24v2}, {"v_and_b32", k_and},
        {"v_max_i32", k_max}, {"v_perm_b32", k_perm}, {"v_sat_pk_u8_i16", k_satpk}, {"v_add_u32_sdwa", k_addsdwa}, {"v_mul_u32_u24_sdwa", k_mulsdwa},
        {"v_cndmask_b32_e32", k_cnd2}, {"v_lshl_add_u32", k_lshladd}, {"v_cndmask_b32_e64 sgpr", k_cnd64}, {"v_dot2_i32_i16", k_dot2}, {"v_dot4_i32_i8", k_dot4}, {"v_pk_mul_lo_u16", k_pkmul}, {"v_mad_i32_i16", k_madi16}, {"v_pk_ashrrev_i16", k_pkashr}, {"v_mul_hi_u32", k_mulhi}, {"v_mul_hi_u32_u24", k_mulhi24}, {"v_xor_b32", k_xor}};
    const int iters = 2000;
    for (int wg_per_cu : {4}) {
        printf("== %d workgroups of 256 per CU (%d waves/SIMD)\n", wg_per_cu, wg_per_cu);
        for (auto& e : ks) {
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            int grid = 256 * wg_per_cu;
            hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, out, 3, 5, 10);
            hipEventRecord(a);
            hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, out, 3, 5, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            double winstr = (double)grid * 4 * iters * 64;          // wave-instructions
            double per_simd_per_us = winstr / 1024 / (ms * 1e3);
            printf("  %-16s %.3f ms  %.1f wave-instr/us/SIMD  -> %.2f cycles/instr at 2.4 GHz\n", e.name, ms, per_simd_per_us, 2400.0 / per_simd_per_us);
        }
    }
    return 0;
}
