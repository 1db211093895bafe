// Dev tool: what a lone wave pays for the building blocks of the progressive walker's "scalar machine" on gfx950:
//   dependent SALU chain, v_readlane -> SALU -> v_readlane chain, taken scalar branches, uniform LDS lookups.
// build: hipcc --offload-arch=gfx950 -O3 tools/scalar_chain_rate.hip -o tools/scalar_chain_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define N 16384

__global__ void k_salu(unsigned* out, unsigned seed)
{
    unsigned x = __builtin_amdgcn_readfirstlane(seed);
    unsigned long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; i++) {
        asm volatile("s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 5\n s_add_u32 %0, %0, 3\n s_xor_b32 %0, %0, 9\n"
                     "s_add_u32 %0, %0, 1\n s_xor_b32 %0, %0, 5\n s_add_u32 %0, %0, 3\n s_xor_b32 %0, %0, 9\n" : "+s"(x) : : "scc");
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = x; out[1] = (unsigned)(t1 - t0); }
}

__global__ void k_readlane(unsigned* out, unsigned seed)
{
    unsigned v = threadIdx.x * 7 + seed;
    unsigned x = __builtin_amdgcn_readfirstlane(seed) & 63;
    unsigned long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            unsigned y = __builtin_amdgcn_readlane(v, x);
            x = (y >> 3) & 63;   // SALU between two readlanes
        }
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = x; out[1] = (unsigned)(t1 - t0); }
}

__global__ void k_branch(unsigned* out, unsigned seed)
{
    unsigned x = __builtin_amdgcn_readfirstlane(seed);
    unsigned long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; i++) {
        // 8 taken forward branches over one instruction each
        asm volatile("s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n1:\n"
                     "s_cmp_eq_u32 0, 0\n s_cbranch_scc1 2f\n s_add_u32 %0, %0, 1\n2:\n"
                     "s_cmp_eq_u32 0, 0\n s_cbranch_scc1 3f\n s_add_u32 %0, %0, 1\n3:\n"
                     "s_cmp_eq_u32 0, 0\n s_cbranch_scc1 4f\n s_add_u32 %0, %0, 1\n4:\n"
                     "s_cmp_eq_u32 0, 0\n s_cbranch_scc1 5f\n s_add_u32 %0, %0, 1\n5:\n"
                     "s_cmp_eq_u32 0, 0\n s_cbranch_scc1 6f\n s_add_u32 %0, %0, 1\n6:\n"
                     "s_cmp_eq_u32 0, 0\n s_cbranch_scc1 7f\n s_add_u32 %0, %0, 1\n7:\n"
                     "s_cmp_eq_u32 0, 0\n s_cbranch_scc1 8f\n s_add_u32 %0, %0, 1\n8:\n" : "+s"(x) : : "scc");
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = x; out[1] = (unsigned)(t1 - t0); }
}

__global__ void k_nottaken(unsigned* out, unsigned seed)
{
    unsigned x = __builtin_amdgcn_readfirstlane(seed);
    unsigned long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; i++) {
        asm volatile("s_cmp_eq_u32 0, 1\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n1:\n"
                     "s_cmp_eq_u32 0, 1\n s_cbranch_scc1 2f\n s_add_u32 %0, %0, 1\n2:\n"
                     "s_cmp_eq_u32 0, 1\n s_cbranch_scc1 3f\n s_add_u32 %0, %0, 1\n3:\n"
                     "s_cmp_eq_u32 0, 1\n s_cbranch_scc1 4f\n s_add_u32 %0, %0, 1\n4:\n"
                     "s_cmp_eq_u32 0, 1\n s_cbranch_scc1 5f\n s_add_u32 %0, %0, 1\n5:\n"
                     "s_cmp_eq_u32 0, 1\n s_cbranch_scc1 6f\n s_add_u32 %0, %0, 1\n6:\n"
                     "s_cmp_eq_u32 0, 1\n s_cbranch_scc1 7f\n s_add_u32 %0, %0, 1\n7:\n"
                     "s_cmp_eq_u32 0, 1\n s_cbranch_scc1 8f\n s_add_u32 %0, %0, 1\n8:\n" : "+s"(x) : : "scc");
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = x; out[1] = (unsigned)(t1 - t0); }
}

__global__ void k_lds(unsigned* out, unsigned seed)
{
    __shared__ unsigned short tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = (unsigned short)(i * 37 + seed);
    __syncthreads();
    unsigned x = __builtin_amdgcn_readfirstlane(seed) & 1023;
    unsigned long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) x = __builtin_amdgcn_readfirstlane(tab[x]) & 1023;  // uniform LDS read -> SGPR -> next address
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = x; out[1] = (unsigned)(t1 - t0); }
}

__global__ void k_valu(unsigned* out, unsigned seed)
{
    unsigned x = threadIdx.x + seed;
    unsigned long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N; i++) {
        asm volatile("v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 5\n v_add_u32 %0, %0, 3\n v_xor_b32 %0, %0, 9\n"
                     "v_add_u32 %0, %0, 1\n v_xor_b32 %0, %0, 5\n v_add_u32 %0, %0, 3\n v_xor_b32 %0, %0, 9\n" : "+v"(x));
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = x; out[1] = (unsigned)(t1 - t0); }
}

int main()
{
    unsigned* d;
    hipMalloc(&d, 64);
    unsigned h[2];
    struct { const char* name; void (*k)(unsigned*, unsigned); } ks[] = {{"dependent SALU (8 per iteration)", k_salu}, {"dependent VALU (8)", k_valu},
        {"readlane -> 2 SALU -> readlane (8)", k_readlane}, {"taken s_cbranch (8) + s_cmp", k_branch}, {"not-taken s_cbranch (8) + s_cmp + s_add", k_nottaken},
        {"uniform LDS read -> readfirstlane -> address (8)", k_lds}};
    for (auto& e : ks) {
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(e.k, dim3(1), dim3(64), 0, 0, d, 12345u);
            hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        }
        printf("%-52s %8.1f clock64 ticks per element (loop overhead included)\n", e.name, h[1] / (double)(N * 8));
    }
    // The same chains with MANY waves resident: what a compute unit's one scalar unit delivers when the progressive walker's waves of
    // several batches share it.  One workgroup = one wave; `waves per CU` x 256 workgroups (the dispatcher spreads them evenly).
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct { const char* name; void (*k)(unsigned*, unsigned); int per_iter; } ms[] = {{"dependent SALU", k_salu, 8}, {"readlane -> 2 SALU -> readlane", k_readlane, 24},
        {"dependent VALU", k_valu, 8}};
    for (auto& e : ms)
        for (int waves : {1, 2, 4, 8, 16, 32}) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(e.k, dim3(256 * waves), dim3(64), 0, 0, d, 12345u);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float t;
                hipEventElapsedTime(&t, e0, e1);
                best = t < best ? t : best;
            }
            const double instr = (double)N * e.per_iter * 256.0 * waves;
            printf("%-34s %2d waves per CU: %7.3f ms, %6.2f G instructions/s per CU = %.2f per cycle per CU at 2.4 GHz\n", e.name, waves, best,
                   instr / (best * 1e-3) / 256.0 / 1e9, instr / (best * 1e-3) / 256.0 / 2.4e9);
        }
    return 0;
}
