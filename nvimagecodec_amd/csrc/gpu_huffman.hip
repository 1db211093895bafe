// gpu_huffman.hip -- gfx950 kernels of the GPU entropy decoder (algorithm and data structures: huffman_gpu_core.h).
//
// This is the analogue of nvJPEG's GPU_HYBRID backend that the reference switches to for large baseline images
// (extensions/nvjpeg/cuda_decoder.cpp:512-521): the Huffman stage moves to the device, so neither the host cores nor the
// PCIe copy of 6 MB of coefficients per image sit on the critical path any more -- only the ~0.5 MB bitstream crosses.
//
// Mapping: one lane decodes one 1024-bit subsequence; a workgroup owns 256 consecutive subsequences of ONE image and keeps
// that image's Huffman tables in LDS (8 slots x 2.4 KB).  Decoding is inherently serial per lane (data-dependent code
// lengths); parallelism comes from the number of subsequences (4096 per 0.5 MB image, ~1M per 256-image batch).
//   huff_sync_kernel   pass 0 + workgroup-local synchronisation loop in LDS; publishes end states; counts global changes
//   huff_scan_kernel   per image: exclusive scan of completed-block counts -> first block index of every subsequence
//   huff_write_kernel  final decode with coefficient stores
//   huff_dc_kernel     per (image, component): DC differences -> DC values in MCU order
#include <hip/hip_runtime.h>

#include "gpu_huffman.h"
#include "huffman_gpu_core.h"

namespace hipjpeg {

namespace {

constexpr int kThreads = 256;

struct LdsTables {
    HuffDecodeTable t[8];
    __device__ const HuffDecodeTable& operator[](int i) const { return t[i]; }
};

// cooperative copy of the image's tables into LDS (dword granularity; sizeof(HuffDecodeTable) is a multiple of 4)
__device__ __forceinline__ void load_tables(LdsTables* dst, const HuffDecodeTable* src)
{
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    constexpr int n = (int)(sizeof(LdsTables) / 4);
    for (int i = threadIdx.x; i < n; i += kThreads) d[i] = s[i];
}

__device__ __forceinline__ unsigned long long pack_state(const SubseqState& s)
{
    return ((unsigned long long)s.end_bit) | ((unsigned long long)s.zk << 32) | ((unsigned long long)s.nblocks << 48);
}
__device__ __forceinline__ SubseqState unpack_state(unsigned long long v)
{
    SubseqState s;
    s.end_bit = (uint32_t)v;
    s.zk = (uint16_t)(v >> 32);
    s.nblocks = (uint16_t)(v >> 48);
    return s;
}

// states[]: one 8-byte record per subsequence (batch-wide indexing through HuffImage::first_subseq).
__global__ __launch_bounds__(kThreads) void huff_sync_kernel(const HuffImage* __restrict__ images, const HuffUnit* __restrict__ units,
                                                             unsigned long long* __restrict__ states, unsigned int* __restrict__ changed,
                                                             int first_pass)
{
    __shared__ LdsTables tables;
    __shared__ unsigned long long s_end[kThreads + 1];  // [0] = state entering the workgroup, [t+1] = end state of lane t
    const HuffUnit u = units[blockIdx.x];
    const HuffImage& im = images[u.image];
    load_tables(&tables, im.tables);
    const int t = threadIdx.x;
    const uint32_t j = u.first + t;  // subsequence index inside the image
    const bool active = j < im.num_subseq;
    unsigned long long* gstate = states + im.first_subseq;
    const unsigned long long kInitial = 0;  // end_bit 0, z 0, k 0: the exact state at the start of the scan

    unsigned long long old_global = 0, mine = 0;
    if (t == 0) s_end[0] = u.first == 0 ? kInitial : __hip_atomic_load(&gstate[u.first - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    uint32_t err = 0;
    // the start state this lane decoded from last time (bit position + zk); ~0 = never
    unsigned long long last_start = ~0ull;
    if (active) {
        if (first_pass) {
            // pass 0: from the first bit of the own subsequence, as if a block started there
            SubseqState e = decode_subsequence<false>(im, tables, j * kSubseqBits, (j + 1) * kSubseqBits, 0, 0, 0, &err);
            mine = pack_state(e);
            old_global = ~0ull;
        } else {
            old_global = gstate[j];
            mine = old_global;
        }
    }
    s_end[t + 1] = mine;
    // workgroup-local fixpoint: lane t re-decodes whenever the end state of lane t-1 (or the incoming state) is not the
    // one it started from last time.  Corrections travel one lane per round; rounds stop when nothing moved.
    for (int round = 0; round < kThreads + 1; round++) {
        __syncthreads();
        const unsigned long long prev = s_end[t] & 0x0000FFFFFFFFFFFFull;  // start state = predecessor's end (without its block count)
        bool moved = false;
        if (active && prev != last_start && !(j == 0)) {
            const SubseqState p = unpack_state(prev);
            SubseqState e = decode_subsequence<false>(im, tables, p.end_bit, (j + 1) * kSubseqBits, p.zk & 255, p.zk >> 8, 0, &err);
            const unsigned long long now = pack_state(e);
            moved = now != mine;
            mine = now;
            last_start = prev;
        } else if (active && j == 0 && last_start == ~0ull) {
            // subsequence 0 of the image starts in the exact state; pass 0 already decoded it from there
            last_start = prev;
        }
        __syncthreads();
        s_end[t + 1] = mine;
        if (!__syncthreads_or(moved ? 1 : 0)) break;
    }
    if (active) {
        if (mine != old_global) {
            __hip_atomic_store(&gstate[j], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(changed, 1u);
        }
    }
}

// One workgroup per image: first_block[j] = sum of nblocks of subsequences before j.
__global__ __launch_bounds__(kThreads) void huff_scan_kernel(HuffImage* __restrict__ images, const uint32_t* __restrict__ image_list,
                                                             const unsigned long long* __restrict__ states, uint32_t* __restrict__ first_block)
{
    __shared__ uint32_t s_sum[kThreads];
    HuffImage& im = images[image_list[blockIdx.x]];
    const unsigned long long* st = states + im.first_subseq;
    uint32_t* fb = first_block + im.first_subseq;
    const uint32_t n = im.num_subseq;
    const uint32_t per = (n + kThreads - 1) / kThreads;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += (uint32_t)(st[i] >> 48);
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over 256 partial sums
    for (int off = 1; off < kThreads; off <<= 1) {
        uint32_t v = threadIdx.x >= (unsigned)off ? s_sum[threadIdx.x - off] : 0;
        __syncthreads();
        s_sum[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? s_sum[threadIdx.x - 1] : 0;
    for (uint32_t i = lo; i < hi; i++) {
        fb[i] = run;
        run += (uint32_t)(st[i] >> 48);
    }
    if (threadIdx.x == kThreads - 1 && s_sum[kThreads - 1] < im.total_blocks) im.status = 2;  // the stream ends before the last block
}

__global__ __launch_bounds__(kThreads) void huff_write_kernel(HuffImage* __restrict__ images, const HuffUnit* __restrict__ units,
                                                              const unsigned long long* __restrict__ states, const uint32_t* __restrict__ first_block)
{
    __shared__ LdsTables tables;
    const HuffUnit u = units[blockIdx.x];
    HuffImage& im = images[u.image];
    load_tables(&tables, im.tables);
    __syncthreads();
    const uint32_t j = u.first + threadIdx.x;
    if (j >= im.num_subseq) return;
    const unsigned long long* st = states + im.first_subseq;
    uint32_t begin = 0;
    int z = 0, k = 0;
    if (j > 0) {
        const SubseqState p = unpack_state(st[j - 1]);
        begin = p.end_bit;
        z = p.zk & 255;
        k = p.zk >> 8;
    }
    uint32_t err = 0;
    decode_subsequence<true>(im, tables, begin, (j + 1) * kSubseqBits, z, k, first_block[im.first_subseq + j], &err);
    if (err) im.status = 1;  // benign race: every writer stores the same value
}

// One workgroup per (image, component): integrate the DC differences in MCU order.
__global__ __launch_bounds__(kThreads) void huff_dc_kernel(const HuffImage* __restrict__ images, const HuffUnit* __restrict__ units)
{
    __shared__ int s_sum[kThreads];
    const HuffUnit u = units[blockIdx.x];  // first = component
    const HuffImage& im = images[u.image];
    const int c = (int)u.first;
    const uint32_t h = im.comp_h[c], v = im.comp_v[c], bpc = h * v;  // blocks of this component per MCU
    const uint32_t mcus = im.total_blocks / im.blocks_per_mcu;
    const uint32_t n = mcus * bpc;
    int16_t* coef = im.coef[c];
    const uint32_t bw = im.blocks_w[c], mcus_x = im.mcus_x;
    auto dc_ptr = [&](uint32_t s) -> int16_t* {
        const uint32_t mcu = s / bpc, jj = s - mcu * bpc;
        const uint32_t my = mcu / mcus_x, mx = mcu - my * mcus_x;
        const uint32_t dy = jj / h, dx = jj - dy * h;
        return coef + ((size_t)(my * v + dy) * bw + (mx * h + dx)) * 64;
    };
    const uint32_t per = (n + kThreads - 1) / kThreads;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    int sum = 0;
    for (uint32_t s = lo; s < hi; s++) sum += *dc_ptr(s);
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < kThreads; off <<= 1) {
        int t = threadIdx.x >= (unsigned)off ? s_sum[threadIdx.x - off] : 0;
        __syncthreads();
        s_sum[threadIdx.x] += t;
        __syncthreads();
    }
    int run = threadIdx.x ? s_sum[threadIdx.x - 1] : 0;
    for (uint32_t s = lo; s < hi; s++) {
        int16_t* p = dc_ptr(s);
        run += *p;
        *p = (int16_t)run;
    }
}

}  // namespace

int launch_huff_sync(const HuffImage* images, const HuffUnit* units, int nunits, unsigned long long* states, unsigned int* changed, int first_pass,
                     void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(huff_sync_kernel, dim3(nunits), dim3(kThreads), 0, (hipStream_t)stream, images, units, states, changed, first_pass);
    return (int)hipGetLastError();
}

int launch_huff_scan(HuffImage* images, const uint32_t* image_list, int nimages, const unsigned long long* states, uint32_t* first_block, void* stream)
{
    if (nimages <= 0) return 0;
    hipLaunchKernelGGL(huff_scan_kernel, dim3(nimages), dim3(kThreads), 0, (hipStream_t)stream, images, image_list, states, first_block);
    return (int)hipGetLastError();
}

int launch_huff_write(HuffImage* images, const HuffUnit* units, int nunits, const unsigned long long* states, const uint32_t* first_block, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(huff_write_kernel, dim3(nunits), dim3(kThreads), 0, (hipStream_t)stream, images, units, states, first_block);
    return (int)hipGetLastError();
}

int launch_huff_dc(const HuffImage* images, const HuffUnit* units, int nunits, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(huff_dc_kernel, dim3(nunits), dim3(kThreads), 0, (hipStream_t)stream, images, units);
    return (int)hipGetLastError();
}

}  // namespace hipjpeg
