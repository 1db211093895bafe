// gpu_huffman.hip -- gfx950 kernels of the GPU entropy decoder (algorithm and data structures: huffman_gpu_core.h).
//
// This is the analogue of nvJPEG's GPU_HYBRID backend that the reference switches to for large baseline images
// (extensions/nvjpeg/cuda_decoder.cpp:512-521): the Huffman stage moves to the device, so neither the host cores nor the
// PCIe copy of 6 MB of coefficients per image sit on the critical path any more -- only the ~0.5 MB bitstream crosses.
//
// Mapping: one lane decodes one 1024-bit subsequence; a workgroup owns 256 consecutive subsequences of ONE image.
// Decoding is inherently serial per lane (data-dependent code lengths); parallelism comes from the number of subsequences
// (4096 per 0.5 MB image, ~1M per 256-image batch).  Everything a lane touches per symbol lives in LDS:
//   * the workgroup's 32 KB of bitstream, loaded once with coalesced dword loads, byte-swapped to MSB-first words and stored
//     one row of 33 words per lane (32 own words + a copy of the successor's first word): the odd row stride spreads lanes
//     that sit at the same column over different banks, and a lane's two-word window never leaves its row;
//   * the image's two-level Huffman lookup tables (uint16 entries, typically ~10 KB);
//   * per-MCU-position table offsets and block addressing constants, the zigzag permutation.
// Kernels:
//   huff_sync_kernel   pass 0 + workgroup-local synchronisation loop in LDS; publishes end states; counts the workgroups
//                      whose LAST end state moved (the only state another workgroup consumes)
//   huff_scan_kernel   per image: exclusive scan of completed-block counts -> first block index of every subsequence
//   huff_write_kernel  final decode with coefficient stores (DC differences to a compact array)
//   huff_dc_kernel     per (image, component): DC differences -> DC values, stored into the blocks
#include <hip/hip_runtime.h>

#include "gpu_huffman.h"
#include "huffman_gpu_core.h"

namespace hipjpeg {

namespace {

constexpr int kThreads = 256;
constexpr int kRowWords = kSubseqWords + 1;

#define HJ_LDS __attribute__((address_space(3)))

struct KSlot {
    int16_t* base;  // coef[comp] + blk0 * 64
    uint32_t stride_y, stride_x;
};

struct WgShared {
    uint32_t stream[kThreads * kRowWords];
    unsigned long long end[kThreads + 1];  // [0] = state entering the workgroup, [t+1] = end state of lane t
    KSlot kslot[10];
    uint32_t tsel[10];
    uint32_t zz[16];  // zigzag permutation, 4 entries per word
};

// LDS accessors for decode_subsequence; one instance per lane.
struct DevEnv {
    const HJ_LDS uint32_t* row;     // this lane's stream row
    const HJ_LDS uint16_t* pool;
    const HJ_LDS uint32_t* tsel;
    const HJ_LDS KSlot* kslot;
    const HJ_LDS uint8_t* zz;
    uint32_t row_bit0;              // bit position of the row's first bit
    __device__ __forceinline__ uint32_t window(uint32_t pos) const
    {
        const uint32_t q = pos - row_bit0;
        const HJ_LDS uint32_t* p = row + (q >> 5);
        const unsigned long long two = ((unsigned long long)p[0] << 32) | p[1];
        return (uint32_t)(two >> (32 - (q & 31)));
    }
    __device__ __forceinline__ uint32_t entry(uint32_t i) const { return pool[i]; }
    __device__ __forceinline__ uint32_t tables(int k) const { return tsel[k]; }
    __device__ __forceinline__ int16_t* block_ptr(int k, uint32_t mx, uint32_t my) const
    {
        const HJ_LDS KSlot* s = kslot + k;
        int16_t* base = s->base;
        const uint32_t sy = s->stride_y, sx = s->stride_x;
        return base + (size_t)(my * sy + mx * sx) * 64;
    }
    __device__ __forceinline__ int zigzag(int z) const { return zz[z]; }
};

// Cooperative staging of the workgroup's slice of the image: stream rows, lookup tables, per-position constants.
__device__ __forceinline__ void stage_workgroup(WgShared& sh, HJ_LDS uint16_t* pool, const HuffImage& im, uint32_t first_subseq, bool with_addresses)
{
    const int t = threadIdx.x;
    const uint32_t* g = reinterpret_cast<const uint32_t*>(im.stream);
    const uint32_t nwords = im.stream_words;
    const uint32_t w0 = first_subseq * kSubseqWords;
    for (uint32_t d = t; d <= (uint32_t)kThreads * kSubseqWords; d += kThreads) {
        const uint32_t gd = w0 + d;
        const uint32_t w = gd < nwords ? __builtin_bswap32(g[gd]) : 0xFFFFFFFFu;
        const uint32_t row = d >> 5, col = d & 31;
        if (row < (uint32_t)kThreads) sh.stream[row * kRowWords + col] = w;
        if (col == 0 && row > 0) sh.stream[(row - 1) * kRowWords + kSubseqWords] = w;
    }
    const uint32_t* gp = reinterpret_cast<const uint32_t*>(im.pool);
    HJ_LDS uint32_t* lp = reinterpret_cast<HJ_LDS uint32_t*>(pool);
    const uint32_t npool = im.pool_words >> 1;  // pool_words is a multiple of 64
    for (uint32_t i = t; i < npool; i += kThreads) lp[i] = gp[i];
    if (t < 10) {
        const HuffK hk = im.k[t];
        sh.tsel[t] = (uint32_t)hk.tdc | ((uint32_t)hk.tac << 16);
        if (with_addresses) {
            KSlot s;
            s.base = im.coef[hk.comp & 3] + (size_t)hk.blk0 * 64;
            s.stride_y = hk.stride_y;
            s.stride_x = hk.stride_x;
            sh.kslot[t] = s;
        }
    }
    if (t >= 64 && t < 80) {
        constexpr uint8_t zz[64] = HJ_ZIGZAG_DEVICE_TABLE;
        const int i = (t - 64) * 4;
        sh.zz[t - 64] = (uint32_t)zz[i] | ((uint32_t)zz[i + 1] << 8) | ((uint32_t)zz[i + 2] << 16) | ((uint32_t)zz[i + 3] << 24);
    }
}

__device__ __forceinline__ DevEnv make_env(WgShared& sh, HJ_LDS uint16_t* pool, uint32_t j)
{
    DevEnv env;
    env.row = (const HJ_LDS uint32_t*)&sh.stream[threadIdx.x * kRowWords];
    env.pool = pool;
    env.tsel = (const HJ_LDS uint32_t*)sh.tsel;
    env.kslot = (const HJ_LDS KSlot*)sh.kslot;
    env.zz = (const HJ_LDS uint8_t*)sh.zz;
    env.row_bit0 = j * kSubseqBits;
    return env;
}

// ---- byte-stuffing removal ------------------------------------------------------------------------------------------
// The file's entropy-coded segment escapes every 0xFF data byte as FF 00.  Two kernels turn it into the plain bitstream the
// decoders read: the first counts the stuffed bytes of every 16 KB chunk, the second compacts each chunk to its final
// position (chunk offset minus the stuffed bytes before it) through LDS.  Streams with anything else behind an FF (restart
// markers, fill bytes) never get here (gpu_entropy_eligible()).

// 0x80 in every byte of x (little-endian dword) that is 0x00 and whose predecessor byte is 0xFF; prev = the byte before x.
__device__ __forceinline__ uint32_t stuffed_mask(uint32_t x, uint32_t prev)
{
    const uint32_t np = ~((x << 8) | prev);  // byte i = ~predecessor of x's byte i: zero where the predecessor is 0xFF
    const uint32_t zero_x = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
    const uint32_t zero_np = ~(((np & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | np | 0x7F7F7F7Fu);
    return zero_x & zero_np;
}

// stuffed-byte masks of the 64 bytes at p (16-byte aligned); returns the count
__device__ __forceinline__ uint32_t lane_masks(const uint8_t* p, bool has_prev, uint32_t m[16], uint32_t x[16])
{
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint32_t prev = has_prev ? p[-1] : 0u;
    uint32_t n = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint4 v = q[i];
        x[4 * i + 0] = v.x;
        x[4 * i + 1] = v.y;
        x[4 * i + 2] = v.z;
        x[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        m[i] = stuffed_mask(x[i], prev);
        prev = x[i] >> 24;
        n += __popc(m[i]);
    }
    return n;
}

__device__ __forceinline__ uint32_t wg_sum(uint32_t v, uint32_t* scratch /*[4]*/)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    const uint32_t total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    __syncthreads();
    return total;
}

// unit.first = chunk index inside the image; drops[im.first_chunk + chunk] = stuffed bytes in the chunk
__global__ __launch_bounds__(kThreads) void destuff_count_kernel(const HuffImage* __restrict__ images, const HuffUnit* __restrict__ units,
                                                                 uint32_t* __restrict__ drops)
{
    __shared__ uint32_t scratch[4];
    const HuffUnit u = units[blockIdx.x];
    const HuffImage& im = images[u.image];
    const uint32_t off = u.first * kDestuffChunk + threadIdx.x * 64;
    uint32_t n = 0;
    if (off < im.raw_bytes) {  // the raw copy is padded to 16 bytes with a neutral value; whole 16-byte pieces are readable
        uint32_t m[16], x[16];
        const uint32_t pieces = min(4u, (im.raw_bytes - off + 15) / 16);
        if (pieces == 4) {
            n = lane_masks(im.raw + off, off > 0, m, x);
        } else {
            uint32_t prev = off > 0 ? im.raw[off - 1] : 0u;
            for (uint32_t i = 0; i < pieces * 4; i++) {
                const uint32_t w = reinterpret_cast<const uint32_t*>(im.raw + off)[i];
                n += __popc(stuffed_mask(w, prev));
                prev = w >> 24;
            }
        }
    }
    const uint32_t total = wg_sum(n, scratch);
    if (threadIdx.x == 0) drops[im.first_chunk + u.first] = total;
}

__global__ __launch_bounds__(kThreads) void destuff_compact_kernel(HuffImage* __restrict__ images, const HuffUnit* __restrict__ units,
                                                                   const uint32_t* __restrict__ drops)
{
    __shared__ uint32_t scratch[4];
    __shared__ uint32_t wave_base[4];
    __shared__ uint32_t out_words[kDestuffChunk / 4 + 2];
    HJ_LDS uint8_t* out = (HJ_LDS uint8_t*)out_words;
    const HuffUnit u = units[blockIdx.x];
    HuffImage& im = images[u.image];
    const int t = threadIdx.x;
    const uint32_t raw_bytes = im.raw_bytes;
    // stuffed bytes in the chunks before this one
    uint32_t before = 0;
    for (uint32_t c = t; c < u.first; c += kThreads) before += drops[im.first_chunk + c];
    before = wg_sum(before, scratch);
    const uint32_t chunk_begin = u.first * kDestuffChunk;
    const uint32_t chunk_len = min((uint32_t)kDestuffChunk, raw_bytes - chunk_begin);
    const uint32_t gout = chunk_begin - before;  // destination offset of the chunk's first kept byte
    const uint32_t a = gout & 3;                 // the LDS image shares the destination's misalignment

    const uint32_t off = chunk_begin + t * 64;
    uint32_t m[16], x[16];
    uint32_t n = 0, len = 0;
    if (off < raw_bytes) {
        len = min(64u, raw_bytes - off);
        const uint32_t pieces = (len + 15) / 16;
        uint32_t prev = off > 0 ? im.raw[off - 1] : 0u;
        for (uint32_t i = 0; i < 16; i++) {
            x[i] = i < pieces * 4 ? reinterpret_cast<const uint32_t*>(im.raw + off)[i] : 0x01010101u;
            m[i] = stuffed_mask(x[i], prev);
            prev = x[i] >> 24;
            n += __popc(m[i]);
        }
    }
    // exclusive scan of the per-lane counts: inside the wave by shuffles, across the four waves through LDS
    uint32_t incl = n;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d);
        if ((t & 63) >= d) incl += v;
    }
    if ((t & 63) == 63) wave_base[t >> 6] = incl;
    __syncthreads();
    uint32_t excl = incl - n;
    for (int w = 0; w < (t >> 6); w++) excl += wave_base[w];
    const uint32_t chunk_drops = wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3];
    // kept bytes of this lane -> LDS
    if (len) {
        uint32_t o = a + t * 64 - excl;
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t w = x[i], mask = m[i];
            if (i * 4 + 4 <= len && mask == 0) {
                out[o] = (uint8_t)w;
                out[o + 1] = (uint8_t)(w >> 8);
                out[o + 2] = (uint8_t)(w >> 16);
                out[o + 3] = (uint8_t)(w >> 24);
                o += 4;
            } else {
                for (uint32_t b = 0; b < 4; b++)
                    if (i * 4 + b < len && !((mask >> (8 * b + 7)) & 1)) out[o++] = (uint8_t)(w >> (8 * b));
            }
        }
    }
    __syncthreads();
    // LDS -> destination: whole dwords in the middle, single bytes at the ragged ends (neighbouring chunks share those dwords)
    const uint32_t n_out = chunk_len - chunk_drops;
    uint8_t* dst = const_cast<uint8_t*>(im.stream) + (gout - a);  // 4-byte aligned
    const uint32_t first_full = a ? 1 : 0, end_full = (a + n_out) / 4;
    for (uint32_t d = first_full + t; d < end_full; d += kThreads) reinterpret_cast<uint32_t*>(dst)[d] = out_words[d];
    if (t < 4) {
        if (a && (uint32_t)t >= a && (uint32_t)t < a + n_out) dst[t] = out[t];  // head
        const uint32_t tail = max(end_full, first_full) * 4 + t;
        if (tail >= a && tail < a + n_out && tail >= 4 * first_full) dst[tail] = out[tail];
    }
    // the last chunk knows the destuffed length: publish it and lay down the 0xFF slack behind the data
    if (chunk_begin + chunk_len == raw_bytes) {
        const uint32_t total = raw_bytes - before - chunk_drops;
        const uint32_t padded = ((total + 3) & ~3u) + kStreamSlackBytes;
        uint8_t* s = const_cast<uint8_t*>(im.stream);
        if (total + t < padded) s[total + t] = 0xFF;
        if (t == 0) {
            im.total_bits = total * 8;
            im.num_subseq = (total * 8 + kSubseqBits - 1) / kSubseqBits;
            im.stream_words = padded / 4;
        }
    }
}

// states[]: one 8-byte record per subsequence (batch-wide indexing through HuffImage::first_subseq).
// changed[0] += 1 for every workgroup whose outgoing state (end state of its last subsequence) differs from the published one.
__global__ __launch_bounds__(kThreads) void huff_sync_kernel(const HuffImage* __restrict__ images, const HuffUnit* __restrict__ units,
                                                             unsigned long long* __restrict__ states, unsigned int* __restrict__ changed,
                                                             int first_pass)
{
    __shared__ WgShared sh;
    extern __shared__ uint16_t dyn_pool[];
    HJ_LDS uint16_t* pool = (HJ_LDS uint16_t*)dyn_pool;
    const HuffUnit u = units[blockIdx.x];
    const HuffImage& im = images[u.image];
    const HuffGeom geom = make_geom(im);
    const uint32_t nsub = (geom.total_bits + kSubseqBits - 1) / kSubseqBits;
    if (u.first >= nsub) return;  // uniform: the stream turned out shorter than planned
    stage_workgroup(sh, pool, im, u.first, false);
    const int t = threadIdx.x;
    const uint32_t j = u.first + t;  // subsequence index inside the image
    const bool active = j < nsub;
    const bool last = j == min(nsub, u.first + kThreads) - 1;
    unsigned long long* gstate = states + im.first_subseq;
    const DevEnv env = make_env(sh, pool, j);

    // the state a lane assumes in pass 0: a block of the first MCU position starts exactly at the subsequence boundary
    const unsigned long long assumed = (unsigned long long)j * kSubseqBits;
    unsigned long long old_global = 0, mine = 0, last_start = assumed;
    if (t == 0) sh.end[0] = (first_pass || u.first == 0) ? assumed : __hip_atomic_load(&gstate[u.first - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kSyncMask;
    __syncthreads();
    uint32_t err = 0;
    if (active) {
        if (first_pass) {
            mine = pack_state(decode_subsequence<false>(geom, env, j * kSubseqBits, (j + 1) * kSubseqBits, 0, 0, nullptr, &err));
            old_global = ~0ull;
        } else {
            old_global = gstate[j];
            mine = old_global;
        }
    }
    sh.end[t + 1] = mine;
    if (!first_pass) {
        // the published states of a workgroup are consistent among themselves (the previous launch ended in a local
        // fixpoint): only lane 0 has to look at its -- possibly new -- incoming state
        __syncthreads();
        last_start = t == 0 ? (u.first == 0 ? assumed : ~0ull) : (sh.end[t] & kSyncMask);
    }
    // workgroup-local fixpoint: lane t re-decodes whenever the end state of lane t-1 (or the incoming state) is not the
    // one it started from last time.  Corrections travel one lane per round; rounds stop when nothing moved.
    for (int round = 0; round < kThreads + 1; round++) {
        __syncthreads();
        const unsigned long long prev = sh.end[t] & kSyncMask;
        bool moved = false;
        if (active && prev != last_start) {
            const SubseqState p = unpack_state(prev);
            const unsigned long long now =
                pack_state(decode_subsequence<false>(geom, env, p.end_bit, (j + 1) * kSubseqBits, p.zk & 255, p.zk >> 8, nullptr, &err));
            moved = ((now ^ mine) & kSyncMask) != 0;
            mine = now;
            last_start = prev;
        }
        __syncthreads();
        sh.end[t + 1] = mine;
        if (!__syncthreads_or(moved ? 1 : 0)) break;
    }
    if (active && mine != old_global) {
        __hip_atomic_store(&gstate[j], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (last && ((mine ^ old_global) & kSyncMask) != 0) atomicAdd(changed, 1u);
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
    const uint32_t n = (im.total_bits + kSubseqBits - 1) / kSubseqBits;
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
    __shared__ WgShared sh;
    extern __shared__ uint16_t dyn_pool[];
    HJ_LDS uint16_t* pool = (HJ_LDS uint16_t*)dyn_pool;
    const HuffUnit u = units[blockIdx.x];
    HuffImage& im = images[u.image];
    const HuffGeom geom = make_geom(im);
    const uint32_t nsub = (geom.total_bits + kSubseqBits - 1) / kSubseqBits;
    if (u.first >= nsub) return;
    stage_workgroup(sh, pool, im, u.first, true);
    __syncthreads();
    const uint32_t j = u.first + threadIdx.x;
    if (j >= nsub) return;
    const DevEnv env = make_env(sh, pool, j);
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
    HuffCursor cursor = make_cursor(geom, env, first_block[im.first_subseq + j], k);
    decode_subsequence<true>(geom, env, begin, (j + 1) * kSubseqBits, z, k, &cursor, &err);
    if (err) im.status = 1;  // benign race: every writer stores the same value
}

// One workgroup per (image, component): integrate the DC differences in MCU order and store them into the blocks.
__global__ __launch_bounds__(kThreads) void huff_dc_kernel(const HuffImage* __restrict__ images, const HuffUnit* __restrict__ units)
{
    __shared__ int s_sum[kThreads];
    const HuffUnit u = units[blockIdx.x];  // first = component
    const HuffImage& im = images[u.image];
    const int c = (int)u.first;
    const uint32_t h = im.comp_h[c], v = im.comp_v[c], bpc = h * v;  // blocks of this component per MCU
    const uint32_t bpm = im.blocks_per_mcu, k0 = im.comp_k0[c];
    const uint32_t mcus = im.mcus_x * im.mcus_y;
    const uint32_t n = mcus * bpc;
    int16_t* coef = im.coef[c];
    const int16_t* diff = im.dc_diff;
    const uint32_t bw = im.blocks_w[c], mcus_x = im.mcus_x;
    const uint32_t per = (n + kThreads - 1) / kThreads;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    int sum = 0;
    {
        uint32_t mcu = lo / bpc, jj = lo - mcu * bpc;
        for (uint32_t s = lo; s < hi; s++) {
            sum += diff[mcu * bpm + k0 + jj];
            if (++jj == bpc) {
                jj = 0;
                mcu++;
            }
        }
    }
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < kThreads; off <<= 1) {
        int t = threadIdx.x >= (unsigned)off ? s_sum[threadIdx.x - off] : 0;
        __syncthreads();
        s_sum[threadIdx.x] += t;
        __syncthreads();
    }
    int run = threadIdx.x ? s_sum[threadIdx.x - 1] : 0;
    uint32_t mcu = lo / bpc, jj = lo - mcu * bpc;
    uint32_t my = mcu / mcus_x, mx = mcu - my * mcus_x;
    for (uint32_t s = lo; s < hi; s++) {
        run += diff[mcu * bpm + k0 + jj];
        const uint32_t dy = jj / h, dx = jj - dy * h;
        coef[((size_t)(my * v + dy) * bw + (mx * h + dx)) * 64] = (int16_t)run;
        if (++jj == bpc) {
            jj = 0;
            mcu++;
            if (++mx == mcus_x) {
                mx = 0;
                my++;
            }
        }
    }
}

}  // namespace

int launch_destuff(HuffImage* images, const HuffUnit* chunk_units, int nchunks, uint32_t* drops, void* stream)
{
    if (nchunks <= 0) return 0;
    hipLaunchKernelGGL(destuff_count_kernel, dim3(nchunks), dim3(kThreads), 0, (hipStream_t)stream, images, chunk_units, drops);
    hipLaunchKernelGGL(destuff_compact_kernel, dim3(nchunks), dim3(kThreads), 0, (hipStream_t)stream, images, chunk_units, drops);
    return (int)hipGetLastError();
}

int launch_huff_sync(const HuffImage* images, const HuffUnit* units, int nunits, unsigned long long* states, unsigned int* changed, int first_pass,
                     unsigned pool_bytes, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(huff_sync_kernel, dim3(nunits), dim3(kThreads), pool_bytes, (hipStream_t)stream, images, units, states, changed, first_pass);
    return (int)hipGetLastError();
}

int launch_huff_scan(HuffImage* images, const uint32_t* image_list, int nimages, const unsigned long long* states, uint32_t* first_block, void* stream)
{
    if (nimages <= 0) return 0;
    hipLaunchKernelGGL(huff_scan_kernel, dim3(nimages), dim3(kThreads), 0, (hipStream_t)stream, images, image_list, states, first_block);
    return (int)hipGetLastError();
}

int launch_huff_write(HuffImage* images, const HuffUnit* units, int nunits, const unsigned long long* states, const uint32_t* first_block,
                      unsigned pool_bytes, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(huff_write_kernel, dim3(nunits), dim3(kThreads), pool_bytes, (hipStream_t)stream, images, units, states, first_block);
    return (int)hipGetLastError();
}

int launch_huff_dc(const HuffImage* images, const HuffUnit* units, int nunits, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(huff_dc_kernel, dim3(nunits), dim3(kThreads), 0, (hipStream_t)stream, images, units);
    return (int)hipGetLastError();
}

}  // namespace hipjpeg
