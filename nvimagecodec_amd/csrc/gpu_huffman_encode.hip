// gpu_huffman_encode.hip -- gfx950 kernels of the GPU entropy coder (see gpu_huffman_encode.h for the pipeline).
// Arithmetic follows entropy_encode.cpp (the host coder, itself pinned byte-for-byte to libjpeg-turbo's output by the encode
// goldens): jchuff.c encode_one_block for the symbols, jccoefct.c compress_data for the dummy blocks of the MCU padding.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "gpu_huffman_encode.h"

namespace hipjpeg {

namespace {

constexpr int kThreads = 256;

#define HJ_LDS __attribute__((address_space(3)))

struct LdsTables {
    uint16_t dc_code[2][16];
    uint16_t ac_code[2][256];
    uint8_t dc_size[2][16];
    uint8_t ac_size[2][256];
};
static_assert(sizeof(LdsTables) == sizeof(StandardCodeTables), "same layout as the host struct");

__device__ __forceinline__ void load_tables(LdsTables* dst, const StandardCodeTables* src)
{
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    for (int i = threadIdx.x; i < (int)(sizeof(LdsTables) / 4); i += kThreads) d[i] = s[i];
}

// Where block s of the scan (MCU order) lives, and where the previous block of the same component is.
struct BlockRef {
    int c;              // component
    uint32_t bx, by;    // block coordinates in the component's grid
    bool has_prev;
    uint32_t pbx, pby;  // the same component's previous block in scan order (DC predictor)
};

__device__ __forceinline__ BlockRef locate(const HencImage& im, uint32_t s)
{
    BlockRef r;
    const uint32_t mcu = s / im.bpm, k = s - mcu * im.bpm;
    const uint32_t my = mcu / im.mcus_x, mx = mcu - my * im.mcus_x;
    uint32_t mh = 1, mv = 1, j = 0;
    r.c = 0;
    if (im.ncomp == 3) {
        const uint32_t nl = im.hs * im.vs;
        if (k < nl) {
            j = k;
            mh = im.hs;
            mv = im.vs;
        } else {
            r.c = (int)(k - nl + 1);
        }
    }
    const uint32_t dy = j / mh, dx = j - dy * mh;
    r.bx = mx * mh + dx;
    r.by = my * mv + dy;
    if (j > 0) {
        const uint32_t pj = j - 1, pdy = pj / mh, pdx = pj - pdy * mh;
        r.has_prev = true;
        r.pbx = mx * mh + pdx;
        r.pby = my * mv + pdy;
    } else if (mcu > 0) {
        const uint32_t pm = mcu - 1, pmy = pm / im.mcus_x, pmx = pm - pmy * im.mcus_x;
        r.has_prev = true;
        r.pbx = pmx * mh + (mh - 1);
        r.pby = pmy * mv + (mv - 1);
    } else {
        r.has_prev = false;
        r.pbx = r.pby = 0;
    }
    return r;
}

// DC value libjpeg gives a block: real blocks their own; dummy blocks the DC of the preceding block in MCU order
// (entropy_encode.cpp BlockSource::dc_of).
__device__ __forceinline__ int dc_value(const HencImage& im, int c, uint32_t bx, uint32_t by)
{
    const uint32_t mh = (c == 0 && im.ncomp == 3) ? im.hs : 1;
    while (by >= im.real_h[c]) {
        bx = (bx / mh) * mh + mh - 1;
        by--;
    }
    if (bx >= im.real_w[c]) bx = im.real_w[c] - 1;
    return im.coef[c][((size_t)by * im.blocks_w[c] + bx) * 64];
}

__device__ __forceinline__ int bit_length(unsigned v) { return v ? 32 - __builtin_clz(v) : 0; }

// Bit sink of the write kernel.  The workgroup's 256 blocks cover one contiguous bit range of the image's bit buffer; that
// range is assembled in LDS (zeroed, bits OR-ed in with ds_or -- LDS atomics are cheap) and then copied out as whole words,
// coalesced.  Only the first and the last word of the range can be shared with a neighbouring workgroup: those two go out
// with a global atomic OR (the bit buffer starts out zeroed).  A range that does not fit the window (very high bit rates)
// falls back to OR-ing every word into the global buffer directly.
constexpr int kWindowWords = 4096;  // 16 KB: 256 blocks x 64 bytes on average

struct Emitter {
    uint32_t* gwords;             // the image's bit buffer
    HJ_LDS uint32_t* window;      // LDS window (LDS address 0 is a valid place for it: never test this pointer)
    bool use_window;              // false: write through to gwords
    uint32_t window_word0;        // index (in gwords) of window[0]
    unsigned long long acc;
    uint32_t n;        // valid bits at the low end of acc
    uint32_t widx;     // next word (index into gwords)
    uint32_t emitted;  // bits of this block so far
    __device__ __forceinline__ void start(uint8_t* raw, uint32_t off, HJ_LDS uint32_t* win, bool use_win, uint32_t win_word0)
    {
        gwords = reinterpret_cast<uint32_t*>(raw);
        window = win;
        use_window = use_win;
        window_word0 = win_word0;
        acc = 0;
        n = off & 31;
        widx = off >> 5;
        emitted = 0;
    }
    __device__ __forceinline__ void out(uint32_t w)  // w: bits in stream order, most significant first
    {
        if (use_window)
            __hip_atomic_fetch_or(&window[widx - window_word0], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
            atomicOr(&gwords[widx], __builtin_bswap32(w));
        widx++;
    }
    __device__ __forceinline__ void put(uint32_t bits, uint32_t size)  // size <= 16, bits already masked
    {
        acc = (acc << size) | bits;
        n += size;
        emitted += size;
        if (n >= 32) {
            out((uint32_t)(acc >> (n - 32)));
            n -= 32;
        }
    }
    __device__ __forceinline__ void finish()
    {
        if (n > 0) out((uint32_t)(acc << (32 - n)));
    }
};

// One block.  q = its 64 coefficients in zigzag order (two per dword), ignored for dummy blocks.  Returns the bit length.
template <bool WRITE>
__device__ __forceinline__ uint32_t code_block(const uint4 (&q)[8], bool real, int diff, const HJ_LDS LdsTables* T, int ti, Emitter* em)
{
    uint32_t len;
    {
        const unsigned t = (unsigned)(diff < 0 ? -diff : diff);
        const int nb = bit_length(t);
        const uint32_t size = T->dc_size[ti][nb];
        len = size + nb;
        if (WRITE) {
            em->put(T->dc_code[ti][nb], size);
            if (nb) em->put((uint32_t)(diff < 0 ? diff - 1 : diff) & ((1u << nb) - 1), nb);
        }
    }
    if (!real) {  // dummy block: all AC zero -> EOB
        const uint32_t size = T->ac_size[ti][0];
        if (WRITE) em->put(T->ac_code[ti][0], size);
        return len + size;
    }
    const uint32_t w[32] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w,
                            q[4].x, q[4].y, q[4].z, q[4].w, q[5].x, q[5].y, q[5].z, q[5].w, q[6].x, q[6].y, q[6].z, q[6].w, q[7].x, q[7].y, q[7].z, q[7].w};
    int run = 0;
#pragma unroll
    for (int k = 1; k < 64; k++) {
        const int v = (k & 1) ? ((int)w[k >> 1] >> 16) : ((int)(w[k >> 1] << 16) >> 16);
        if (v == 0) {
            run++;
            continue;
        }
        while (run > 15) {  // ZRL
            const uint32_t size = T->ac_size[ti][0xF0];
            len += size;
            if (WRITE) em->put(T->ac_code[ti][0xF0], size);
            run -= 16;
        }
        const int nb = bit_length((unsigned)(v < 0 ? -v : v));
        const int sym = (run << 4) + nb;
        const uint32_t size = T->ac_size[ti][sym];
        len += size + nb;
        if (WRITE) {
            em->put(T->ac_code[ti][sym], size);
            em->put((uint32_t)(v < 0 ? v - 1 : v) & ((1u << nb) - 1), nb);
        }
        run = 0;
    }
    if (run > 0) {
        const uint32_t size = T->ac_size[ti][0];
        len += size;
        if (WRITE) em->put(T->ac_code[ti][0], size);
    }
    return len;
}

// The symbols of one block, counted instead of coded (jchuff.c htest_one_block): hist[0][category] for the DC difference, hist[1][run/size]
// for the coefficients, ZRL and EOB included -- the same walk as code_block.
__device__ __forceinline__ void count_block(const uint4 (&q)[8], bool real, int diff, HJ_LDS uint32_t (*hist)[256])
{
    __hip_atomic_fetch_add(&hist[0][bit_length((unsigned)(diff < 0 ? -diff : diff))], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (!real) {
        __hip_atomic_fetch_add(&hist[1][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    const uint32_t w[32] = {q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w, q[3].x, q[3].y, q[3].z, q[3].w,
                            q[4].x, q[4].y, q[4].z, q[4].w, q[5].x, q[5].y, q[5].z, q[5].w, q[6].x, q[6].y, q[6].z, q[6].w, q[7].x, q[7].y, q[7].z, q[7].w};
    int run = 0;
#pragma unroll
    for (int k = 1; k < 64; k++) {
        const int v = (k & 1) ? ((int)w[k >> 1] >> 16) : ((int)(w[k >> 1] << 16) >> 16);
        if (v == 0) {
            run++;
            continue;
        }
        while (run > 15) {
            __hip_atomic_fetch_add(&hist[1][0xF0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            run -= 16;
        }
        __hip_atomic_fetch_add(&hist[1][(run << 4) + bit_length((unsigned)(v < 0 ? -v : v))], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        run = 0;
    }
    if (run > 0) __hip_atomic_fetch_add(&hist[1][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Loads what code_block needs for block s.
__device__ __forceinline__ void fetch_block(const HencImage& im, uint32_t s, uint4 (&q)[8], bool* real, int* diff, int* ti)
{
    const BlockRef r = locate(im, s);
    *ti = r.c == 0 ? 0 : 1;
    *real = r.bx < im.real_w[r.c] && r.by < im.real_h[r.c];
    int dc;
    if (*real) {
        const uint4* p = reinterpret_cast<const uint4*>(im.coef[r.c] + ((size_t)r.by * im.blocks_w[r.c] + r.bx) * 64);
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] = p[i];
        dc = (int)(short)(q[0].x & 0xFFFF);
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] = make_uint4(0u, 0u, 0u, 0u);
        dc = dc_value(im, r.c, r.bx, r.by);
    }
    const int pred = r.has_prev ? dc_value(im, r.c, r.pbx, r.pby) : 0;
    *diff = dc - pred;
}

__global__ __launch_bounds__(kThreads) void henc_hist_kernel(const HencImage* __restrict__ images, const HencUnit* __restrict__ units)
{
    __shared__ uint32_t hist[2][2][256];
    const HencUnit u = units[blockIdx.x];
    const HencImage& im = images[u.image];
    if (!im.hist) return;  // uniform: this image is coded with the standard tables
    for (int i = threadIdx.x; i < 1024; i += kThreads) (&hist[0][0][0])[i] = 0;
    __syncthreads();
    const uint32_t s = u.first + threadIdx.x;
    if (s < im.total_blocks) {
        uint4 q[8];
        bool real;
        int diff, ti;
        fetch_block(im, s, q, &real, &diff, &ti);
        count_block(q, real, diff, (HJ_LDS uint32_t(*)[256])hist[ti]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += kThreads) {
        const uint32_t v = (&hist[0][0][0])[i];
        if (v) atomicAdd(&im.hist[i], v);
    }
}

__global__ __launch_bounds__(kThreads) void henc_length_kernel(const HencImage* __restrict__ images, const HencUnit* __restrict__ units,
                                                               const StandardCodeTables* __restrict__ tables, uint16_t* __restrict__ block_bits)
{
    __shared__ LdsTables T;
    const HencUnit u = units[blockIdx.x];
    const HencImage& im = images[u.image];
    load_tables(&T, im.tables ? im.tables : tables);
    __syncthreads();
    const uint32_t s = u.first + threadIdx.x;
    if (s >= im.total_blocks) return;
    uint4 q[8];
    bool real;
    int diff, ti;
    fetch_block(im, s, q, &real, &diff, &ti);
    block_bits[im.first_block + s] = (uint16_t)code_block<false>(q, real, diff, (const HJ_LDS LdsTables*)&T, ti, nullptr);
}

// One workgroup per image: block_off = exclusive prefix sum of block_bits; total_bits[image] = the sum.
__global__ __launch_bounds__(kThreads) void henc_scan_kernel(const HencImage* __restrict__ images, const uint16_t* __restrict__ block_bits,
                                                             uint32_t* __restrict__ block_off, uint32_t* __restrict__ total_bits)
{
    __shared__ uint32_t s_sum[kThreads];
    const HencImage& im = images[blockIdx.x];
    const uint16_t* bits = block_bits + im.first_block;
    uint32_t* off = block_off + im.first_block;
    const uint32_t n = im.total_blocks;
    const uint32_t per = (n + kThreads - 1) / kThreads;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    constexpr int kBatch = 8;  // independent loads in flight per lane
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i += kBatch) {
        uint32_t d[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; k++) d[k] = i + k < hi ? bits[i + k] : 0u;
#pragma unroll
        for (int k = 0; k < kBatch; k++) sum += d[k];
    }
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < kThreads; d <<= 1) {
        const uint32_t v = threadIdx.x >= (unsigned)d ? s_sum[threadIdx.x - d] : 0;
        __syncthreads();
        s_sum[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? s_sum[threadIdx.x - 1] : 0;
    for (uint32_t i = lo; i < hi; i += kBatch) {
        uint32_t d[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; k++) d[k] = i + k < hi ? bits[i + k] : 0u;
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            if (i + k < hi) off[i + k] = run;
            run += d[k];
        }
    }
    if (threadIdx.x == kThreads - 1) total_bits[blockIdx.x] = s_sum[kThreads - 1];
}

// Zeroes the bit buffers (16 bytes per lane).
__global__ __launch_bounds__(kThreads) void henc_zero_kernel(uint4* __restrict__ p, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

__global__ __launch_bounds__(kThreads) void henc_write_kernel(const HencImage* __restrict__ images, const HencUnit* __restrict__ units,
                                                              const StandardCodeTables* __restrict__ tables, const uint32_t* __restrict__ block_off,
                                                              const uint16_t* __restrict__ block_bits)
{
    __shared__ LdsTables T;
    __shared__ uint32_t window[kWindowWords];
    __shared__ uint32_t span[2];  // first word of the workgroup's range, number of words (0: does not fit the window)
    const HencUnit u = units[blockIdx.x];
    const HencImage& im = images[u.image];
    load_tables(&T, im.tables ? im.tables : tables);
    const int t = threadIdx.x;
    const uint32_t s = u.first + t;
    const bool live = s < im.total_blocks;
    if (t == 0) {
        const uint32_t last = min(u.first + kThreads, im.total_blocks) - 1;
        const uint32_t first_bit = block_off[im.first_block + u.first];
        // one past the range's last bit; the image's last block is followed by up to 7 padding bits.  The range must not be
        // overestimated: its last word is the one that may be shared with the next workgroup.
        const uint32_t end_bit = block_off[im.first_block + last] + block_bits[im.first_block + last] + (last == im.total_blocks - 1 ? 7u : 0u);
        const uint32_t w0 = first_bit >> 5, w1 = (end_bit - 1) >> 5;
        span[0] = w0;
        span[1] = (w1 - w0 + 1 <= (uint32_t)kWindowWords) ? w1 - w0 + 1 : 0u;
    }
    __syncthreads();
    const uint32_t w0 = span[0], nwin = span[1];
    for (uint32_t i = t; i < nwin; i += kThreads) window[i] = 0;
    __syncthreads();
    if (live) {
        uint4 q[8];
        bool real;
        int diff, ti;
        fetch_block(im, s, q, &real, &diff, &ti);
        const uint32_t off = block_off[im.first_block + s];
        Emitter em;
        em.start(im.raw, off, (HJ_LDS uint32_t*)window, nwin != 0, w0);
        code_block<true>(q, real, diff, (const HJ_LDS LdsTables*)&T, ti, &em);
        if (s == im.total_blocks - 1) {
            // jchuff.c flush_bits: the last byte of the scan is filled up with one-bits
            const uint32_t padn = (8 - ((off + em.emitted) & 7)) & 7;
            if (padn) em.put((1u << padn) - 1, padn);
        }
        em.finish();
    }
    __syncthreads();
    if (nwin) {
        uint32_t* g = reinterpret_cast<uint32_t*>(im.raw);
        for (uint32_t i = t; i < nwin; i += kThreads) {
            const uint32_t w = __builtin_bswap32(window[i]);
            if (i == 0 || i == nwin - 1) {
                if (w) atomicOr(&g[w0 + i], w);  // may be shared with the neighbouring workgroup
            } else {
                g[w0 + i] = w;
            }
        }
    }
}

// 0x80 in every byte of x that is 0xFF
__device__ __forceinline__ uint32_t ff_mask(uint32_t x)
{
    const uint32_t y = ~x;
    return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
}

// valid bytes of this lane's 16-byte piece of chunk `chunk`
__device__ __forceinline__ uint32_t piece_len(const HencImage& im, uint32_t chunk)
{
    const uint32_t off = chunk * kHencChunk + threadIdx.x * 16;
    return off >= im.raw_bytes ? 0u : min(16u, im.raw_bytes - off);
}

__device__ __forceinline__ uint32_t count_ff(const uint4& v, uint32_t len)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t n = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t m = ff_mask(w[i]);
        const int valid = (int)len - 4 * i;  // bytes of this dword that belong to the stream
        if (valid <= 0)
            m = 0;
        else if (valid < 4)
            m &= (1u << (8 * valid)) - 1;
        n += __popc(m);
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

__global__ __launch_bounds__(kThreads) void henc_count_kernel(const HencImage* __restrict__ images, const HencUnit* __restrict__ units,
                                                              uint32_t* __restrict__ chunk_ff)
{
    __shared__ uint32_t scratch[4];
    const HencUnit u = units[blockIdx.x];
    const HencImage& im = images[u.image];
    const uint32_t len = piece_len(im, u.first);
    uint32_t n = 0;
    if (len) n = count_ff(*reinterpret_cast<const uint4*>(im.raw + (size_t)u.first * kHencChunk + threadIdx.x * 16), len);
    const uint32_t total = wg_sum(n, scratch);
    if (threadIdx.x == 0) chunk_ff[im.first_chunk + u.first] = total;
}

// One workgroup in total.  Per image: chunk_out[c] = 0xFF bytes in the chunks before c; final_len = header + data + stuffed
// zeros + EOI; final_off = files packed back to back at 16-byte boundaries.
__global__ __launch_bounds__(kThreads) void henc_layout_kernel(const HencImage* __restrict__ images, int nimages, const uint32_t* __restrict__ chunk_ff,
                                                               uint32_t* __restrict__ chunk_out, uint32_t* __restrict__ final_len,
                                                               unsigned long long* __restrict__ final_off)
{
    __shared__ unsigned long long s_sum[kThreads];
    const int per = (nimages + kThreads - 1) / kThreads;
    const int lo = min(nimages, (int)threadIdx.x * per), hi = min(nimages, lo + per);
    unsigned long long sum = 0;
    for (int i = lo; i < hi; i++) {
        const HencImage& im = images[i];
        uint32_t ff = 0;
        for (uint32_t c = 0; c < im.num_chunks; c++) {
            chunk_out[im.first_chunk + c] = ff;
            ff += chunk_ff[im.first_chunk + c];
        }
        const uint32_t len = im.header_bytes + im.raw_bytes + ff + 2;
        final_len[i] = len;
        sum += (len + 15) & ~15u;
    }
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < kThreads; d <<= 1) {
        const unsigned long long v = threadIdx.x >= (unsigned)d ? s_sum[threadIdx.x - d] : 0;
        __syncthreads();
        s_sum[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned long long run = threadIdx.x ? s_sum[threadIdx.x - 1] : 0;
    for (int i = lo; i < hi; i++) {
        final_off[i] = run;
        run += (final_len[i] + 15) & ~15u;
    }
}

// Per chunk: bytes with a 0x00 behind every 0xFF, assembled in LDS at the destination's misalignment and copied out as
// dwords (bytes at the ragged ends, which neighbouring chunks share).  Chunk 0 also writes the header, the last chunk EOI.
__global__ __launch_bounds__(kThreads) void henc_expand_kernel(const HencImage* __restrict__ images, const HencUnit* __restrict__ units,
                                                               const uint32_t* __restrict__ chunk_out, const uint32_t* __restrict__ final_len,
                                                               const unsigned long long* __restrict__ final_off, uint8_t* __restrict__ arena, int nchunks)
{
    __shared__ uint32_t wave_base[4];
    __shared__ uint32_t out_words[2 * kHencChunk / 4 + 2];
    HJ_LDS uint8_t* out = (HJ_LDS uint8_t*)out_words;
    const int t = threadIdx.x;
    // A few resident workgroups walk the chunk list: with the files going straight to host memory every store waits on PCIe,
    // and one workgroup per chunk would fill every wave slot of the chip with waves that only wait -- starving whatever runs
    // on the other streams (measured: the forward kernel of the next batch stretched from 1.1 to 3.7 ms).
    for (int unit = blockIdx.x; unit < nchunks; unit += gridDim.x) {
    const HencUnit u = units[unit];
    const HencImage& im = images[u.image];
    uint8_t* file = arena + final_off[u.image];
    const uint32_t ff_before = chunk_out[im.first_chunk + u.first];
    const uint32_t dst_off = im.header_bytes + u.first * kHencChunk + ff_before;  // inside the file
    const uint32_t a = (uint32_t)((uintptr_t)(file + dst_off) & 3);

    const uint32_t len = piece_len(im, u.first);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (len) v = *reinterpret_cast<const uint4*>(im.raw + (size_t)u.first * kHencChunk + t * 16);
    const uint32_t n = len ? count_ff(v, len) : 0;
    uint32_t incl = n;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t x = __shfl_up(incl, d);
        if ((t & 63) >= d) incl += x;
    }
    if ((t & 63) == 63) wave_base[t >> 6] = incl;
    __syncthreads();
    uint32_t excl = incl - n;
    for (int w = 0; w < (t >> 6); w++) excl += wave_base[w];
    const uint32_t chunk_ffs = wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3];
    if (len) {
        uint32_t o = a + t * 16 + excl;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (uint32_t i = 0; i < len; i++) {
            const uint8_t b = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
            out[o++] = b;
            if (b == 0xFF) out[o++] = 0;
        }
    }
    __syncthreads();
    const uint32_t chunk_len = min((uint32_t)kHencChunk, im.raw_bytes - u.first * kHencChunk);
    const uint32_t n_out = chunk_len + chunk_ffs;
    uint8_t* dst = file + dst_off - a;  // 4-byte aligned
    const uint32_t first_full = a ? 1 : 0, end_full = (a + n_out) / 4;
    for (uint32_t d = first_full + t; d < end_full; d += kThreads) reinterpret_cast<uint32_t*>(dst)[d] = out_words[d];
    if (t < 4) {
        if (a && (uint32_t)t >= a && (uint32_t)t < a + n_out) dst[t] = out[t];  // head
        const uint32_t tail = max(end_full, first_full) * 4 + t;
        if (tail >= a && tail < a + n_out) dst[tail] = out[tail];
    }
    if (u.first == 0)
        for (uint32_t i = t; i < im.header_bytes; i += kThreads) file[i] = im.header[i];
    if (u.first + 1 == im.num_chunks && t == 0) {
        const uint32_t flen = final_len[u.image];
        file[flen - 2] = 0xFF;
        file[flen - 1] = 0xD9;
    }
    __syncthreads();  // the staging buffer is reused by the next chunk
    }
}

}  // namespace

constexpr int kExpandGrid = 64;  // resident workgroups of the expand kernel: enough stores in flight for PCIe (swept 32..1024 on MI355X)

int launch_henc_hist(const HencImage* images, const HencUnit* units, int nunits, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(henc_hist_kernel, dim3(nunits), dim3(kThreads), 0, (hipStream_t)stream, images, units);
    return (int)hipGetLastError();
}

int launch_henc_length(const HencImage* images, const HencUnit* units, int nunits, const StandardCodeTables* tables, uint16_t* block_bits, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(henc_length_kernel, dim3(nunits), dim3(kThreads), 0, (hipStream_t)stream, images, units, tables, block_bits);
    return (int)hipGetLastError();
}

int launch_henc_scan(const HencImage* images, int nimages, const uint16_t* block_bits, uint32_t* block_off, uint32_t* total_bits, void* stream)
{
    if (nimages <= 0) return 0;
    hipLaunchKernelGGL(henc_scan_kernel, dim3(nimages), dim3(kThreads), 0, (hipStream_t)stream, images, block_bits, block_off, total_bits);
    return (int)hipGetLastError();
}

int launch_henc_write(const HencImage* images, const HencUnit* units, int nunits, const StandardCodeTables* tables, const uint32_t* block_off,
                      const uint16_t* block_bits, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(henc_write_kernel, dim3(nunits), dim3(kThreads), 0, (hipStream_t)stream, images, units, tables, block_off, block_bits);
    return (int)hipGetLastError();
}

int launch_henc_zero(void* p, size_t bytes, void* stream)
{
    const size_t n16 = (bytes + 15) / 16;
    if (n16 == 0) return 0;
    hipLaunchKernelGGL(henc_zero_kernel, dim3((unsigned)((n16 + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream, static_cast<uint4*>(p), n16);
    return (int)hipGetLastError();
}

int launch_henc_count(const HencImage* images, const HencUnit* chunk_units, int nchunks, uint32_t* chunk_ff, void* stream)
{
    if (nchunks <= 0) return 0;
    hipLaunchKernelGGL(henc_count_kernel, dim3(nchunks), dim3(kThreads), 0, (hipStream_t)stream, images, chunk_units, chunk_ff);
    return (int)hipGetLastError();
}

int launch_henc_layout(const HencImage* images, int nimages, const uint32_t* chunk_ff, uint32_t* chunk_out, uint32_t* final_len,
                       unsigned long long* final_off, void* stream)
{
    if (nimages <= 0) return 0;
    hipLaunchKernelGGL(henc_layout_kernel, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, images, nimages, chunk_ff, chunk_out, final_len, final_off);
    return (int)hipGetLastError();
}

int launch_henc_expand(const HencImage* images, const HencUnit* chunk_units, int nchunks, const uint32_t* chunk_out, const uint32_t* final_len,
                       const unsigned long long* final_off, uint8_t* arena, void* stream)
{
    if (nchunks <= 0) return 0;
    static const int tuned = getenv("HIPJPEG_EXPAND_GRID") ? atoi(getenv("HIPJPEG_EXPAND_GRID")) : kExpandGrid;  // dev aid
    const int grid = nchunks < tuned ? nchunks : (tuned > 0 ? tuned : kExpandGrid);
    hipLaunchKernelGGL(henc_expand_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, images, chunk_units, chunk_out, final_len, final_off,
                       arena, nchunks);
    return (int)hipGetLastError();
}

}  // namespace hipjpeg
