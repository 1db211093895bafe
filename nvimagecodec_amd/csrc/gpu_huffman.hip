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
//   huff_pos_kernel    where every block starts (walk from the converged states)
//   huff_blocks_kernel one lane per block: decode into LDS, store whole 128-byte lines (DC differences to a compact array)
//   huff_dc_kernel     per (image, component): DC differences -> DC values, stored into the blocks
#include <hip/hip_runtime.h>

#include "gpu_huffman.h"
#include "huffman_gpu_core.h"

namespace hipjpeg {

namespace {

constexpr int kThreads = 256;

constexpr int kRowShift = kSubseqWords == 32 ? 5 : kSubseqWords == 16 ? 4 : kSubseqWords == 64 ? 6 : -1;  // log2(words per subsequence)
static_assert(kRowShift > 0, "subsequences of 512, 1024 or 2048 bits");

#define HJ_LDS __attribute__((address_space(3)))
#define HJ_GLOBAL __attribute__((address_space(1)))
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // plain vector type: usable behind address-space pointers

struct KSlot {
    int16_t* base;  // coef[comp] + blk0 * 64
    uint32_t stride_y, stride_x;
};

// Stream words in LDS: every subsequence ("row") is staged together with the kStagedExtra words behind it, which cover the bit
// reader's look-ahead (the last symbol of a walk starts before the row's end: its window reaches into word 32, the word
// requested ahead is word 33) -- a walk never leaves its row -- and the rows are TRANSPOSED: word k of row r sits at index
// k * rows + r.  The bank of an access is then r mod 32 whatever k is: lanes that walk their own rows never collide, however
// far apart their positions are, and stepping to the next word is adding a constant to an address (BitReader's cursor).
constexpr int kStagedExtra = 2;
constexpr int kRowWords = kSubseqWords + kStagedExtra;
static_assert(kSubseqWords % 4 == 0 && kStagedExtra == 2, "stage_rows: 16-byte groups and one 8-byte group per row");
__host__ __device__ constexpr int staged_lds_words(int rows) { return rows * kRowWords; }

struct WgShared {
    uint32_t stream[staged_lds_words(kSyncThreads)];
    unsigned long long end[kSyncThreads];  // end state of lane t's subsequence ([0] = halo = state entering the workgroup)
    uint32_t tsel[10];
    uint32_t queue[kSyncThreads];
    uint32_t wave_count[kSyncThreads / 64];
};

// restart boundaries live in global memory (a handful of reads per decode; the destuffed positions come from the host's
// marker walk)
__device__ __forceinline__ uint32_t load_boundary(const uint32_t* boundaries, uint32_t n, uint32_t i)
{
    return i < n ? ((const HJ_GLOBAL uint32_t*)boundaries)[i] : 0xFFFFFFFFu;
}
// LDS accessors for decode_subsequence.
struct DevEnv {
    uint32_t stream_base;  // LDS byte address of the staged words
    uint32_t word0;        // image word index of the first word of staged row 0
    uint32_t row_addr;     // this lane's walk: LDS byte address of word 0 of its row ...
    uint32_t row_word0;    // ... and the image word index of that word (set_row)
    uint32_t pool;         // LDS byte address of the lookup tables
    const HJ_LDS uint32_t* tsel;  // per MCU position: BYTE offsets of the DC / AC first-level tables, packed lo/hi
    const uint32_t* boundaries;
    uint32_t num_boundaries;
    __device__ __forceinline__ uint32_t boundary(uint32_t i) const { return load_boundary(boundaries, num_boundaries, i); }
    static constexpr uint32_t kCursorStep = (uint32_t)kSyncThreads * 4u;  // consecutive words of a row, in bytes
    __device__ __forceinline__ void set_row(int row)
    {
        row_addr = stream_base + ((uint32_t)row << 2);
        row_word0 = word0 + ((uint32_t)row << kRowShift);
    }
    __device__ __forceinline__ uint32_t cursor(uint32_t i) const { return row_addr + (i - row_word0) * kCursorStep; }
    __device__ __forceinline__ uint32_t fetch(uint32_t c) const { return *(const HJ_LDS uint32_t*)(uintptr_t)c; }
    __device__ __forceinline__ uint32_t tables(int k) const { return tsel[k]; }
    __device__ __forceinline__ uint32_t lookup1(uint32_t t, uint32_t w) const
    {
        return *(const HJ_LDS uint16_t*)(uintptr_t)(pool + t + ((w >> (31 - kHuffFastBits)) & ((2u << kHuffFastBits) - 2)));
    }
    __device__ __forceinline__ uint32_t lookup2(uint32_t e, uint32_t w) const
    {
        return *(const HJ_LDS uint16_t*)(uintptr_t)(pool + ((e & 0x1FFu) << 7) + ((w >> 15) & ((2u << kHuffSubBits) - 2)));
    }
    __device__ __forceinline__ uint32_t lookup_pair(uint32_t t, uint32_t w) const  // lookup1's address + a constant: one more LDS read, no more arithmetic
    {
        return *(const HJ_LDS uint16_t*)(uintptr_t)(pool + t + 2 * kPairOffset + ((w >> (31 - kHuffFastBits)) & ((2u << kHuffFastBits) - 2)));
    }
};

__device__ __forceinline__ uint32_t boundary0_of(const HuffImage& im, uint32_t subseq)
{
    return im.restart_interval ? ((const HJ_GLOBAL uint32_t*)im.sub_boundary)[subseq] : 0u;
}
// decode_subsequence with or without restart handling (wave-uniform choice: one image per workgroup)
template <class Env>
__device__ __forceinline__ SubseqState walk_subsequence(bool rst, const HuffGeom& geom, const Env& env, uint32_t begin, uint32_t limit, int z, int k,
                                                        uint32_t boundary0)
{
    return rst ? decode_subsequence<true>(geom, env, begin, limit, z, k, boundary0) : decode_subsequence<false>(geom, env, begin, limit, z, k);
}

// Cooperative staging of stream words: lane t brings in row t = subsequence first_row + t of the image (first_row may be -1:
// that row is filled with ones) and the kStagedExtra words behind it, transposed (see WgShared).  16-byte global loads, all of
// a lane's loads in flight together; words behind the end of the stream read as ones.
template <int ROWS>
__device__ __forceinline__ void stage_rows(uint32_t* lds_stream, const HuffImage& im, int first_row)
{
    const int t = threadIdx.x;
    const HJ_GLOBAL uint32_t* g = (const HJ_GLOBAL uint32_t*)im.stream;
    const int nwords = (int)im.stream_words;
    const int w0 = (first_row + t) * kSubseqWords;
    constexpr int kGroups = kSubseqWords / 4;  // 16-byte groups per row; the look-ahead words are one 8-byte group more
    u32x4 v[kGroups];
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 tail;
    {
        const int gd = w0 + kSubseqWords;
        const int gc = min(max(gd, 0), (nwords - 1) & ~1);
        const u32x2 x = *(const HJ_GLOBAL u32x2*)(g + gc);
        tail = (gd >= 0 && gd < nwords) ? x : u32x2{~0u, ~0u};
    }
#pragma unroll
    for (int i = 0; i < kGroups; i++) {
        const int gd = w0 + i * 4;
        // streams are allocated in whole 64-byte units: a 16-byte group that starts inside the stream is readable.  The
        // load itself is unconditional (clamped address) so that all of them are in flight together.
        const int gc = min(max(gd, 0), (nwords - 1) & ~3);
        const u32x4 x = *(const HJ_GLOBAL u32x4*)(g + gc);
        v[i] = (gd >= 0 && gd < nwords) ? x : u32x4{~0u, ~0u, ~0u, ~0u};
    }
    HJ_LDS uint32_t* dst = (HJ_LDS uint32_t*)lds_stream + t;
#pragma unroll
    for (int i = 0; i < kGroups; i++) {
        dst[(4 * i + 0) * ROWS] = __builtin_bswap32(v[i].x);
        dst[(4 * i + 1) * ROWS] = __builtin_bswap32(v[i].y);
        dst[(4 * i + 2) * ROWS] = __builtin_bswap32(v[i].z);
        dst[(4 * i + 3) * ROWS] = __builtin_bswap32(v[i].w);
    }
    dst[(kSubseqWords + 0) * ROWS] = __builtin_bswap32(tail.x);
    dst[(kSubseqWords + 1) * ROWS] = __builtin_bswap32(tail.y);
}

// lookup tables -> LDS
template <int THREADS>
__device__ __forceinline__ void stage_pool(HJ_LDS uint16_t* pool, const HuffImage& im)
{
    const HJ_GLOBAL u32x4* gp = (const HJ_GLOBAL u32x4*)im.pool;
    const uint32_t npool = im.pool_words >> 3;  // pool_words is a multiple of 64
    for (uint32_t i = threadIdx.x; i < npool; i += THREADS) {
        const u32x4 x = gp[i];
        HJ_LDS uint32_t* lp = reinterpret_cast<HJ_LDS uint32_t*>(pool) + i * 4;
        lp[0] = x.x;
        lp[1] = x.y;
        lp[2] = x.z;
        lp[3] = x.w;
    }
}

// per-MCU-position constants and the zigzag permutation -> LDS (needs >= 80 lanes)
__device__ __forceinline__ void stage_constants(uint32_t* tsel, KSlot* kslot, uint32_t* zzw, const HuffImage& im, bool with_addresses)
{
    const int t = threadIdx.x;
    if (t < 10) {
        const HuffK hk = im.k[t];
        tsel[t] = ((uint32_t)hk.tdc * 2) | ((uint32_t)hk.tac * 2 << 16);  // byte offsets (<= 2 * kMaxPoolWords < 65536)
        if (with_addresses) {
            KSlot s;
            s.base = im.coef[hk.comp & 3] + (size_t)hk.blk0 * 64;
            s.stride_y = hk.stride_y;
            s.stride_x = hk.stride_x;
            kslot[t] = s;
        }
    }
    if (zzw && t >= 64 && t < 80) {
        constexpr uint8_t zz[64] = HJ_ZIGZAG_DEVICE_TABLE;
        const int i = (t - 64) * 4;
        zzw[t - 64] = (uint32_t)zz[i] | ((uint32_t)zz[i + 1] << 8) | ((uint32_t)zz[i + 2] << 16) | ((uint32_t)zz[i + 3] << 24);
    }
}

__device__ __forceinline__ DevEnv make_env(WgShared& sh, HJ_LDS uint16_t* pool, uint32_t first, const HuffGeom& geom)
{
    DevEnv env;
    env.stream_base = (uint32_t)(uintptr_t)(HJ_LDS uint32_t*)sh.stream;
    env.word0 = (first - 1) * kSubseqWords;  // row 0 = the halo (wraps for the first workgroup of an image; so do the indices)
    env.row_addr = env.stream_base;
    env.row_word0 = env.word0;
    env.pool = (uint32_t)(uintptr_t)pool;
    env.tsel = (const HJ_LDS uint32_t*)sh.tsel;
    env.boundaries = geom.boundaries;
    env.num_boundaries = geom.num_boundaries;
    return env;
}

// Packs the lanes with has == true to the front: returns how many there are; *task = value of the t-th of them for the
// lanes t below that count, -1 for the others.  Contains barriers: call from uniform control flow.
__device__ __forceinline__ int compact_tasks(WgShared& sh, bool has, int value, int* task)
{
    const int t = threadIdx.x;
    const unsigned long long mask = __ballot(has);
    const int below = __popcll(mask & ((1ull << (t & 63)) - 1));
    if ((t & 63) == 0) sh.wave_count[t >> 6] = (uint32_t)__popcll(mask);
    __syncthreads();
    int base = 0, n = 0;
#pragma unroll
    for (int w = 0; w < kSyncThreads / 64; w++) {
        const int c = (int)sh.wave_count[w];
        if (w < (t >> 6)) base += c;
        n += c;
    }
    if (has) sh.queue[base + below] = (uint32_t)value;
    __syncthreads();
    *task = t < n ? (int)sh.queue[t] : -1;
    return n;
}

// ---- zero-copy input: bitstreams fetched from the caller's pinned host memory ------------------------------------------------
// One DMA call per image costs the submitting thread 15-20 us (256 of them: 5 ms per batch, measured); one kernel that pulls the bytes
// over PCIe costs one launch.  Few resident workgroups on purpose -- waves that wait on PCIe reads slow the kernels beside them as
// long as they occupy wave slots -- each walking the destuff kernels' chunk list (kDestuffChunk raw bytes per item).
constexpr int kGatherGroups = 64;
__global__ __launch_bounds__(kThreads) void gather_raw_kernel(const HuffImage* __restrict__ images, const HuffUnit* __restrict__ units, int nunits)
{
    for (int ui = blockIdx.x; ui < nunits; ui += gridDim.x) {
        const HuffUnit u = units[ui];
        const HuffImage& im = images[u.image];
        const uint8_t* src = im.raw_src;
        if (!src) continue;  // staged by the host (uniform per item)
        const uint32_t begin = u.first * (uint32_t)kDestuffChunk;
        if (begin >= im.raw_bytes) continue;
        const uint32_t n = min((uint32_t)kDestuffChunk, im.raw_bytes - begin);
        const uint8_t* s = src + begin;
        uint8_t* d = const_cast<uint8_t*>(im.raw) + begin;  // 16-byte aligned: raw offsets and kDestuffChunk are multiples of 16
        typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;  // the caller's buffer starts at any byte
        for (uint32_t i = threadIdx.x * 16u; i + 16u <= n; i += kThreads * 16u) *(HJ_GLOBAL u32x4*)(d + i) = *(const HJ_GLOBAL u32x4_unaligned*)(s + i);
        const uint32_t tail = n & ~15u;
        if (threadIdx.x < (n & 15u)) d[tail + threadIdx.x] = s[tail + threadIdx.x];
    }
}

// ---- byte-stuffing removal ------------------------------------------------------------------------------------------
// The file's entropy-coded segment escapes every 0xFF data byte as FF 00.  Two kernels turn it into the plain bitstream the
// decoders read: the first counts the stuffed bytes of every 16 KB chunk, the second compacts each chunk to its final
// position (chunk offset minus the stuffed bytes before it) through LDS.  Streams with anything else behind an FF (restart
// markers, fill bytes) never get here (gpu_entropy_eligible()).

// 0x80 in every byte of x (little-endian dword) that is 0x00 and whose predecessor byte is 0xFF; prev = the byte before x.
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) { return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu); }  // 0x80 per zero byte

// Bytes of x to drop: a 0x00 behind an 0xFF (stuffing), and both bytes of a restart marker FF D0..D7.  prev = the byte in
// front of x, next = the byte behind it.
__device__ __forceinline__ uint32_t stuffed_mask(uint32_t x, uint32_t prev, uint32_t next)
{
    const uint32_t ff_before = zero_bytes(~((x << 8) | prev));        // predecessor is 0xFF
    const uint32_t rst_x = zero_bytes((x & 0xF8F8F8F8u) ^ 0xD0D0D0D0u);  // x's byte is D0..D7
    const uint32_t second = (zero_bytes(x) | rst_x) & ff_before;      // stuffed zero, or the marker's second byte
    const uint32_t after = (x >> 8) | (next << 24);                   // successor of each byte
    const uint32_t first = zero_bytes(~x) & zero_bytes((after & 0xF8F8F8F8u) ^ 0xD0D0D0D0u);  // an FF in front of D0..D7
    return second | first;
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
    const uint32_t behind = p[64];  // the raw copy is padded with 16 neutral bytes: always readable
#pragma unroll
    for (int i = 0; i < 16; i++) {
        m[i] = stuffed_mask(x[i], prev, i < 15 ? (x[i < 15 ? i + 1 : 15] & 0xFFu) : behind);
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
            const uint32_t* words = reinterpret_cast<const uint32_t*>(im.raw + off);  // 16 neutral bytes of padding follow the data
            for (uint32_t i = 0; i < pieces * 4; i++) {
                const uint32_t w = words[i];
                n += __popc(stuffed_mask(w, prev, words[i + 1] & 0xFFu));
                prev = w >> 24;
            }
        }
    }
    const uint32_t total = wg_sum(n, scratch);
    if (threadIdx.x == 0) drops[im.first_chunk + u.first] = total;
}

// (256 x 1080p: 94 us for 2,048 chunks, eight resident per CU.  The time is the chunks' work, not their number: two chunks per workgroup 147 us,
// four 149.  Per CU and launch the LDS pipeline is busy 33 us -- 64 single-byte stores per lane -- and the vector ALUs 20.)
__global__ __launch_bounds__(kThreads) void destuff_compact_kernel(HuffImage* __restrict__ images, const HuffUnit* __restrict__ units,
                                                                   const uint32_t* __restrict__ drops, unsigned int* __restrict__ counters)
{
    // the stage's convergence counters (64 words) start at zero: cleared here, by the first kernel of the stage, instead of by a
    // fill kernel of their own in front of it
    if (blockIdx.x == 0 && threadIdx.x < 64 && counters) counters[threadIdx.x] = 0u;
    __shared__ uint32_t scratch[4];
    __shared__ uint32_t wave_base[4];
    // One LDS buffer, twice: first the raw chunk (loaded as consecutive dwords: whole lines per wave), then the compacted image.  Both are
    // PADDED by a dword per 64 bytes -- dword g sits at g + (g >> 4), byte k at k + (k >> 6 << 2): a lane owns 64 consecutive bytes, its
    // neighbour the next 64, and without the padding the lanes of a wave would all hit the same two banks (round 2's version did: 55 % of
    // its LDS cycles were bank conflicts, and its global loads touched 64 lines per instruction).
    __shared__ uint32_t buf[kDestuffChunk / 4 + kDestuffChunk / 64 + 8];
    HJ_LDS uint8_t* bytes = (HJ_LDS uint8_t*)buf;
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

    // raw chunk -> LDS (the raw copy is padded to 16 bytes with a neutral value and 16 more: whole 16-byte pieces are readable)
    {
        const uint32_t readable = (chunk_len + 15) / 16 * 4;  // dwords
        const HJ_GLOBAL uint32_t* src = (const HJ_GLOBAL uint32_t*)(im.raw + chunk_begin);
#pragma unroll
        for (uint32_t i = 0; i < kDestuffChunk / 4 / kThreads; i++) {
            const uint32_t g = i * kThreads + t;
            buf[g + (g >> 4)] = g < readable ? src[g] : 0x01010101u;
        }
    }
    const uint32_t before_chunk = chunk_begin > 0 ? im.raw[chunk_begin - 1] : 0u;                    // (lane 0's predecessor byte)
    const uint32_t behind_chunk = chunk_len == (uint32_t)kDestuffChunk ? im.raw[chunk_begin + kDestuffChunk] : 0x01u;  // (lane 255's successor)
    __syncthreads();

    const uint32_t off = chunk_begin + t * 64;
    uint32_t m[16], x[16];
    uint32_t n = 0, len = 0;
    if (off < raw_bytes) {
        len = min(64u, raw_bytes - off);
        const uint32_t pieces = (len + 15) / 16;
        uint32_t prev = t > 0 ? buf[(t - 1) * 17 + 15] >> 24 : before_chunk;
        for (uint32_t i = 0; i < 16; i++) x[i] = i < pieces * 4 ? buf[t * 17 + i] : 0x01010101u;
        const uint32_t behind = pieces == 4 ? (t < kThreads - 1 ? buf[(t + 1) * 17] & 0xFFu : behind_chunk) : 0x01u;
        for (uint32_t i = 0; i < 16; i++) {
            m[i] = stuffed_mask(x[i], prev, i < 15 ? (x[i < 15 ? i + 1 : 15] & 0xFFu) : behind);
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
    __syncthreads();  // (every lane holds its 64 bytes in registers from here on: the buffer is free for the compacted image)
    uint32_t excl = incl - n;
    for (int w = 0; w < (t >> 6); w++) excl += wave_base[w];
    const uint32_t chunk_drops = wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3];
    // kept bytes of this lane -> LDS
    auto padded = [](uint32_t k) { return k + ((k >> 6) << 2); };
    if (len) {
        uint32_t o = a + t * 64 - excl;
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t w = x[i], mask = m[i];
            if (i * 4 + 4 <= len && mask == 0) {
                bytes[padded(o)] = (uint8_t)w;
                bytes[padded(o + 1)] = (uint8_t)(w >> 8);
                bytes[padded(o + 2)] = (uint8_t)(w >> 16);
                bytes[padded(o + 3)] = (uint8_t)(w >> 24);
                o += 4;
            } else {
                for (uint32_t b = 0; b < 4; b++)
                    if (i * 4 + b < len && !((mask >> (8 * b + 7)) & 1)) bytes[padded(o++)] = (uint8_t)(w >> (8 * b));
            }
        }
    }
    __syncthreads();
    // LDS -> destination: whole dwords in the middle, single bytes at the ragged ends (neighbouring chunks share those dwords)
    const uint32_t n_out = chunk_len - chunk_drops;
    uint8_t* dst = const_cast<uint8_t*>(im.stream) + (gout - a);  // 4-byte aligned
    const uint32_t first_full = a ? 1 : 0, end_full = (a + n_out) / 4;
    for (uint32_t d = first_full + t; d < end_full; d += kThreads) reinterpret_cast<uint32_t*>(dst)[d] = buf[d + (d >> 4)];
    if (t < 4) {
        if (a && (uint32_t)t >= a && (uint32_t)t < a + n_out) dst[t] = bytes[padded((uint32_t)t)];  // head
        const uint32_t tail = max(end_full, first_full) * 4 + t;
        if (tail >= a && tail < a + n_out && tail >= 4 * first_full) dst[tail] = bytes[padded(tail)];
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
// incoming[]: per workgroup, the state it assumed to enter with the last time it ran.
// counters[0] += 1 for every workgroup whose outgoing state (end state of its last subsequence) differs from the published
// one (counters[1] in the first pass, where every state is new); counters[2..5] = statistics.
//
// first_pass: every lane decodes its subsequence from the assumed state "a block of MCU position 0 starts at the boundary";
// lane 0 does so for the LAST subsequence of the predecessor workgroup (the halo), whose end state is the workgroup's guess
// of the state it is entered with.  Then corrections ripple: a subsequence whose predecessor's end state differs from the
// one it was decoded from is decoded again.  Every round packs the subsequences to redo into the lowest lanes, so that the
// long tail of rounds with a handful of corrections occupies one wave instead of four.
// later passes: a workgroup whose incoming state still equals its guess has nothing to do and leaves before staging
// anything; otherwise the ripple starts from its first subsequence.
__global__ __launch_bounds__(kSyncThreads) void huff_sync_kernel(const HuffImage* __restrict__ images, const HuffUnit* __restrict__ units,
                                                             unsigned long long* __restrict__ states, unsigned long long* __restrict__ incoming,
                                                             unsigned int* __restrict__ counters, int first_pass, int max_rounds,
                                                             uint16_t* __restrict__ tail_tasks, uint32_t* __restrict__ tail_count)
{
    __shared__ WgShared sh;
    extern __shared__ uint16_t dyn_pool[];
    HJ_LDS uint16_t* pool = (HJ_LDS uint16_t*)dyn_pool;
    const HuffUnit u = units[blockIdx.x];
    const HuffImage& im = images[u.image];
    const HuffGeom geom = make_geom(im);
    const uint32_t nsub = (geom.total_bits + kSubseqBits - 1) / kSubseqBits;
    if (u.first >= nsub) return;  // uniform: the stream turned out shorter than planned
    unsigned long long* gstate = states + im.first_subseq;
    const int t = threadIdx.x;
    const uint32_t j = u.first - 1 + t;  // subsequence of lane t (lane 0: the halo; none for the first workgroup of an image)
    const bool owner = t >= 1 && j < nsub;
    const int n_rows = (int)min((uint32_t)kSyncThreads, nsub - u.first + 1);  // rows 1 .. n_rows-1 are owned
    unsigned long long in_state = 0;
    if (!first_pass) {
        in_state = u.first ? __hip_atomic_load(&gstate[u.first - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kSyncMask : 0ull;
        if (in_state == incoming[blockIdx.x]) return;  // uniform
    }
    stage_rows<kSyncThreads>(sh.stream, im, (int)u.first - 1);
    stage_pool<kSyncThreads>(pool, im);
    stage_constants(sh.tsel, nullptr, nullptr, im, false);
    __syncthreads();

    DevEnv env = make_env(sh, pool, u.first, geom);
    const bool rst = im.restart_interval != 0;
    unsigned long long old_global = ~0ull;
    int task = -1;
    if (first_pass) {
        const unsigned long long assumed = (unsigned long long)j * kSubseqBits;
        unsigned long long mine = 0;  // first workgroup of an image, lane 0: the exact initial state (bit 0, z 0, k 0)
        if (owner || (t == 0 && u.first > 0)) {
            const uint32_t bidx0 = boundary0_of(im, j);
            env.set_row(t);
            mine = pack_state(walk_subsequence(rst, geom, env, j * kSubseqBits, (j + 1) * kSubseqBits, 0, 0, bidx0));
        }
        sh.end[t] = mine;
        __syncthreads();
        if (owner && (sh.end[t - 1] & kSyncMask) != assumed) task = t;
    } else {
        if (owner) old_global = gstate[j];
        sh.end[t] = t == 0 ? in_state : old_global;
        if (t == 1) task = 1;
    }
    int rounds = 0;
    for (int round = 0; round < kSyncThreads + 2; round++) {
        __syncthreads();
        rounds++;
        unsigned long long now = 0;
        bool moved = false;
        if (task >= 0) {
            const SubseqState p = unpack_state(sh.end[task - 1]);
            env.set_row(task);  // the predecessor's walk ended on a symbol that starts in this row
            now = pack_state(walk_subsequence(rst, geom, env, p.end_bit, (u.first + task) * kSubseqBits, p.zk & 255, p.zk >> 8, boundary0_of(im, u.first - 1 + task)));
            moved = ((now ^ sh.end[task]) & kSyncMask) != 0;
        }
        __syncthreads();  // every start state has been read before any end state is replaced
        if (task >= 0) sh.end[task] = now;
        const int pending = compact_tasks(sh, moved && task + 1 < n_rows, task + 1, &task);
        if (pending == 0 || rounds >= max_rounds) {
            // what is left goes to the tail kernel: few subsequences per round, one lane busy per chain -- not worth holding
            // 47 KB of LDS for
            if (tail_count) {
                if (t < pending) tail_tasks[(size_t)blockIdx.x * kSyncThreads + t] = (uint16_t)task;
                if (t == 0) tail_count[blockIdx.x] = (uint32_t)pending;
            }
            break;
        }
    }
    __syncthreads();
    if (owner) {
        const unsigned long long mine = sh.end[t];
        if (mine != old_global) {
            __hip_atomic_store(&gstate[j], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == n_rows - 1 && ((mine ^ old_global) & kSyncMask) != 0) atomicAdd(counters + (first_pass ? 1 : 0), 1u);
        }
    }
    if (t == 0) {
        incoming[blockIdx.x] = first_pass ? (sh.end[0] & kSyncMask) : in_state;
        // statistics: correction rounds (sum, maximum) of the first launch / of the later launches
        atomicAdd(counters + (first_pass ? 2 : 4), (unsigned)rounds);
        atomicMax(counters + (first_pass ? 3 : 5), (unsigned)rounds);
    }
}

// ---- tail corrections ---------------------------------------------------------------------------------------------------
// After pass 0 and the first correction rounds most subsequences are settled; what remains are chains: a decoder that has
// not yet found the MCU phase hands a wrong end state to its successor, which must be decoded again, and so on for a few
// thousand bits (per round a third of the subsequences decoded still change their end state: 8-10 rounds for a 1080p picture).
// Each link is one full single-lane decode, strictly after the previous one, so the kernel's time is the longest chain -- IF
// every group is resident.  One WAVE per group of 255 subsequences; kTailWaves waves share a workgroup and with it the lookup
// tables in LDS; a wave has kTailSlots row buffers (a lane stages the row it is about to decode: reading the stream through
// the vector cache instead costs a lone lane a third more per round) and takes more tasks than that in several helpings -- only
// the first tail round has that many; end states live in global memory.  So all groups of a 256 x 1080p batch are resident at
// once.  (One wave per workgroup with its own 12 KB of tables and 64 rows: 7 per CU, the batch took 2.1 shifts -- 543 us
// against 300 us for a batch that fits.)  The waves of a workgroup never synchronise with each other after the tables are in
// place.
constexpr int kTailWaves = 4;
constexpr int kTailThreads = 64 * kTailWaves;
#ifndef HJ_TAIL_SLOTS
#define HJ_TAIL_SLOTS 32  // A/B switch (tools/ab_entropy.sh): 64 row buffers per wave halve the helpings of the first tail round -- and take 414 us against 287
#endif
constexpr int kTailSlots = HJ_TAIL_SLOTS;
static_assert(kTailSlots == 32 || kTailSlots == 64, "a power of two, at most one row buffer per lane");
constexpr int kTailRowWords = kSubseqWords + 4;  // whole 16-byte groups

struct TailWave {
    uint32_t rows[kTailSlots * kTailRowWords];
    uint16_t list[2][kSyncThreads];  // subsequences to decode this round / next round
    uint32_t count[2];
};
struct TailShared {
    TailWave wave[kTailWaves];
    uint32_t tsel[10];
};

struct TailEnv {
    uint32_t row_base;  // LDS byte address of this lane's row buffer
    uint32_t word0;     // image word index of its first word
    uint32_t pool;
    const HJ_LDS uint32_t* tsel;
    const uint32_t* boundaries;
    uint32_t num_boundaries;
    __device__ __forceinline__ uint32_t boundary(uint32_t i) const { return load_boundary(boundaries, num_boundaries, i); }
    static constexpr uint32_t kCursorStep = (uint32_t)kTailSlots * 4u;  // the slots are transposed like the rows of the sync kernel
    __device__ __forceinline__ uint32_t cursor(uint32_t i) const { return row_base + (i - word0) * kCursorStep; }
    __device__ __forceinline__ uint32_t fetch(uint32_t c) const { return *(const HJ_LDS uint32_t*)(uintptr_t)c; }
    __device__ __forceinline__ uint32_t tables(int k) const { return tsel[k]; }
    __device__ __forceinline__ uint32_t lookup1(uint32_t t, uint32_t w) const
    {
        return *(const HJ_LDS uint16_t*)(uintptr_t)(pool + t + ((w >> (31 - kHuffFastBits)) & ((2u << kHuffFastBits) - 2)));
    }
    __device__ __forceinline__ uint32_t lookup2(uint32_t e, uint32_t w) const
    {
        return *(const HJ_LDS uint16_t*)(uintptr_t)(pool + ((e & 0x1FFu) << 7) + ((w >> 15) & ((2u << kHuffSubBits) - 2)));
    }
    __device__ __forceinline__ uint32_t lookup_pair(uint32_t t, uint32_t w) const  // lookup1's address + a constant: one more LDS read, no more arithmetic
    {
        return *(const HJ_LDS uint16_t*)(uintptr_t)(pool + t + 2 * kPairOffset + ((w >> (31 - kHuffFastBits)) & ((2u << kHuffFastBits) - 2)));
    }
};

// orders a wave's own LDS / global traffic between its lanes (the waves of the tail kernel run on their own)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// RIPPLE mode (the launches behind the first): nothing is handed over by the sync kernel; a group whose predecessor's last end state
// is no longer the state it was entered with starts over at its first subsequence, and the correction runs as far as it changes
// anything.  counters[0] counts the groups whose own last end state moved -- zero: the batch has converged.

// the state in front of group `u` as it is now (0 for the first group of an image)
__device__ __forceinline__ unsigned long long state_in_front(const unsigned long long* states, const HuffImage& im, const HuffUnit u)
{
    return u.first ? __hip_atomic_load(&states[im.first_subseq + u.first - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kSyncMask : 0ull;
}

// the chains of one group (sync unit `ui`), walked by one wave
// ---- one chain link by a whole wave ---------------------------------------------------------------------------------------------
// The last links of a chain are one lane's work with 63 lanes idle, 28 us per subsequence: a lone wave pays 10+ cycles for every
// dependent instruction of the decode step.  With one or two chains left the wave turns to ONE subsequence instead
// (cooperative_subsequence in huffman_gpu_core.h, the window below): lane l decodes
// the symbol that WOULD start l bits behind the current position (as a DC symbol and as an AC symbol of the current MCU position:
// two lookups per lane, all lanes at once), and the walk itself is scalar -- read the entry at the current offset, add its length,
// track zigzag position and MCU position -- about a dozen scalar instructions per symbol instead of fifty vector ones.  Same state
// evolution as decode_subsequence<false>, symbol by symbol (a pair step there is two steps here).  Streams with restart intervals
// keep the lane-parallel walk.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t lane_read(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }

// row: the wave's row buffer (word j of the subsequence at row[j * kTailSlots], kTailRowWords of them), staged by the caller
// The window of cooperative_subsequence (huffman_gpu_core.h) on a wave: lane l holds the entries for bit offset l.
// row: the wave's row buffer (word j of the subsequence at row[j * kTailSlots], kTailRowWords of them), staged by the caller.
struct CoopWindow {
    uint32_t pool;
    const HJ_LDS uint32_t* tsel;
    const HJ_LDS uint32_t* row;
    uint32_t row_bit0, lane;
    uint32_t edc, eac;
    __device__ __forceinline__ uint32_t tables(uint32_t k) const { return tsel[k]; }
    __device__ __forceinline__ uint32_t entry(uint32_t table, uint32_t w) const
    {
        uint32_t e = *(const HJ_LDS uint16_t*)(uintptr_t)(pool + table + ((w >> (31 - kHuffFastBits)) & ((2u << kHuffFastBits) - 2)));
        if ((e >> 9) == kZadvLong) e = *(const HJ_LDS uint16_t*)(uintptr_t)(pool + ((e & 0x1FFu) << 7) + ((w >> 15) & ((2u << kHuffSubBits) - 2)));
        return e;
    }
    __device__ __forceinline__ void open(uint32_t pos, uint32_t ts)
    {
        const uint32_t b = pos + lane - row_bit0, bit = b & 31u;  // the 32 bits from bit pos + lane on
        const uint32_t w0 = row[(b >> 5) * kTailSlots], w1 = row[((b >> 5) + 1) * kTailSlots];
        const uint32_t w = bit ? __builtin_amdgcn_alignbit(w0, w1, 32u - bit) : w0;
        edc = entry(ts & 0xFFFFu, w);
        eac = entry(ts >> 16, w);
    }
    __device__ __forceinline__ uint32_t dc(uint32_t rel) const { return lane_read(edc, rel); }
    __device__ __forceinline__ uint32_t ac(uint32_t rel) const { return lane_read(eac, rel); }
};

// Round budget.  A 1080p photograph at q90 needs 11 rounds in the tail and 3 in a ripple launch; noise at q98 -- long codes, few
// symbols per subsequence, slow to fall into step -- 125 and 40 (tests/devtools/rounds_by_quality.py).  A chain that is still
// going after 192 subsequences (a fifth of a megabit without meeting the true trajectory) sits on a periodic stream and would
// walk through the rest of the image, 28 us per subsequence: the image is given up (the host decoder takes it).
constexpr int kTailRoundBudget = 192, kRippleRoundBudget = 192;
#ifndef HJ_COOP_CHAINS
#define HJ_COOP_CHAINS 2
#endif
constexpr int kCoopChains = HJ_COOP_CHAINS;  // chains a round may have left for the wave to take them one by one (0: never)

template <bool RIPPLE>
__device__ __forceinline__ void tail_group(TailWave& ws, TailEnv env, HuffImage& im, const HuffUnit u, uint32_t ui, int lane,
                                           unsigned long long* __restrict__ states, unsigned long long* __restrict__ incoming,
                                           unsigned int* __restrict__ counters, const uint16_t* __restrict__ tail_tasks, uint32_t pending,
                                           uint32_t pass_id)
{
    const HuffGeom geom = make_geom(im);
    const uint32_t nsub = (geom.total_bits + kSubseqBits - 1) / kSubseqBits;
    if (u.first >= nsub) return;
    unsigned long long* gstate = states + im.first_subseq;
    const int n_rows = (int)min((uint32_t)kSyncThreads, nsub - u.first + 1);
    unsigned long long entering;  // the state the group is entered with ([0] of its rows)
    if (RIPPLE) {
        entering = state_in_front(states, im, u);
        if (lane == 0) {
            incoming[ui] = entering;
            ws.list[0][0] = 1;
        }
    } else {
        entering = incoming[ui];
        for (uint32_t i = lane; i < pending; i += 64) ws.list[0][i] = tail_tasks[(size_t)ui * kSyncThreads + i];
    }
    if (lane == 0) ws.count[0] = pending;
    wave_sync();
    const bool rst = im.restart_interval != 0;
    bool out_moved = false;  // the group's last end state changed
    const HJ_GLOBAL uint32_t* g = (const HJ_GLOBAL uint32_t*)im.stream;
    const uint32_t gwords = im.stream_words;
    env.row_base = (uint32_t)(uintptr_t)(HJ_LDS uint32_t*)&ws.rows[lane & (kTailSlots - 1)];  // word k of slot s at index k * kTailSlots + s
    int rounds = 0, cur = 0;
    bool unfinished = false;
    for (int round = 0;; round++) {
        const uint32_t n = ws.count[cur];
        if (n == 0) break;
        if (round >= (RIPPLE ? kRippleRoundBudget : kTailRoundBudget)) {
            unfinished = true;
            break;
        }
        rounds++;
        wave_sync();  // every lane has read the count before it is reused
        if (lane == 0) ws.count[cur ^ 1] = 0;
        wave_sync();
        if (kCoopChains > 0 && n <= (uint32_t)kCoopChains && !rst) {
            // the few chains that are left, one after the other, each followed to its end by the whole wave (cooperative_subsequence)
            CoopWindow win;
            win.pool = env.pool;
            win.tsel = env.tsel;
            win.row = (const HJ_LDS uint32_t*)&ws.rows[0];
            win.lane = (uint32_t)lane;
            win.edc = win.eac = 0;
            const uint32_t changes = cooperative_table_changes(win, geom.blocks_per_mcu);
            for (uint32_t q = 0; q < n && !unfinished; q++) {
                for (int task = (int)uni(ws.list[cur][q]);; task++) {
                    if (rounds++ >= (RIPPLE ? kRippleRoundBudget : kTailRoundBudget)) {
                        unfinished = true;
                        break;
                    }
                    const unsigned long long before =
                        task == 1 ? entering : __hip_atomic_load(&gstate[u.first - 2 + task], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long old = __hip_atomic_load(&gstate[u.first - 1 + task], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t word0 = (u.first - 1 + task) * kSubseqWords;
                    wave_sync();  // the row buffer's last reader is done
                    if (lane < kTailRowWords) {
                        const uint32_t d = word0 + lane;
                        ws.rows[lane * kTailSlots] = d < gwords ? __builtin_bswap32(g[d]) : ~0u;
                    }
                    wave_sync();
                    const SubseqState p = unpack_state(before);
                    win.row_bit0 = word0 * 32u;
                    const unsigned long long now =
                        pack_state(cooperative_subsequence(geom, win, changes, p.end_bit, (u.first + task) * kSubseqBits, p.zk & 255u, p.zk >> 8));
                    const bool moved = ((now ^ old) & kSyncMask) != 0;
                    if (lane == 0) __hip_atomic_store(&gstate[u.first - 1 + task], now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    wave_sync();
                    if (!moved) break;
                    if (task + 1 >= n_rows) {
                        out_moved = true;
                        break;
                    }
                }
            }
            break;  // every chain has been followed to its end (or the budget is spent)
        }
        for (uint32_t base = 0; base < n; base += kTailSlots) {
            const bool busy = lane < kTailSlots && base + lane < n;
            const int task = busy ? (int)ws.list[cur][base + lane] : 1;
            unsigned long long now = 0;
            bool moved = false;
            if (busy) {
                // Jacobi: every lane of the iteration starts from the end state its predecessor had BEFORE this iteration
                const unsigned long long before =
                    task == 1 ? entering : __hip_atomic_load(&gstate[u.first - 2 + task], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long old = __hip_atomic_load(&gstate[u.first - 1 + task], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // stage the row: 32 words + the reader's look-ahead, straight from the destuffed stream
                env.word0 = (u.first - 1 + task) * kSubseqWords;
                HJ_LDS uint32_t* row = (HJ_LDS uint32_t*)(uintptr_t)env.row_base;
#pragma unroll
                for (int i = 0; i < kTailRowWords / 4; i++) {
                    const uint32_t d = env.word0 + 4 * i;
                    const u32x4 x = *(const HJ_GLOBAL u32x4*)(g + min(d, (gwords - 1) & ~3u));
                    const bool ok = d < gwords;
                    row[(4 * i + 0) * kTailSlots] = ok ? __builtin_bswap32(x.x) : ~0u;
                    row[(4 * i + 1) * kTailSlots] = ok ? __builtin_bswap32(x.y) : ~0u;
                    row[(4 * i + 2) * kTailSlots] = ok ? __builtin_bswap32(x.z) : ~0u;
                    row[(4 * i + 3) * kTailSlots] = ok ? __builtin_bswap32(x.w) : ~0u;
                }
                const SubseqState p = unpack_state(before);
                now = pack_state(walk_subsequence(rst, geom, env, p.end_bit, (u.first + task) * kSubseqBits, p.zk & 255, p.zk >> 8, boundary0_of(im, u.first - 1 + task)));
                moved = ((now ^ old) & kSyncMask) != 0;
            }
            wave_sync();  // all start states have been read
            if (busy) __hip_atomic_store(&gstate[u.first - 1 + task], now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out_moved |= busy && moved && task + 1 == n_rows;
            const bool more = busy && moved && task + 1 < n_rows;
            const unsigned long long mask = __ballot(more);
            const uint32_t have = ws.count[cur ^ 1];
            if (more) ws.list[cur ^ 1][have + __popcll(mask & ((1ull << lane) - 1))] = (uint16_t)(task + 1);
            wave_sync();
            if (lane == 0) ws.count[cur ^ 1] = have + (uint32_t)__popcll(mask);
            wave_sync();
        }
        cur ^= 1;
    }
    const bool any_out = __ballot(out_moved) != 0;
    if (lane == 0) {
        atomicAdd(counters + (RIPPLE ? 4 : 6), (unsigned)rounds);
        atomicMax(counters + (RIPPLE ? 5 : 7), (unsigned)rounds);
        if (unfinished) {
            im.gave_up = 1;  // benign race: every writer stores the same value
        } else if (RIPPLE && any_out) {
            atomicAdd(counters + 0, 1u);
            im.moved_pass = pass_id;  // (same)
        }
    }
}

template <bool RIPPLE>
__global__ __launch_bounds__(kTailThreads) void huff_tail_kernel(HuffImage* __restrict__ images, const HuffUnit* __restrict__ units, int nunits,
                                                                 unsigned long long* __restrict__ states, unsigned long long* __restrict__ incoming,
                                                                 unsigned int* __restrict__ counters, const uint16_t* __restrict__ tail_tasks,
                                                                 const uint32_t* __restrict__ tail_count, uint32_t pass_id)
{
    __shared__ TailShared sh;
    extern __shared__ uint16_t dyn_pool[];
    HJ_LDS uint16_t* pool = (HJ_LDS uint16_t*)dyn_pool;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t ui0 = blockIdx.x * kTailWaves, ui = ui0 + wave;
    const bool valid = ui < (uint32_t)nunits;
    const HuffUnit u = units[valid ? ui : ui0];
    // what every group of the workgroup has to do (uniform: every lane looks at all of them)
    uint32_t todo[kTailWaves];
#pragma unroll
    for (int q = 0; q < kTailWaves; q++) {
        todo[q] = 0;
        if (ui0 + q < (uint32_t)nunits) {
            if (RIPPLE) {
                const HuffUnit uq = units[ui0 + q];
                const HuffImage& iq = images[uq.image];
                const uint32_t nsub = (iq.total_bits + kSubseqBits - 1) / kSubseqBits;
                todo[q] = (uq.first < nsub && iq.gave_up == 0 && state_in_front(states, iq, uq) != incoming[ui0 + q]) ? 1u : 0u;
            } else {
                todo[q] = tail_count[ui0 + q];
            }
        }
    }
    uint32_t pending = 0;
#pragma unroll
    for (int q = 0; q < kTailWaves; q++) pending = q == wave ? todo[q] : pending;
    // the groups of a workgroup usually belong to one image; where an image ends inside it, its tables are staged in turn
#pragma unroll 1
    for (int s = 0; s < kTailWaves; s++) {
        if (ui0 + s >= (uint32_t)nunits) break;  // uniform
        const uint32_t image = units[ui0 + s].image;
        if (s > 0 && image == units[ui0 + s - 1].image) continue;  // uniform: staged with the slot before
        uint32_t work = 0;
#pragma unroll
        for (int q = 0; q < kTailWaves; q++)
            if (ui0 + q < (uint32_t)nunits && units[ui0 + q].image == image) work |= todo[q];
        if (work == 0) continue;  // uniform: nothing left in this image's groups
        HuffImage& im = images[image];
        __syncthreads();  // the tables of the image before are no longer in use
        stage_pool<kTailThreads>(pool, im);
        stage_constants(sh.tsel, nullptr, nullptr, im, false);
        __syncthreads();
        if (valid && u.image == image && pending != 0) {
            TailEnv env;
            env.row_base = 0;
            env.word0 = 0;
            env.pool = (uint32_t)(uintptr_t)pool;
            env.tsel = (const HJ_LDS uint32_t*)sh.tsel;
            env.boundaries = im.boundaries;
            env.num_boundaries = im.num_boundaries;
            tail_group<RIPPLE>(sh.wave[wave], env, im, u, ui, lane, states, incoming, counters, tail_tasks, pending, pass_id);
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
    const uint32_t n = (im.total_bits + kSubseqBits - 1) / kSubseqBits;
    const uint32_t per = (n + kThreads - 1) / kThreads;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    constexpr int kBatch = 8;  // independent loads in flight per lane
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i += kBatch) {
        uint32_t d[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; k++) d[k] = i + k < hi ? (uint32_t)(st[i + k] >> 48) : 0u;
#pragma unroll
        for (int k = 0; k < kBatch; k++) sum += d[k];
    }
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
    if (threadIdx.x == kThreads - 1) {
        im.decoded_blocks = s_sum[kThreads - 1];
        if (s_sum[kThreads - 1] < im.total_blocks) im.status = 2;  // the stream ends before the last block
    }
}

// ---- write pass, step 1: where the blocks start -----------------------------------------------------------------------------
// Same workgroup shape as the sync kernel (lane 0 idles): every lane walks its subsequence from the converged start state
// and records the bit position of each block that starts inside it.
__global__ __launch_bounds__(kSyncThreads) void huff_pos_kernel(HuffImage* __restrict__ images, const HuffUnit* __restrict__ units,
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
    stage_rows<kSyncThreads>(sh.stream, im, (int)u.first - 1);
    stage_pool<kSyncThreads>(pool, im);
    stage_constants(sh.tsel, nullptr, nullptr, im, false);
    __syncthreads();
    const int t = threadIdx.x;
    const uint32_t j = u.first - 1 + t;
    if (t == 0 || j >= nsub) return;
    DevEnv env = make_env(sh, pool, u.first, geom);
    env.set_row(t);
    const bool rst = im.restart_interval != 0;
    const unsigned long long* st = states + im.first_subseq;
    uint32_t begin = 0;
    int z = 0, k = 0;
    if (j > 0) {
        const SubseqState p = unpack_state(st[j - 1]);
        begin = p.end_bit;
        z = p.zk & 255;
        k = p.zk >> 8;
    }
    HJ_GLOBAL uint32_t* out = (HJ_GLOBAL uint32_t*)im.block_pos;
    const uint32_t total_blocks = im.total_blocks;
    auto rec = [&](uint32_t block, uint32_t pos) {
        if (block < total_blocks) out[block] = pos;
    };
    if (rst) {
        uint32_t fault = 0;
        position_subsequence<true>(geom, env, begin, (j + 1) * kSubseqBits, z, k, first_block[im.first_subseq + j], rec, boundary0_of(im, j), &fault);
        if (fault) im.status = 1;  // damaged restart interval: the host decoder takes the image (benign race: same value)
    } else
        position_subsequence<false>(geom, env, begin, (j + 1) * kSubseqBits, z, k, first_block[im.first_subseq + j], rec);
}

// ---- write pass, step 2: one lane per block -----------------------------------------------------------------------------------
// kHuffMcusPerWg consecutive MCUs (scan order) of one image per workgroup, 256 blocks per round.  Every lane decodes
// its block into a 128-byte LDS buffer, and the finished blocks leave as whole 128-byte lines, eight lanes per block --
// each line written once, no memset, no partial-line traffic.  Lanes idle once their block is done: chroma blocks are
// short, luma blocks long; the wave runs as long as its longest block.
constexpr int kBThreads = kHuffBlocksPerWg;
// Words of the bitstream a workgroup may stage in LDS.  0 (shipped): the lanes read the stream through the vector cache instead
// -- a workgroup's span is ~7 KB and every word is needed by one or two neighbouring lanes, the prefetch in the bit reader covers
// the longer latency, and without the 16 KB stage three workgroups fit a CU instead of two (A/B on 256 x 1080p: 771 us against
// 959 us with 4096 words staged, 971 us with 2048).
#ifndef HJ_BLOCK_STREAM_WORDS
#define HJ_BLOCK_STREAM_WORDS 0
#endif
constexpr int kBStreamWords = HJ_BLOCK_STREAM_WORDS;
constexpr int kBlockBufBytes = 144;  // 128 + 16: 16-byte aligned buffers whose starts are spread over the banks

struct BlockShared {
    uint32_t stream[kBStreamWords > 0 ? kBStreamWords : 1];
    __attribute__((aligned(16))) uint8_t blocks[kBThreads * kBlockBufBytes];
    int16_t* dst[kBThreads];
    KSlot kslot[10];
    uint32_t tsel[10];
    uint32_t zz[16];  // zigzag permutation, 4 entries per word
    uint32_t span[2]; // first staged word, number of staged words (0: the span does not fit, read from memory)
    int32_t dc_sum[4];  // per component: sum of the DC differences of the workgroup's blocks
    uint32_t kcomp[10]; // component of MCU position k
};

struct BlockEnv {
    uint32_t stream_base, word0, staged;  // LDS byte address of the staged words, image index of the first, how many
    const uint32_t* gstream;
    uint32_t gwords;
    uint32_t pool, buf;
    const HJ_LDS uint32_t* tsel;
    const HJ_LDS KSlot* kslot;
    const HJ_LDS uint8_t* zz;
    static constexpr uint32_t kCursorStep = 1;  // the cursor is the word index
    __device__ __forceinline__ uint32_t cursor(uint32_t i) const { return i; }
    __device__ __forceinline__ uint32_t fetch(uint32_t i) const { return word(i); }
    __device__ __forceinline__ uint32_t word(uint32_t i) const
    {
        if (kBStreamWords > 0) {
            const uint32_t local = i - word0;
            if (local < staged) return *(const HJ_LDS uint32_t*)(uintptr_t)(stream_base + (local << 2));
        }
        // unconditional load from a clamped index, selection afterwards: the request leaves at the top of the decode step and
        // nothing waits for it before the value is consumed at the bottom
        // (byte offset in 32 bits -- a stream is far below 4 GB: SGPR base + VGPR offset addressing)
        const uint32_t x = *(const HJ_GLOBAL uint32_t*)((const HJ_GLOBAL char*)gstream + (min(i, gwords - 1) << 2));
        return i < gwords ? __builtin_bswap32(x) : ~0u;
    }
    __device__ __forceinline__ uint32_t tables(int k) const { return tsel[k]; }
    __device__ __forceinline__ uint32_t lookup1(uint32_t t, uint32_t w) const
    {
        return *(const HJ_LDS uint16_t*)(uintptr_t)(pool + t + ((w >> (31 - kHuffFastBits)) & ((2u << kHuffFastBits) - 2)));
    }
    __device__ __forceinline__ uint32_t lookup2(uint32_t e, uint32_t w) const
    {
        return *(const HJ_LDS uint16_t*)(uintptr_t)(pool + ((e & 0x1FFu) << 7) + ((w >> 15) & ((2u << kHuffSubBits) - 2)));
    }
    __device__ __forceinline__ int16_t* block_ptr(int k, uint32_t mx, uint32_t my) const
    {
        const HJ_LDS KSlot* s = kslot + k;
        int16_t* base = s->base;
        const uint32_t sy = s->stride_y, sx = s->stride_x;
        return base + (size_t)(my * sy + mx * sx) * 64;
    }
    __device__ __forceinline__ int zigzag(int z) const { return zz[z]; }
    __device__ __forceinline__ void put(int index, int value) const { *(HJ_LDS int16_t*)(uintptr_t)(buf + index * 2) = (int16_t)value; }
};

__global__ __launch_bounds__(kBThreads) void huff_blocks_kernel(HuffImage* __restrict__ images, const HuffUnit* __restrict__ units,
                                                                int32_t* __restrict__ group_sums)
{
    __shared__ BlockShared sh;
    extern __shared__ uint16_t dyn_pool[];
    HJ_LDS uint16_t* pool = (HJ_LDS uint16_t*)dyn_pool;
    const HuffUnit u = units[blockIdx.x];  // first = first MCU of the workgroup's kHuffMcusPerWg MCUs
    HuffImage& im = images[u.image];
    const HuffGeom geom = make_geom(im);
    const uint32_t bpm = geom.blocks_per_mcu;
    const uint32_t nblocks = min(im.total_blocks, im.decoded_blocks);  // a truncated stream: the scan kernel has flagged it
    const uint32_t b_first = u.first * bpm;
    const int t = threadIdx.x;
    if (b_first >= nblocks) {
        if (t < 4) group_sums[(size_t)blockIdx.x * 4 + t] = 0;
        return;
    }
    if (t < 4) sh.dc_sum[t] = 0;
    if (t >= 32 && t < 42) sh.kcomp[t - 32] = im.k[t - 32].comp & 3;
    const uint32_t mcus = min((uint32_t)kHuffMcusPerWg, geom.mcus_x * geom.mcus_y - u.first);
    const uint32_t items = mcus * bpm;
    if (t == 0) {
        // the span of the stream these blocks cover: from the first block's word to the start of the block behind the last
        const uint32_t b_end = b_first + items;
        const uint32_t end_bit = b_end < nblocks ? im.block_pos[b_end] : geom.total_bits;
        const uint32_t w_lo = im.block_pos[b_first] >> 5, w_hi = (end_bit >> 5) + 4;
        sh.span[0] = w_lo;
        sh.span[1] = (kBStreamWords > 0 && w_hi - w_lo <= (uint32_t)kBStreamWords) ? w_hi - w_lo : 0u;
    }
    stage_pool<kBThreads>(pool, im);
    stage_constants(sh.tsel, sh.kslot, sh.zz, im, true);
    HJ_LDS u32x4* my_buf = (HJ_LDS u32x4*)&sh.blocks[t * kBlockBufBytes];
#pragma unroll
    for (int i = 0; i < kBlockBufBytes / 16; i++) my_buf[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    const uint32_t w_lo = sh.span[0], staged = sh.span[1];
    {
        const HJ_GLOBAL uint32_t* g = (const HJ_GLOBAL uint32_t*)im.stream;
        const uint32_t gwords = im.stream_words;
        for (uint32_t i = t; i < staged; i += kBThreads) sh.stream[i] = w_lo + i < gwords ? __builtin_bswap32(g[w_lo + i]) : ~0u;
    }
    BlockEnv env;
    env.stream_base = (uint32_t)(uintptr_t)(HJ_LDS uint32_t*)sh.stream;
    env.word0 = w_lo;
    env.staged = staged;
    env.gstream = reinterpret_cast<const uint32_t*>(im.stream);
    env.gwords = im.stream_words;
    env.pool = (uint32_t)(uintptr_t)pool;
    env.buf = (uint32_t)(uintptr_t)(HJ_LDS uint8_t*)&sh.blocks[t * kBlockBufBytes];
    env.tsel = (const HJ_LDS uint32_t*)sh.tsel;
    env.kslot = (const HJ_LDS KSlot*)sh.kslot;
    env.zz = (const HJ_LDS uint8_t*)sh.zz;
    uint32_t err = 0;
    // Work items are ordered by MCU position first: item i is position i / mcus of MCU i % mcus, so that the 256 lanes of
    // a round hold blocks of the same component -- luma blocks run ~4x longer than chroma blocks, and a wave takes as long
    // as its longest block.
    for (uint32_t base = 0; base < items; base += kBThreads) {
        __syncthreads();  // staged stream (first round) / zeroed buffers (later rounds) are in place
        const uint32_t item = base + t;
        int16_t* dst = nullptr;
        if (item < items) {
            const uint32_t k = item / mcus, mcu = u.first + (item - k * mcus);
            const uint32_t b = mcu * bpm + k;
            if (b < nblocks) {
                const uint32_t my = mcu / geom.mcus_x, mx = mcu - my * geom.mcus_x;
                dst = env.block_ptr((int)k, mx, my);
                const uint32_t bpos = im.block_pos[b];
                // restart intervals: the first block of interval j+1 starts exactly at boundary j -- anything else means
                // the trajectory went through a damaged interval (the host decoder takes the image and names the error)
                if (geom.interval_blocks != 0 && k == 0 && b != 0 && b % geom.interval_blocks == 0 &&
                    bpos != load_boundary(geom.boundaries, geom.num_boundaries, b / geom.interval_blocks - 1))
                    err = 1;
                const int dc = decode_block(geom, env, bpos, (int)k, &err);
                ((HJ_GLOBAL int16_t*)geom.dc_diff)[b] = (int16_t)dc;
                atomicAdd(&sh.dc_sum[sh.kcomp[k]], (int)(int16_t)dc);
            }
        }
        sh.dst[t] = dst;
        __syncthreads();
        // eight lanes per block, 16 bytes each: every store instruction writes eight whole lines
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int blk = (t & ~63) + 8 * i + ((t & 63) >> 3), chunk = t & 7;
            int16_t* p = sh.dst[blk];
            if (p) {
                const u32x4 v = *(const HJ_LDS u32x4*)&sh.blocks[blk * kBlockBufBytes + chunk * 16];
#ifdef HJ_BLOCKS_PLAIN_STORES
                *((HJ_GLOBAL u32x4*)p + chunk) = v;
#else
                __builtin_nontemporal_store(v, (HJ_GLOBAL u32x4*)p + chunk);
#endif
            }
        }
        if (base + kBThreads < items) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; i++) my_buf[i] = u32x4{0u, 0u, 0u, 0u};
        }
    }
    if (err) im.status = 1;  // benign race: every writer stores the same value
    __syncthreads();
    if (t < 4) group_sums[(size_t)blockIdx.x * 4 + t] = sh.dc_sum[t];
}

// ---- fused decode (round 3): the DC differences alone ------------------------------------------------------------------------
// When the pixel kernels decode the coefficient blocks themselves (decode_kernels.hip, FUSED builds: Huffman decode into the LDS
// slots the IDCT reads, no coefficient ever reaches HBM), the block pass above is not launched.  What remains of it is this: one
// lane per block reads the block's FIRST symbol -- its DC difference -- at the recorded start position, so that the DC pass can
// integrate the predictors before the pixel kernels run.  Table lookups go to memory through the vector cache (one or two per
// block; staging the tables in LDS would cost more than it saves).  Same work units and group sums as the block pass; also the
// restart-interval check of the block pass (an interval's first block starts exactly at its boundary).
__global__ __launch_bounds__(kThreads) void huff_dcdiff_kernel(HuffImage* __restrict__ images, const HuffUnit* __restrict__ units,
                                                              int32_t* __restrict__ group_sums)
{
    __shared__ int32_t s_sum[4];
    __shared__ uint32_t s_tdc[10], s_comp[10];
    const HuffUnit u = units[blockIdx.x];  // first = first MCU of the workgroup's kHuffMcusPerWg MCUs
    HuffImage& im = images[u.image];
    const HuffGeom geom = make_geom(im);
    const uint32_t bpm = geom.blocks_per_mcu;
    const uint32_t nblocks = min(im.total_blocks, im.decoded_blocks);
    const uint32_t b_first = u.first * bpm;
    const int t = threadIdx.x;
    if (t < 4) s_sum[t] = 0;
    if (t >= 32 && t < 42) {
        s_tdc[t - 32] = im.k[t - 32].tdc;
        s_comp[t - 32] = im.k[t - 32].comp & 3;
    }
    __syncthreads();
    if (b_first < nblocks) {
        const uint32_t mcus = min((uint32_t)kHuffMcusPerWg, geom.mcus_x * geom.mcus_y - u.first);
        const uint32_t items = mcus * bpm;
        const HJ_GLOBAL uint32_t* g = (const HJ_GLOBAL uint32_t*)im.stream;
        const HJ_GLOBAL uint16_t* pool = (const HJ_GLOBAL uint16_t*)im.pool;
        const uint32_t gwords = im.stream_words;
        uint32_t err = 0;
        for (uint32_t item = t; item < items; item += kThreads) {
            const uint32_t b = b_first + item;
            if (b >= nblocks) break;
            const uint32_t k = item % bpm;
            const uint32_t bpos = im.block_pos[b];
            if (geom.interval_blocks != 0 && k == 0 && b != 0 && b % geom.interval_blocks == 0 &&
                bpos != load_boundary(geom.boundaries, geom.num_boundaries, b / geom.interval_blocks - 1))
                err = 1;
            const uint32_t i = bpos >> 5, sft = bpos & 31;
            const uint32_t w0 = i < gwords ? __builtin_bswap32(g[i]) : ~0u, w1 = i + 1 < gwords ? __builtin_bswap32(g[i + 1]) : ~0u;
            const uint32_t w = sft ? (w0 << sft) | (w1 >> (32 - sft)) : w0;
            uint32_t e = pool[s_tdc[k] + (w >> (32 - kHuffFastBits))];
            if ((e >> 9) == kZadvLong) e = pool[((e & 0x1FFu) << 6) + ((w >> (32 - kHuffFastBits - kHuffSubBits)) & ((1u << kHuffSubBits) - 1))];
            const uint32_t total = e & 31u, nb_raw = (e >> 5) & 15u;
            const bool bad = nb_raw >= total;
            const uint32_t nb = bad ? 0u : nb_raw;
            const uint32_t v = nb ? (w << (total - nb)) >> (32 - nb) : 0u;
            const uint32_t m = (1u << nb) - 1u;
            const int dc = (v << 1) <= m ? (int)v - (int)m : (int)v;
            if (bad || bpos + total > geom.total_bits) err = 1;
            ((HJ_GLOBAL int16_t*)geom.dc_diff)[b] = (int16_t)dc;
            atomicAdd(&s_sum[s_comp[k]], (int)(int16_t)dc);
        }
        if (err) im.status = 1;  // benign race: every writer stores the same value
    }
    __syncthreads();
    if (t < 4) group_sums[(size_t)blockIdx.x * 4 + t] = s_sum[t];
}

// ---- DC differences -> DC values ------------------------------------------------------------------------------------------
// Images without restart intervals: one workgroup per block-pass unit (kHuffMcusPerWg MCUs).  The predictor value a group
// starts from is the sum of the group sums in front of it (the block pass left them: a few dozen numbers), so every group
// integrates its own MCUs without waiting for anybody; wave c takes component c and scans its entries 64 at a time.
__global__ __launch_bounds__(kThreads) void huff_dc_group_kernel(const HuffImage* __restrict__ images, const HuffUnit* __restrict__ units,
                                                                 const int32_t* __restrict__ group_sums)
{
    __shared__ int16_t s_diff[kHuffMcusPerWg * 10];
    __shared__ int s_base[4];
    const HuffUnit u = units[blockIdx.x];  // first = first MCU of the group
    const HuffImage& im = images[u.image];
    if (im.restart_interval != 0) return;  // uniform: huff_dc_kernel takes those
    const int t = threadIdx.x;
    const uint32_t bpm = im.blocks_per_mcu;
    const uint32_t total_mcus = im.mcus_x * im.mcus_y;
    const uint32_t mcus = min((uint32_t)kHuffMcusPerWg, total_mcus - u.first);
    const uint32_t n = mcus * bpm;
    const uint32_t g = u.first / kHuffMcusPerWg;  // groups in front of this one: units blockIdx.x - g .. blockIdx.x - 1
    if (t < 4) s_base[t] = 0;
    {
        const HJ_GLOBAL int16_t* diff = (const HJ_GLOBAL int16_t*)im.dc_diff + (size_t)u.first * bpm;
        for (uint32_t i = t; i < n; i += kThreads) s_diff[i] = diff[i];
    }
    __syncthreads();
    {
        const HJ_GLOBAL int32_t* sums = (const HJ_GLOBAL int32_t*)group_sums + (size_t)(blockIdx.x - g) * 4;
        int part = 0;
        for (uint32_t i = t; i < g * 4; i += kThreads) part += sums[i];  // i & 3 = t & 3: a lane stays with one component
        if (part != 0) atomicAdd(&s_base[t & 3], part);
    }
    __syncthreads();
    const int c = t >> 6, lane = t & 63;
    if (c >= (int)im.ncomp) return;
    const uint32_t h = im.comp_h[c], v = im.comp_v[c], bpc = h * v, k0 = im.comp_k0[c];
    const uint32_t bw = im.blocks_w[c], mcus_x = im.mcus_x;
    HJ_GLOBAL int16_t* plane = (HJ_GLOBAL int16_t*)im.dc_plane[c];
    int run = s_base[c];
    const uint32_t nc = mcus * bpc;
    for (uint32_t s0 = 0; s0 < nc; s0 += 64) {
        const uint32_t s = s0 + lane;
        const bool live = s < nc;
        const uint32_t m = s / bpc, jj = s - m * bpc;
        int x = live ? (int)s_diff[m * bpm + k0 + jj] : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (live) {
            const uint32_t mcu = u.first + m;
            const uint32_t my = mcu / mcus_x, mx = mcu - my * mcus_x;
            const uint32_t dy = jj / h, dx = jj - dy * h;
            plane[(my * v + dy) * bw + (mx * h + dx)] = (int16_t)(run + x);
        }
        run += __shfl(x, 63, 64);
    }
}

// Images with restart intervals (the predictor starts over inside the image): one workgroup per (image, component)
// integrates the DC differences in MCU order; DC values go to the component's compact
// DC plane (raster block order), which the IDCT kernels read beside the coefficient blocks.
__global__ __launch_bounds__(kThreads) void huff_dc_kernel(const HuffImage* __restrict__ images, const HuffUnit* __restrict__ units)
{
    __shared__ int s_sum[kThreads];
    __shared__ int s_flag[kThreads];
    const HuffUnit u = units[blockIdx.x];  // first = component
    const HuffImage& im = images[u.image];
    const int c = (int)u.first;
    const uint32_t h = im.comp_h[c], v = im.comp_v[c], bpc = h * v;  // blocks of this component per MCU
    const uint32_t bpm = im.blocks_per_mcu, k0 = im.comp_k0[c];
    const uint32_t mcus = im.mcus_x * im.mcus_y;
    const uint32_t n = mcus * bpc;
    HJ_GLOBAL int16_t* plane = (HJ_GLOBAL int16_t*)im.dc_plane[c];
    const HJ_GLOBAL int16_t* diff = (const HJ_GLOBAL int16_t*)im.dc_diff;
    const uint32_t bw = im.blocks_w[c], mcus_x = im.mcus_x;
    // restart intervals: the predictor starts over every `seg` entries of the component (0xFFFFFFFF: never)
    const uint32_t seg = im.restart_interval ? im.restart_interval * bpc : 0xFFFFFFFFu;
    // entry s of the component (MCU order) sits at diff[(s / bpc) * bpm + k0 + s % bpc]; every lane takes a contiguous run
    const uint32_t per = (n + kThreads - 1) / kThreads;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    auto index_of = [&](uint32_t s) {
        const uint32_t mcu = s / bpc;
        return mcu * bpm + k0 + (s - mcu * bpc);
    };
    constexpr int kBatch = 8;  // independent loads in flight per lane
    // per lane: the running sum at the end of its run (since the last restart inside the run, if any)
    int sum = 0, flag = 0;
    uint32_t until = lo < hi ? (seg == 0xFFFFFFFFu ? 0xFFFFFFFFu : (seg - lo % seg) % seg) : 0xFFFFFFFFu;  // entries until the next restart
    for (uint32_t s = lo; s < hi; s += kBatch) {
        int d[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; i++) d[i] = s + i < hi ? (int)diff[index_of(s + i)] : 0;
#pragma unroll
        for (int i = 0; i < kBatch; i++) {
            if (s + i < hi) {
                if (until == 0) {
                    sum = 0;
                    flag = 1;
                    until = seg;
                }
                until--;
            }
            sum += d[i];
        }
    }
    s_sum[threadIdx.x] = sum;
    s_flag[threadIdx.x] = flag;
    __syncthreads();
    // segmented inclusive scan over the lanes: (a, fa) . (b, fb) = fb ? (b, 1) : (a + b, fa)
    for (int off = 1; off < kThreads; off <<= 1) {
        int a = 0, fa = 0;
        if (threadIdx.x >= (unsigned)off) {
            a = s_sum[threadIdx.x - off];
            fa = s_flag[threadIdx.x - off];
        }
        __syncthreads();
        if (threadIdx.x >= (unsigned)off && !s_flag[threadIdx.x]) {
            s_sum[threadIdx.x] += a;
            s_flag[threadIdx.x] = fa;
        }
        __syncthreads();
    }
    int run = threadIdx.x ? s_sum[threadIdx.x - 1] : 0;
    until = lo < hi ? (seg == 0xFFFFFFFFu ? 0xFFFFFFFFu : (seg - lo % seg) % seg) : 0xFFFFFFFFu;
    for (uint32_t s = lo; s < hi; s += kBatch) {
        int d[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; i++) d[i] = s + i < hi ? (int)diff[index_of(s + i)] : 0;
#pragma unroll
        for (int i = 0; i < kBatch; i++) {
            if (s + i < hi) {
                if (until == 0) {
                    run = 0;
                    until = seg;
                }
                until--;
                run += d[i];
                const uint32_t mcu = (s + i) / bpc, jj = (s + i) - mcu * bpc;
                const uint32_t my = mcu / mcus_x, mx = mcu - my * mcus_x;
                const uint32_t dy = jj / h, dx = jj - dy * h;
                plane[(my * v + dy) * bw + (mx * h + dx)] = (int16_t)run;
            }
        }
    }
}

}  // namespace

int launch_gather_raw(const HuffImage* images, const HuffUnit* chunk_units, int nchunks, void* stream)
{
    if (nchunks <= 0) return 0;
    hipLaunchKernelGGL(gather_raw_kernel, dim3(min(nchunks, kGatherGroups)), dim3(kThreads), 0, (hipStream_t)stream, images, chunk_units, nchunks);
    return (int)hipGetLastError();
}

int launch_destuff(HuffImage* images, const HuffUnit* chunk_units, int nchunks, uint32_t* drops, bool count_on_device, unsigned int* counters,
                   void* stream)
{
    if (nchunks <= 0) return 0;
    if (count_on_device)
        hipLaunchKernelGGL(destuff_count_kernel, dim3(nchunks), dim3(kThreads), 0, (hipStream_t)stream, images, chunk_units, drops);
    hipLaunchKernelGGL(destuff_compact_kernel, dim3(nchunks), dim3(kThreads), 0, (hipStream_t)stream, images, chunk_units, drops, counters);
    return (int)hipGetLastError();
}

// HIPJPEG_RIPPLE_IN_SYNC=1: the second and later launches use the sync kernel's ripple branch as in round 1 (A/B aid)
static bool ripple_in_tail_kernel()
{
    static const bool v = getenv("HIPJPEG_RIPPLE_IN_SYNC") == nullptr;
    return v;
}

int launch_huff_sync(HuffImage* images, const HuffUnit* units, int nunits, unsigned long long* states, unsigned long long* incoming,
                     unsigned int* changed, int first_pass, int max_rounds, uint16_t* tail_tasks, uint32_t* tail_count, unsigned pool_bytes, void* stream,
                     unsigned pass_id)
{
    if (nunits <= 0) return 0;
    const dim3 tail_grid((nunits + kTailWaves - 1) / kTailWaves);
    if (!first_pass && ripple_in_tail_kernel()) {
        // corrections across group borders: the tail kernel's workgroups (little LDS, a row staged per decode) instead of the
        // sync kernel's, which would stage 255 rows to decode one or two
        hipLaunchKernelGGL(huff_tail_kernel<true>, tail_grid, dim3(kTailThreads), pool_bytes, (hipStream_t)stream, images, units, nunits, states, incoming,
                           changed, nullptr, nullptr, pass_id);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(huff_sync_kernel, dim3(nunits), dim3(kSyncThreads), pool_bytes, (hipStream_t)stream, images, units, states, incoming, changed,
                       first_pass, max_rounds, tail_tasks, tail_count);
    if (tail_count)
        hipLaunchKernelGGL(huff_tail_kernel<false>, tail_grid, dim3(kTailThreads), pool_bytes, (hipStream_t)stream, images, units, nunits, states, incoming,
                           changed, tail_tasks, tail_count, 0u);
    return (int)hipGetLastError();
}

int launch_huff_scan(HuffImage* images, const uint32_t* image_list, int nimages, const unsigned long long* states, uint32_t* first_block, void* stream)
{
    if (nimages <= 0) return 0;
    hipLaunchKernelGGL(huff_scan_kernel, dim3(nimages), dim3(kThreads), 0, (hipStream_t)stream, images, image_list, states, first_block);
    return (int)hipGetLastError();
}

int launch_huff_write(HuffImage* images, const HuffUnit* sync_units, int nsync_units, const HuffUnit* block_units, int nblock_units,
                      const unsigned long long* states, const uint32_t* first_block, int32_t* group_sums, unsigned pool_bytes, void* stream,
                      bool dc_only)
{
    if (nsync_units <= 0) return 0;
    hipLaunchKernelGGL(huff_pos_kernel, dim3(nsync_units), dim3(kSyncThreads), pool_bytes, (hipStream_t)stream, images, sync_units, states, first_block);
    if (nblock_units > 0) {
        if (dc_only)
            hipLaunchKernelGGL(huff_dcdiff_kernel, dim3(nblock_units), dim3(kThreads), 0, (hipStream_t)stream, images, block_units, group_sums);
        else
            hipLaunchKernelGGL(huff_blocks_kernel, dim3(nblock_units), dim3(kBThreads), pool_bytes, (hipStream_t)stream, images, block_units, group_sums);
    }
    return (int)hipGetLastError();
}

int launch_huff_dc(const HuffImage* images, const HuffUnit* rst_units, int nrst_units, const HuffUnit* block_units, int nblock_units,
                   const int32_t* group_sums, void* stream)
{
    if (nblock_units > 0)
        hipLaunchKernelGGL(huff_dc_group_kernel, dim3(nblock_units), dim3(kThreads), 0, (hipStream_t)stream, images, block_units, group_sums);
    if (nrst_units > 0) hipLaunchKernelGGL(huff_dc_kernel, dim3(nrst_units), dim3(kThreads), 0, (hipStream_t)stream, images, rst_units);
    return (int)hipGetLastError();
}

}  // namespace hipjpeg
