// decode_kernels.hip -- gfx950 kernels of the JPEG decode device stage
// (what nvjpegDecodeJpegDevice does for the reference: extensions/nvjpeg/cuda_decoder.cpp:546-549).
//
// Mapping to the machine (wave64, 256 CUs, 160 KB LDS/CU, no MFMA -- this is integer butterfly + byte work).
// Profiling the first version (one lane per block, ~200 VGPRs, 2 waves/SIMD) showed it VALU-issue bound at ~4 cycles
// per instruction even with every memory stream ablated, so the design goal here is register economy -> occupancy:
//   * TWO LANES own one 8x8 block (lane pair 2k, 2k+1; p = lane & 1).  Lane p runs the column pass of columns
//     4p..4p+3, the pair swaps half of the (int16, packed) workspace with one v_cndmask_b32_dpp per register, and lane p
//     runs the row pass of four rows (lane 0 rows 0..3, lane 1 rows 7..4).
//   * the arithmetic is that of libjpeg-turbo's SIMD jpeg_idct_islow (the routine the reference's CPU path runs): values
//     live as int16 pairs, products are v_dot2_i32_i16 (= pmaddwd), the workspace is packed with saturation
//     (v_cvt_pk_i16_i32 = packssdw) -- see idct1d_pk.
//   * coefficient blocks are fetched from HBM with fully coalesced 16 B/lane loads (a wave reads 4 KB contiguous), staged
//     through LDS at a 144-byte block stride, and each lane pulls its four 16-byte column chunks back out.  The host
//     stores blocks column-major so one chunk is one IDCT column.
//   * idct_plane_kernel writes component planes; luma_color_kernel fuses the luma IDCT with chroma upsampling
//     (libjpeg "fancy" triangle filters), YCbCr->RGB and the output store; interleaved RGB rows are staged in LDS and
//     leave the wave as fully coalesced 16-byte stores.
//   * work is described by WorkUnit tables so a batch of different-shaped images is ONE launch per kernel.
//
// Arithmetic is the integer arithmetic of libjpeg-turbo's jidctint-{sse2,avx2}.asm / jdsample.c / jdcolor.c, restated; results
// are compared bit-for-bit with the CPU oracle and with vectors from the library itself in tests/.
#include <hip/hip_runtime.h>

#include "decode_kernels.h"
#include "device_layout.h"
#include "huffman_gpu_core.h"

namespace hipjpeg {

namespace {

constexpr int kThreads = 256;
#ifndef HJ_MIN_WAVES
#define HJ_MIN_WAVES 4
#endif
#ifndef HJ_MIN_WAVES_LUMA
#define HJ_MIN_WAVES_LUMA 5  // measured: 4 -> 1.017 ms, 5 -> 0.966 ms, 6 -> 1.18 ms (spills) per 256 x 1080p, generic flavour
#endif
constexpr int kBlocksPerWave = 32;     // two lanes per block
constexpr int kLdsBlockStride = 144;   // 128 B of coefficients + 16 B pad -> conflict-free ds_read_b128 at this lane stride

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
// Explicitly GLOBAL pointers for the hot loads and stores: through the generic pointers of the descriptors hipcc emits FLAT
// instructions (64-bit address arithmetic per access, and they count against the LDS counter as well).
#define HJ_GLOBAL __attribute__((address_space(1)))

constexpr int F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633;
constexpr int F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;


// a * k + c in ONE instruction.  hipcc rewrites __mul24 of provably small operands into a plain 32-bit multiply and then emits
// v_mul_lo_u32 + v_add (seen in the ISA: 130 of them per lane in the colour stage); there is no 32-bit integer mad, so the
// colour arithmetic pins the 24-bit one.  (v_mul_lo_u32 itself is not slow on gfx950 -- tools/valu_rate.hip -- the saving is
// the fused add.)  |a| < 2^8 and |k| < 2^17 here: exact.
__device__ __forceinline__ int mad24(int a, int k, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c));
    return r;
}

// ---- two int16 per register (colour stage)
using i16x2 = __attribute__((ext_vector_type(2))) short;
__device__ __forceinline__ unsigned lo_pair(int a, int b) { return __builtin_amdgcn_perm((unsigned)b, (unsigned)a, 0x05040100u); }  // a[15:0] | b[15:0] << 16
__device__ __forceinline__ unsigned hi_pair(int a, int b) { return __builtin_amdgcn_perm((unsigned)b, (unsigned)a, 0x07060302u); }  // a[31:16] | b[31:16] << 16
__device__ __forceinline__ unsigned pk_add16(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, (i16x2)(__builtin_bit_cast(i16x2, a) + __builtin_bit_cast(i16x2, b)));
}
__device__ __forceinline__ unsigned clamp_s8_pair(unsigned a)
{
    const i16x2 lo = {-128, -128}, hi = {127, 127};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(i16x2, a), lo), hi));
}
// four int16 (two registers) -> four bytes, each saturated to 0..255: a.lo a.hi b.lo b.hi
__device__ __forceinline__ unsigned sat_pk4(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_sat_pk_u8_i16_e32 %0, %1" : "=v"(r) : "v"(a));
    asm("v_sat_pk_u8_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(r) : "v"(b));
    return r;
}

__device__ __forceinline__ unsigned pack4(int a, int b, int c, int d) { return (unsigned)a | ((unsigned)b << 8) | ((unsigned)c << 16) | ((unsigned)d << 24); }

// value held by the other lane of the pair (quad_perm [1,0,3,2])
__device__ __forceinline__ int pair_swap(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS operations of one wave execute in order; only the compiler has to be kept from reordering across the hand-off
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- FUSED builds (round 3): the coefficient blocks come straight from the bitstream ------------------------------------------
// For images whose entropy stage runs on the GPU the block pass of gpu_huffman.hip used to write every coefficient block to HBM
// (1.6 GB per 256 x 1080p) for these kernels to read back.  The FUSED builds skip that round trip: where the plain builds stage a
// wave's 32 blocks from HBM into its LDS slots, here 32 lanes of the wave Huffman-decode them into the very same slots
// (huffman_gpu_core.h decode_block: one lane per block from the start position the position pass recorded; bitstream through the
// vector cache, lookup tables in LDS), and everything behind the slots -- the two-lane IDCT, upsampling, colour, stores -- is the
// same code.  DC values come from the compact DC planes as for every GPU-decoded image (the DC pass has run before).
#define HJ_LDS __attribute__((address_space(3)))
struct FusedShared {
    uint32_t tsel[10];
    uint32_t zz[16];
};
struct FusedEnv {
    const uint32_t* gstream;
    uint32_t gwords;
    uint32_t pool, buf;  // LDS byte addresses: lookup tables, this lane's block slot
    const HJ_LDS uint32_t* tsel;
    const HJ_LDS uint8_t* zz;
    static constexpr uint32_t kCursorStep = 1;  // the cursor is the word index
    __device__ __forceinline__ uint32_t cursor(uint32_t i) const { return i; }
    __device__ __forceinline__ uint32_t fetch(uint32_t i) const
    {
        // unconditional load from a clamped index, selection afterwards (gpu_huffman.hip BlockEnv::word)
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
    __device__ __forceinline__ int zigzag(int z) const { return zz[z]; }
    __device__ __forceinline__ void put(int index, int value) const { *(HJ_LDS int16_t*)(uintptr_t)(buf + index * 2) = (int16_t)value; }
};

// lookup tables, per-MCU-position table offsets and the zigzag permutation -> LDS; ends with a workgroup barrier
__device__ __forceinline__ void fused_stage(const HuffImage& hi, HJ_LDS uint16_t* pool, FusedShared& fs)
{
    const int t = threadIdx.x;
    const HJ_GLOBAL u32x4* gp = (const HJ_GLOBAL u32x4*)hi.pool;
    const uint32_t npool = hi.pool_words >> 3;  // pool_words is a multiple of 64
    for (uint32_t i = t; i < npool; i += kThreads) {
        const u32x4 x = gp[i];
        HJ_LDS uint32_t* lp = reinterpret_cast<HJ_LDS uint32_t*>(pool) + i * 4;
        lp[0] = x.x;
        lp[1] = x.y;
        lp[2] = x.z;
        lp[3] = x.w;
    }
    if (t < 10) fs.tsel[t] = ((uint32_t)hi.k[t].tdc * 2) | ((uint32_t)hi.k[t].tac * 2 << 16);  // byte offsets
    if (t >= 64 && t < 80) {
        constexpr uint8_t zz[64] = HJ_ZIGZAG_DEVICE_TABLE;
        const int i = (t - 64) * 4;
        fs.zz[t - 64] = (uint32_t)zz[i] | ((uint32_t)zz[i + 1] << 8) | ((uint32_t)zz[i + 2] << 16) | ((uint32_t)zz[i + 3] << 24);
    }
    __syncthreads();
}

// Decodes the wave's 32 blocks into its LDS slots: lane j < 32 takes slot j = block (row, col) of component c's grid (have = the
// slot holds a block that is needed).  Slots are zeroed first (decode_block stores only the coefficients that are there); the DC
// position stays zero -- the caller patches the DC value in from the DC plane.
__device__ __forceinline__ void fused_decode_slots(HuffImage& hi, HJ_LDS uint16_t* pool, FusedShared& fs, int c, bool have, int row, int col, char* lds_wave,
                                                   int lane)
{
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int g = k * 64 + lane;  // 288 16-byte pieces
        if (g < kBlocksPerWave * kLdsBlockStride / 16) *reinterpret_cast<u32x4*>(lds_wave + g * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    wave_lds_fence();
    if (lane < kBlocksPerWave && have) {
        const uint32_t h = hi.comp_h[c], v = hi.comp_v[c];
        const uint32_t my = (uint32_t)row / v, mx = (uint32_t)col / h;
        const uint32_t k = hi.comp_k0[c] + ((uint32_t)row - my * v) * h + ((uint32_t)col - mx * h);
        const uint32_t b = (my * hi.mcus_x + mx) * hi.blocks_per_mcu + k;
        if (b < min(hi.total_blocks, hi.decoded_blocks)) {  // (a truncated stream: the scan kernel has flagged the image)
            const HuffGeom geom = make_geom(hi);
            FusedEnv env;
            env.gstream = reinterpret_cast<const uint32_t*>(hi.stream);
            env.gwords = hi.stream_words;
            env.pool = (uint32_t)(uintptr_t)pool;
            env.buf = (uint32_t)(uintptr_t)(HJ_LDS char*)(lds_wave + lane * kLdsBlockStride);
            env.tsel = (const HJ_LDS uint32_t*)fs.tsel;
            env.zz = (const HJ_LDS uint8_t*)fs.zz;
            uint32_t err = 0;
            (void)decode_block(geom, env, ((const HJ_GLOBAL uint32_t*)hi.block_pos)[b], (int)k, &err);
            if (err) hi.status = 1;  // benign race: every writer stores the same value
        }
    }
    wave_lds_fence();
}

// each lane's four 16-byte column chunks out of the wave's slots, DC patched in by lane 0 of the pair
__device__ __forceinline__ void read_half_block(const char* lds_wave, int lane, unsigned dc, bool dc_apart, u32x4 (&cols)[4])
{
    const char* mine = lds_wave + (lane >> 1) * kLdsBlockStride + (lane & 1) * 64;
#pragma unroll
    for (int j = 0; j < 4; j++) cols[j] = *reinterpret_cast<const u32x4*>(mine + j * 16);
    if (dc_apart && !(lane & 1)) cols[0].x = (cols[0].x & 0xFFFF0000u) | dc;
}

// ---- zero-run-compressed staging (host entropy stage, entropy_decode.h) -------------------------------------------------------
// The picture came over PCIe as a SPARSE stream: per block a record [n][DC lo hi] n x {position, value lo, value hi}, found through
// a table of byte offsets.  The lane pair of a block expands its record into the block's LDS slot (zeroed first): lane 0 the DC value and
// the even entries, lane 1 the odd ones -- one unaligned 4-byte load and one 2-byte LDS store per coefficient, ~6 per lane for a q90
// photograph.  have = this pair's slot holds a block of the grid; bidx = its raster index.
__device__ __forceinline__ void sparse_expand_slots(const DecodeComponent& cd, bool have, unsigned bidx, char* lds_wave, int lane)
{
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int g = k * 64 + lane;  // 288 16-byte pieces
        if (g < kBlocksPerWave * kLdsBlockStride / 16) *reinterpret_cast<u32x4*>(lds_wave + g * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    wave_lds_fence();
    if (have) {
        const unsigned off = ((const HJ_GLOBAL unsigned*)cd.block_off)[bidx];
        if (off) {
            const HJ_GLOBAL unsigned char* rec = (const HJ_GLOBAL unsigned char*)cd.coef + off;
            typedef unsigned __attribute__((aligned(1))) u32_unaligned;
            const unsigned head = *(const HJ_GLOBAL u32_unaligned*)rec;  // n, DC lo, DC hi, (first entry's position)
            const int n = (int)(head & 255u), p = lane & 1;
            HJ_LDS unsigned short* slot = (HJ_LDS unsigned short*)(lds_wave + (lane >> 1) * kLdsBlockStride);
            if (!p) slot[0] = (unsigned short)(head >> 8);
            for (int e = p; e < n; e += 2) {
                const unsigned v = *(const HJ_GLOBAL u32_unaligned*)(rec + 3 + 3 * e);  // position, value lo, value hi, (one byte of the next entry)
                slot[v & 63u] = (unsigned short)(v >> 8);
            }
        }
    }
    wave_lds_fence();
}

// Coalesced HBM -> LDS staging of the 32 blocks a wave owns (4 KB contiguous), then each lane reads back the four
// 16-byte column chunks of its half block: columns 4p .. 4p+3 of block lane>>1.
// The DC coefficient comes from cd.dc (see DecodeComponent): lane 0 of each pair patches it into column 0, row 0.
__device__ __forceinline__ void fetch_half_block(const DecodeComponent& cd, int wave_first_block, int block_limit, char* lds_wave, int lane,
                                                 u32x4 (&cols)[4])
{
    const int16_t* __restrict__ comp_coef = cd.coef;
    if (cd.block_off) {  // (wave-uniform: a property of the picture)
        const int mine = wave_first_block + (lane >> 1);
        sparse_expand_slots(cd, mine < block_limit, (unsigned)mine, lds_wave, lane);
        read_half_block(lds_wave, lane, 0u, false, cols);
        return;
    }
    const u32x4* src = reinterpret_cast<const u32x4*>(comp_coef) + (size_t)wave_first_block * 8;
    const int nchunks = min(kBlocksPerWave, block_limit - wave_first_block) * 8;  // valid 16-byte chunks (may be <= 0)
    const int my_block = wave_first_block + (lane >> 1);
    const bool dc_apart = cd.dc_stride != 64;  // wave-uniform: host-decoded images carry the DC inside the block already
    unsigned dc = 0;
    if (dc_apart && my_block < block_limit) dc = ((const HJ_GLOBAL unsigned short*)cd.dc)[my_block];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int g = k * 64 + lane;
        u32x4 v = {0u, 0u, 0u, 0u};
#ifdef HJ_ABLATE_COEF
        if (g < nchunks) v = u32x4{(unsigned)g, 1u, 0u, 0u};
#else
#ifdef HJ_DEC_PLAIN_LOADS
        if (g < nchunks) v = *((const HJ_GLOBAL u32x4*)src + g);
#else
        if (g < nchunks) v = __builtin_nontemporal_load((const HJ_GLOBAL u32x4*)src + g);
#endif
#endif
        *reinterpret_cast<u32x4*>(lds_wave + (g >> 3) * kLdsBlockStride + (g & 7) * 16) = v;
    }
    wave_lds_fence();
    const char* mine = lds_wave + (lane >> 1) * kLdsBlockStride + (lane & 1) * 64;
#pragma unroll
    for (int j = 0; j < 4; j++) cols[j] = *reinterpret_cast<const u32x4*>(mine + j * 16);
    if (dc_apart && !(lane & 1)) cols[0].x = (cols[0].x & 0xFFFF0000u) | dc;
}

// The same for a tile of the luma kernel: slot j of the wave is block (row0 + (j >> row_shift), col0 + (j & col_mask)) --
// one run of 32 blocks (row_shift 5, col_mask 31) or two runs of 16 (row_shift 4, col_mask 15; narrow tiles).
__device__ __forceinline__ void fetch_tile_half_block(const DecodeComponent& cd, int row0, int col0, int row_shift, int col_mask, int bw, int bh,
                                                      char* lds_wave, int lane, u32x4 (&cols)[4])
{
    if (cd.block_off) {  // (wave-uniform: a property of the picture)
        const int j = lane >> 1, row = row0 + (j >> row_shift), col = col0 + (j & col_mask);
        sparse_expand_slots(cd, row < bh && col < bw, (unsigned)(row * bw + col), lds_wave, lane);
        read_half_block(lds_wave, lane, 0u, false, cols);
        return;
    }
    const u32x4* src = reinterpret_cast<const u32x4*>(cd.coef);
    const bool dc_apart = cd.dc_stride != 64;  // wave-uniform: host-decoded images carry the DC inside the block already
    unsigned dc = 0;
    {
        const int j = lane >> 1, row = row0 + (j >> row_shift), col = col0 + (j & col_mask);
        if (dc_apart && row < bh && col < bw) dc = *(const HJ_GLOBAL unsigned short*)((const HJ_GLOBAL char*)cd.dc + (unsigned)(row * bw + col) * 2u);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int g = k * 64 + lane, j = g >> 3;
        const int row = row0 + (j >> row_shift), col = col0 + (j & col_mask);
        u32x4 v = {0u, 0u, 0u, 0u};
        // byte offset in 32 bits (a component's blocks are far below 4 GB): SGPR base + VGPR offset addressing
        if (row < bh && col < bw)
#ifdef HJ_DEC_PLAIN_LOADS
            v = *(const HJ_GLOBAL u32x4*)((const HJ_GLOBAL char*)src + ((unsigned)(row * bw + col) * 128u + (unsigned)(g & 7) * 16u));
#else
            v = __builtin_nontemporal_load((const HJ_GLOBAL u32x4*)((const HJ_GLOBAL char*)src + ((unsigned)(row * bw + col) * 128u + (unsigned)(g & 7) * 16u)));
#endif
        *reinterpret_cast<u32x4*>(lds_wave + j * kLdsBlockStride + (g & 7) * 16) = v;
    }
    wave_lds_fence();
    const char* mine = lds_wave + (lane >> 1) * kLdsBlockStride + (lane & 1) * 64;
#pragma unroll
    for (int j = 0; j < 4; j++) cols[j] = *reinterpret_cast<const u32x4*>(mine + j * 16);
    if (dc_apart && !(lane & 1)) cols[0].x = (cols[0].x & 0xFFFF0000u) | dc;
}

// ---- dequantize + IDCT of one 8x8 block by a lane pair, on int16 pairs ---------------------------------------------------
// What is restated: jsimd_idct_islow_sse2 / _avx2 of libjpeg-turbo (simd/x86_64/jidctint-*.asm), the routine behind
// jpeg_idct_islow in the reference's CPU path (extensions/libjpeg_turbo/jpeg_mem.cpp:174-177; the library is built with its
// defaults, external/build_libjpeg-turbo.sh:36-39).  It equals jidctint.c on every stream an encoder can write and differs out
// of gamut because it works on 16-bit lanes; those semantics are kept exactly (oracle/jpeg_oracle.c oj_idct_block_simd, pinned
// by tests/golden/gamut from the library itself):
//   pmullw      dequantization keeps the low 16 bits of coefficient x quantizer          -> v_pk_mul_lo_u16
//   paddw/psubw in0 +- in4, z3 = in7 + in3, z4 = in5 + in1 wrap in 16 bits               -> v_pk_add_u16 / v_pk_sub_u16
//   pmaddwd     every product pairs two values with two constants, exact in 32 bits     -> v_dot2_i32_i16
//   packssdw    pass-1 results saturate to int16                                         -> v_cvt_pk_i16_i32
//   psllw       a block whose rows 1..7 are all zero skips pass 1: row 0 << 2 in 16 bits -> v_pk_lshlrev_b16 (rare branch)
//   packsswb    the result saturates to int8, + 128                                      -> v_sat_pk_u8_i16 / v_pk_max/min_i16
// No intermediate leaves int32 (|.| < 1.7e9 for int16 inputs), so the 32-bit sums may be regrouped freely; the 16-bit ones may not.
constexpr unsigned pk16(int lo, int hi) { return ((unsigned)lo & 0xFFFFu) | (((unsigned)hi & 0xFFFFu) << 16); }
using u16x2 = __attribute__((ext_vector_type(2))) unsigned short;
__device__ __forceinline__ unsigned pk_mul16(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned pk_sub16(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b)));
}
// a.lo * k.lo + a.hi * k.hi + acc, three-operand form pinned: left alone hipcc picks v_dot2c_i32_i16 (accumulator = destination)
// and spends a v_mov per product on initialising it
__device__ __forceinline__ int dot2(unsigned a, unsigned k, int acc)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(acc));
    return r;
}
__device__ __forceinline__ int dot2(unsigned a, unsigned k)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "s"(k));
    return r;
}
// (int16)a.lo * k + c
__device__ __forceinline__ int mad_lo16(unsigned a, int k, int c)
{
    int r;
    asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c));
    return r;
}
// two int32 -> two int16 with signed saturation (packssdw)
__device__ __forceinline__ unsigned sat_pk16(int lo, int hi)
{
    unsigned r;
    asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ unsigned pk_ashr16_2(unsigned a)
{
    return __builtin_bit_cast(unsigned, (i16x2)(__builtin_bit_cast(i16x2, a) >> (short)2));
}
__device__ __forceinline__ unsigned pk_shl16_2(unsigned a)
{
    return __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, a) << (unsigned short)2));
}

constexpr unsigned kE10 = pk16(F_0_541 + F_0_765, F_0_541), kE13 = pk16(-(F_0_541 + F_0_765), -F_0_541);
constexpr unsigned kE11 = pk16(F_0_541, F_0_541 - F_1_847), kE12 = pk16(-F_0_541, F_1_847 - F_0_541);
constexpr unsigned kZ3 = pk16(F_1_175 - F_1_961, F_1_175), kZ4 = pk16(F_1_175, F_1_175 - F_0_390);
constexpr unsigned kT0 = pk16(F_0_298 - F_0_899, -F_0_899), kT3 = pk16(-F_0_899, F_1_501 - F_0_899);
constexpr unsigned kT1 = pk16(F_2_053 - F_2_562, -F_2_562), kT2 = pk16(-F_2_562, F_3_072 - F_2_562);

// One 1-D pass over eight int16 values given as pairs A = (i0,i1), B = (i2,i3), C = (i4,i5), D = (i6,i7); o[k] = output k before the
// descale shift, rounding term included (rnd lives in a register: a VOP3 instruction takes one scalar operand, and that is the constant).
__device__ __forceinline__ void idct1d_pk(unsigned A, unsigned B, unsigned C, unsigned D, int rnd, int (&o)[8])
{
    const unsigned P = pk_add16(A, C), M = pk_sub16(A, C), Q = pk_add16(B, D);  // (i0+i4, z4), (i0-i4, .), (., z3) in 16 bits
    const unsigned p26 = lo_pair((int)B, (int)D), pz = hi_pair((int)Q, (int)P), p71 = hi_pair((int)D, (int)A), p53 = hi_pair((int)C, (int)B);
    const int t0e = mad_lo16(P, 8192, rnd), t1e = mad_lo16(M, 8192, rnd);
    const int tmp10 = dot2(p26, kE10, t0e), tmp13 = dot2(p26, kE13, t0e), tmp11 = dot2(p26, kE11, t1e), tmp12 = dot2(p26, kE12, t1e);
    const int z3 = dot2(pz, kZ3), z4 = dot2(pz, kZ4);
    const int t0 = dot2(p71, kT0, z3), t3 = dot2(p71, kT3, z4), t1 = dot2(p53, kT1, z4), t2 = dot2(p53, kT2, z3);
    o[0] = tmp10 + t3;
    o[7] = tmp10 - t3;
    o[1] = tmp11 + t2;
    o[6] = tmp11 - t2;
    o[2] = tmp12 + t1;
    o[5] = tmp12 - t1;
    o[3] = tmp13 + t0;
    o[4] = tmp13 - t0;
}

// Eight selects between an own register and the partner lane's: r[k] = vcc ? b[k] : a[k] of the pair's EVEN (ODD = false) or ODD lane.
// v_cndmask_b32_dpp is VOP2: its mask is VCC, which inline asm cannot name as an operand -- hence one block per mask.
template <bool FROM_ODD>
__device__ __forceinline__ void pair_select8(unsigned (&r)[8], const unsigned (&a)[8], const unsigned (&b)[8], unsigned long long mask)
{
    if constexpr (FROM_ODD)
        asm("s_mov_b64 vcc, %24\n\t"
            "v_cndmask_b32_dpp %0, %8, %16, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %1, %9, %17, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %2, %10, %18, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %3, %11, %19, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %4, %12, %20, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %5, %13, %21, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %6, %14, %22, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %7, %15, %23, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf"
            : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
              "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "s"(mask)
            : "vcc");
    else
        asm("s_mov_b64 vcc, %24\n\t"
            "v_cndmask_b32_dpp %0, %8, %16, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %1, %9, %17, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %2, %10, %18, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %3, %11, %19, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %4, %12, %20, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %5, %13, %21, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %6, %14, %22, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_cndmask_b32_dpp %7, %15, %23, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf"
            : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
              "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "s"(mask)
            : "vcc");
}

// The whole transform.  cols[j]: this lane's column 4p+j as stored (row pairs (0,1),(2,3),(4,5),(6,7)); qp: this lane's 16 quantizer
// pairs in the same layout (DecodeComponent::qpk[p]).  Result: out[i][h] = samples (2h, 2h+1) of image row (p ? 7-i : i) as an int16
// pair, BEFORE the final clamp to -128..127 and the +128 (|.| < 2^13: the caller's v_sat_pk_u8_i16 / packed min-max finishes the job).
__device__ __forceinline__ void idct_block_pair(const u32x4 (&cols)[4], const unsigned* __restrict__ qp, bool p, unsigned (&out)[4][4])
{
    unsigned dq[4][4];
    unsigned big = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const u32x4 q4 = *reinterpret_cast<const u32x4*>(qp + j * 4);
        dq[j][0] = pk_mul16(cols[j].x, q4.x);
        dq[j][1] = pk_mul16(cols[j].y, q4.y);
        dq[j][2] = pk_mul16(cols[j].z, q4.z);
        dq[j][3] = pk_mul16(cols[j].w, q4.w);
        big |= (dq[j][0] + 0x2000u) & 0xC000u;  // row 0 outside [-8192, 8191]: only then can the shortcut below differ from the full pass
    }
    // workspace after pass 1: ws[jj][r] = (column 4p+2jj, column 4p+2jj+1) of row r, saturated to int16
    unsigned ws[2][8];
    const int rnd1 = 1 << 10, rnd2 = 1 << 17;
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
        int oa[8], ob[8];
        idct1d_pk(dq[2 * jj][0], dq[2 * jj][1], dq[2 * jj][2], dq[2 * jj][3], rnd1, oa);
        idct1d_pk(dq[2 * jj + 1][0], dq[2 * jj + 1][1], dq[2 * jj + 1][2], dq[2 * jj + 1][3], rnd1, ob);
#pragma unroll
        for (int r = 0; r < 8; r++) ws[jj][r] = sat_pk16(oa[r] >> 11, ob[r] >> 11);
    }
    if (__builtin_amdgcn_ballot_w64(big != 0) != 0) {
        // jidctint-*.asm: "AC terms all zero" -- tested on the block as stored (rows 1..7 of all eight columns), then the
        // workspace is the dequantized row 0 shifted left in 16-bit lanes (psllw wraps where the full pass would saturate)
        unsigned ac = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) ac |= (cols[j].x & 0xFFFF0000u) | cols[j].y | cols[j].z | cols[j].w;
        ac |= (unsigned)pair_swap((int)ac);
        if (ac == 0) {
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const unsigned v = pk_shl16_2(lo_pair((int)dq[2 * jj][0], (int)dq[2 * jj + 1][0]));
#pragma unroll
                for (int r = 0; r < 8; r++) ws[jj][r] = v;
            }
        }
    }
    // hand-over: slot i of lane 0 is image row i, of lane 1 row 7-i.  X = columns 0..3 (lane 0's), Y = columns 4..7 (lane 1's).
    unsigned own_fwd[8], own_rev[8], X[8], Y[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        own_fwd[i] = ws[0][i];
        own_fwd[4 + i] = ws[1][i];
        own_rev[i] = ws[0][7 - i];
        own_rev[4 + i] = ws[1][7 - i];
    }
    // X: even lane keeps its rows i, odd lane takes the even lane's rows 7-i;  Y: odd lane keeps its rows 7-i, even lane takes the odd lane's rows i
    pair_select8<false>(X, own_rev, own_fwd, 0x5555555555555555ull);
    pair_select8<true>(Y, own_fwd, own_rev, 0xAAAAAAAAAAAAAAAAull);
    (void)p;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int o[8];
        idct1d_pk(X[i], X[4 + i], Y[i], Y[4 + i], rnd2, o);
        // >> 18 = the high half >> 2
#pragma unroll
        for (int h = 0; h < 4; h++) out[i][h] = pk_ashr16_2(hi_pair(o[2 * h], o[2 * h + 1]));
    }
}

// ------------------------------------------------------------------------------------------------
// K1: IDCT of 128 consecutive blocks of one component into a u8 plane (internal plane or user output).
// ------------------------------------------------------------------------------------------------
template <bool FUSED>
__device__ __forceinline__ void idct_plane_body(const DecodeImage& im, const WorkUnit& u, char* lds, HuffImage* hi = nullptr, HJ_LDS uint16_t* pool = nullptr,
                                                FusedShared* fs = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool p = lane & 1;
    const DecodeComponent& cd = im.comp[u.comp];
    const int bw = cd.blocks_w, nblocks = bw * cd.blocks_h;
    const int wave_first = u.block_base + wave * kBlocksPerWave;
    u32x4 cols[4];
    if constexpr (FUSED) {
        char* lds_wave = lds + wave * kBlocksPerWave * kLdsBlockStride;
        const int mine = wave_first + (lane >> 1);
        unsigned dc = 0;
        if (mine < nblocks) dc = ((const HJ_GLOBAL unsigned short*)cd.dc)[mine];
        const int jb = wave_first + lane;  // the block lane j < 32 decodes
        const int jrow = jb / bw, jcol = jb - jrow * bw;
        fused_decode_slots(*hi, pool, *fs, (int)u.comp, jb < nblocks, jrow, jcol, lds_wave, lane);
        read_half_block(lds_wave, lane, dc, true, cols);
    } else
        fetch_half_block(cd, wave_first, nblocks, lds + wave * kBlocksPerWave * kLdsBlockStride, lane, cols);
    const int b = wave_first + (lane >> 1);
    if (b >= nblocks) return;  // whole pairs leave together
    unsigned rows[4][4];
    idct_block_pair(cols, cd.qpk[p], p, rows);

    const int by = b / bw, bx = b - by * bw;
    const bool to_out = (u.mode & 0xFF) == kToOutput;
    const int op = u.mode >> 8;
    uint8_t* dst = to_out ? im.out[op] : cd.plane;
    const unsigned pitch = to_out ? im.out_pitch[op] : cd.plane_pitch;
    // kToPlane: every allocated block is written; kToOutput: crop to the true component size
    const int lim_w = to_out ? cd.samp_w : bw * 8;
    const int lim_h = to_out ? cd.samp_h : cd.blocks_h * 8;
    const int x0 = bx * 8, y0 = by * 8;
    if (x0 >= lim_w) return;
    uint8_t* base = dst + (size_t)y0 * pitch + x0;
    const bool fast = (x0 + 8 <= lim_w) && (((uintptr_t)base | pitch) & 7) == 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = p ? 7 - i : i;
        if (y0 + r < lim_h) {
            // packsswb + paddb 128 of the SIMD routine = saturate (v + 128) to 0..255: v_sat_pk_u8_i16 clamps and packs the bytes
            unsigned pr[4];
#pragma unroll
            for (int h = 0; h < 4; h++) pr[h] = pk_add16(rows[i][h], 0x00800080u);
            const uint2 px = make_uint2(sat_pk4(pr[0], pr[1]), sat_pk4(pr[2], pr[3]));
            uint8_t* qd = base + __umul24((unsigned)r, pitch);
            if (fast) {
                *reinterpret_cast<uint2*>(qd) = px;
            } else {
#pragma unroll
                for (int c = 0; c < 8; c++)
                    if (x0 + c < lim_w) qd[c] = (uint8_t)((c < 4 ? px.x : px.y) >> (8 * (c & 3)));
            }
        }
    }
}

__global__ __launch_bounds__(kThreads, HJ_MIN_WAVES) void idct_plane_kernel(const DecodeImage* __restrict__ images, const WorkUnit* __restrict__ units)
{
    __shared__ __attribute__((aligned(16))) char lds[4 * kBlocksPerWave * kLdsBlockStride];
    const WorkUnit u = units[blockIdx.x];
    idct_plane_body<false>(images[u.image], u, lds);
}

__global__ __launch_bounds__(kThreads, HJ_MIN_WAVES) void idct_plane_fused_kernel(const DecodeImage* __restrict__ images, const WorkUnit* __restrict__ units,
                                                                                 HuffImage* __restrict__ himages)
{
    __shared__ __attribute__((aligned(16))) char lds[4 * kBlocksPerWave * kLdsBlockStride];
    __shared__ FusedShared fs;
    extern __shared__ uint16_t dyn_pool[];
    const WorkUnit u = units[blockIdx.x];
    const DecodeImage& im = images[u.image];
    HuffImage& hi = himages[im.huff_index];
    fused_stage(hi, (HJ_LDS uint16_t*)dyn_pool, fs);
    idct_plane_body<true>(im, u, lds, &hi, (HJ_LDS uint16_t*)dyn_pool, &fs);
}

// ------------------------------------------------------------------------------------------------
// K2: luma IDCT fused with chroma upsampling, colour conversion and the output store.
//   HS, VS = chroma upsampling factors (1 or 2); HS == 0 means "no chroma" (gray source -> RGB).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int byte_of(uint2 v, int j) { return (int)__builtin_amdgcn_ubfe(j < 4 ? v.x : v.y, 8 * (j & 3), 8); }

// Loads four chroma window rows.  Row k of the result is window row (p ? first_row + 3 - k : first_row + k) -- lane 1 works
// through its image rows in descending order, so it loads its window upside down and both lanes index it identically.
// Byte j of a row is the sample at column clamp(wx + j, 0, dw-1): libjpeg's edge replication is already applied.
template <int NW>
__device__ __forceinline__ void load_chroma_rows(const uint8_t* __restrict__ plane, unsigned pitch, int dw, int dh, int wx, int first_row, bool p,
                                                 uint2 (&rows)[4])
{
    const int base = max(wx, 0);
    const bool edge = (wx < 0) || (wx + NW - 1 > dw - 1);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int y = min(max(first_row + (p ? 3 - k : k), 0), dh - 1);
        const uint8_t* src = plane + (__umul24((unsigned)y, pitch) + (unsigned)base);  // planes are far smaller than 4 GB
        // planes are allocated with >= 16 bytes of slack per row, so an 8-byte read starting inside the row is in bounds
        uint2 v;
#ifdef HJ_ABLATE_CHROMA
        v = make_uint2(0x80808080u + (unsigned)(uintptr_t)src, 0x80808080u);
#else
        {
            typedef u32x2 __attribute__((aligned(1))) u32x2_unaligned;  // the window starts at any byte; gfx950 loads it in one go
            const u32x2 t = *(const HJ_GLOBAL u32x2_unaligned*)src;
            v = make_uint2(t.x, t.y);
        }
#endif
        rows[k] = v;
    }
    if (__builtin_amdgcn_ballot_w64(edge) != 0) {
        if (edge) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned long long v = ((unsigned long long)rows[k].y << 32) | rows[k].x;
                unsigned long long f = 0;
#pragma unroll
                for (int j = 0; j < NW; j++) {
                    const int idx = min(max(wx + j, 0), dw - 1) - base;
                    f |= ((v >> (8 * idx)) & 0xFFull) << (8 * j);
                }
                rows[k] = make_uint2((unsigned)f, (unsigned)(f >> 32));
            }
        }
    }
}

// Upsampled chroma for one output row of the lane's 8 pixels.  `odd_row`: the image row is odd (matters for h1v2 only).
template <int HS, int VS>
__device__ __forceinline__ void upsample_row(uint2 near, uint2 far, bool odd_row, bool fancy, int (&o)[8])
{
    if constexpr (HS == 2) {
        // window byte j <-> chroma column (4*bx - 1 + j); output pixel 2i+e sits over window column i+1
        if (fancy) {
            // jdsample.c h2v2_fancy_upsample / h2v1_fancy_upsample: vertical (3,1) then horizontal (3,1) triangle filters.
            // Both passes of one output pixel are ONE v_dot4_u32_u8 over the bytes [n_j, n_j+1, f_j, f_j+1] (n = near row,
            // f = far row, j = window column) with weights (3,9,1,3) for the left pixel of column j+1 and (9,3,3,1) for the
            // right pixel of column j; libjpeg's rounding terms (8 / 7, h2v1: 1 / 2) ride in the accumulator operand.
            const unsigned q0 = __builtin_amdgcn_perm(far.x, near.x, 0x05040100u);  // n0 n1 f0 f1
            const unsigned q2 = __builtin_amdgcn_perm(far.x, near.x, 0x07060302u);  // n2 n3 f2 f3
            const unsigned q4 = __builtin_amdgcn_perm(far.y, near.y, 0x05040100u);  // n4 n5 f4 f5
            const unsigned q1 = __builtin_amdgcn_perm(q2, q0, 0x06030401u);         // n1 n2 f1 f2
            const unsigned q3 = __builtin_amdgcn_perm(q4, q2, 0x06030401u);         // n3 n4 f3 f4
            const unsigned qq[5] = {q0, q1, q2, q3, q4};
            constexpr unsigned w_left = VS == 2 ? 0x03010903u : 0x00000301u, w_right = VS == 2 ? 0x01030309u : 0x00000103u;
            constexpr unsigned r_left = VS == 2 ? 8 : 1, r_right = VS == 2 ? 7 : 2;
            constexpr int sh = VS == 2 ? 4 : 2;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                o[2 * i] = (int)(__builtin_amdgcn_udot4(qq[i], w_left, r_left, false) >> sh);
                o[2 * i + 1] = (int)(__builtin_amdgcn_udot4(qq[i + 1], w_right, r_right, false) >> sh);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) o[2 * i] = o[2 * i + 1] = byte_of(near, i + 1);
        }
    } else {
        if (VS == 2 && fancy) {
            const int bias = odd_row ? 2 : 1;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (3 * byte_of(near, j) + byte_of(far, j) + bias) >> 2;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = byte_of(near, j);
        }
    }
}

// Work unit of this kernel: a tile of 32 x 4 luma blocks -- block_base = first block column, comp = first block row;
// wave w of the workgroup owns the 32 consecutive blocks of block row (comp + w).
constexpr int kOutRowBytes = kBlocksPerWave * 24 + 16;  // one pixel row of a wave's 32 blocks (interleaved RGB) + bank-skew pad
constexpr int kLdsLumaWaveBytes = 8 * kOutRowBytes;     // 6,272 B: first the coefficient staging (4,608 B), then the RGB tile
constexpr int kNarrowRowBytes = 16 * 24;                // narrow tiles: 16 pixel rows of 16 blocks (6,144 B)

//   COMMON = the configuration nearly every caller uses, fixed at compile time: YCbCr source, interleaved RGB output (any base
//   address and pitch), libjpeg's default fancy upsampling.  Same arithmetic, but no wave-uniform format branches and
//   ~20 fewer live registers (85 instead of 104 VGPRs: five waves per SIMD without spills).  The host picks the kernel per
//   image (DecodeBatch::finalize).
// LAYOUT: 0 = whatever the descriptor says (run-time branches), 1 = COMMON, 2 = COMMON with planar output (P_RGB / P_BGR: what
// CHW consumers ask for) -- same arithmetic, one set of format flags fixed at compile time each
enum LumaLayout : int { kLayoutAny = 0, kLayoutInterleaved = 1, kLayoutPlanar = 2 };
template <int HS, int VS, int LAYOUT, bool FUSED = false>
__device__ __forceinline__ void luma_color_body(const DecodeImage& im, const WorkUnit& u, char* lds, HuffImage* hi = nullptr, HJ_LDS uint16_t* pool = nullptr,
                                                FusedShared* fs = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool p = lane & 1;
    const int blk = lane >> 1;
    const int bw = im.comp[0].blocks_w, bh = im.comp[0].blocks_h;
    // tile shapes (WorkUnit::mode, wave-uniform): 0 = 32 x 4 blocks, a wave owns 32 blocks of one block row;
    //                                             1 = 16 x 8 blocks ("narrow": the ragged right edge of widths like 1920 = 7.5 x 256
    //                                                 pixels), a wave owns 16 blocks of two block rows -- no half-idle waves
    const bool narrow = u.mode != 0;
    const int row_shift = narrow ? 4 : 5, col_mask = narrow ? 15 : 31;
    const int bx0 = u.block_base, by_wave = (int)u.comp + (narrow ? 2 * wave : wave);
    const int bx = bx0 + (blk & col_mask), by = by_wave + (blk >> row_shift);
    char* lds_wave = lds + wave * kLdsLumaWaveBytes;
    u32x4 cols[4];
    const int x0 = bx * 8, y0 = by * 8;
    const int W = im.width, H = im.height;
    const bool valid = bx < bw && by < bh && x0 < W && y0 < H;  // false: pair idles (block is MCU padding or outside the tile)
    if constexpr (FUSED) {
        unsigned dc = 0;
        if (valid) dc = *(const HJ_GLOBAL unsigned short*)((const HJ_GLOBAL char*)im.comp[0].dc + (unsigned)(by * bw + bx) * 2u);
        const int jrow = by_wave + (lane >> row_shift), jcol = bx0 + (lane & col_mask);  // the block lane j < 32 decodes
        // every block of the grid inside the tile, visible or not: the decode is also the stream's CHECK (the block pass decoded them all),
        // and damage that sits in the MCU padding must flag the image like damage anywhere else
        fused_decode_slots(*hi, pool, *fs, 0, jrow < bh && jcol < bw, jrow, jcol, lds_wave, lane);
        read_half_block(lds_wave, lane, dc, true, cols);
    } else
        fetch_tile_half_block(im.comp[0], by_wave, bx0, row_shift, col_mask, bw, bh, lds_wave, lane, cols);

    const int fmt = im.out_format;
    constexpr bool COMMON = LAYOUT != kLayoutAny;
    const bool planar = LAYOUT == kLayoutPlanar || (!COMMON && (fmt == kOutPlanarRGB || fmt == kOutPlanarBGR));
    const bool bgr = LAYOUT == kLayoutPlanar ? fmt == kOutPlanarBGR : LAYOUT == kLayoutInterleaved ? fmt == kOutInterleavedBGR
                                                                       : (fmt == kOutInterleavedBGR || fmt == kOutPlanarBGR);
    const bool ycc = COMMON || im.color_model == 1;
    const bool full = x0 + 8 <= W;
    // Interleaved output goes through an LDS tile so that the wave emits 16 B per lane, fully coalesced, instead of
    // 24-byte-strided 8-byte stores.
    const bool staged = !planar;  // (any base address and pitch: gfx950 stores an unaligned 16-byte piece in one go)

    if (valid) {
        // chroma window rows for this lane's four image rows (issued before the IDCT so the loads fly while we compute)
        constexpr int NW = HS == 2 ? 6 : 8;
        uint2 cbw[4], crw[4];
        bool fancy = false;
        if constexpr (HS != 0) {
            const int dw = im.comp[1].samp_w, dh = im.comp[1].samp_h;
            // libjpeg picks the triangle filters only when do_fancy_upsampling and (for h2v1/h2v2) downsampled_width > 2
            fancy = COMMON || ((im.flags & kFlagFancyUpsampling) && (HS == 1 || dw > 2));
            const int wx = HS == 2 ? 4 * bx - 1 : 8 * bx;
            // VS == 2: image rows 0..3 need chroma rows 4by-1 .. 4by+2, rows 4..7 need 4by+1 .. 4by+4;  VS == 1: rows map 1:1
            const int first = VS == 2 ? 4 * by - 1 + (p ? 2 : 0) : 8 * by + (p ? 4 : 0);
            // BGR output of a YCbCr source costs nothing: the two chroma windows trade places (a scalar pointer select) and so do the
            // constants of the colour conversion below -- what comes out first is then B, what comes out third is R
            const uint8_t* plane1 = im.comp[1].plane;
            const uint8_t* plane2 = im.comp[2].plane;
            const unsigned pitch1 = im.comp[1].plane_pitch, pitch2 = im.comp[2].plane_pitch;
            const bool swap = ycc && bgr;
            load_chroma_rows<NW>(swap ? plane2 : plane1, swap ? pitch2 : pitch1, dw, dh, wx, first, p, cbw);
            load_chroma_rows<NW>(swap ? plane1 : plane2, swap ? pitch1 : pitch2, dw, dh, wx, first, p, crw);
        }

        unsigned rows[4][4];
        idct_block_pair(cols, im.comp[0].qpk[p], p, rows);
        // additive constants of the colour conversion (kept in VGPRs: a VOP3 instruction reads one scalar operand at most)
        // (+ 128 << 16: the luma offset, see the packed colour stage below)
        const bool swapped = ycc && bgr;  // crw holds Cb and cbw holds Cr (see the window loads above)
        const int mr = swapped ? 116130 : 91881, mb = swapped ? 91881 : 116130;       // multiplier of the term built from crw / cbw
        const int gr = swapped ? -22554 : -46802, gb = swapped ? -46802 : -22554;     // green's multipliers for crw / cbw
        const int kr = 32768 - 128 * mr + (128 << 16), kb = 32768 - 128 * mb + (128 << 16), kg = 32768 + 128 * 22554 + 128 * 46802 + (128 << 16);

#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = p ? 7 - i : i;  // image row inside the block
            if (y0 + r >= H) continue;
            // q: the row's 24 output bytes (interleaved), or q[2k], q[2k+1] = the eight bytes of output plane k (planar)
            unsigned q[6];
            bool packed_done = false;
            if constexpr (HS != 0) {
                if (ycc) {
                    packed_done = true;
                    int cb[8], cr[8];
                    // local window-row indices are the same for both lanes (lane 1's window is loaded upside down):
                    //   VS == 2:  i = 0: near 1 far 0 | 1: near 1 far 2 | 2: near 2 far 1 | 3: near 2 far 3      VS == 1: row i
                    constexpr int kNear[4] = {1, 1, 2, 2}, kFar[4] = {0, 2, 1, 3};
                    const int near = VS == 2 ? kNear[i] : i, far = VS == 2 ? kFar[i] : i;
                    const bool odd_row = (i & 1) != (int)p;  // r = i or 7 - i
                    upsample_row<HS, VS>(cbw[near], cbw[far], odd_row, fancy, cb);
                    upsample_row<HS, VS>(crw[near], crw[far], odd_row, fancy, cr);
                    // jdcolor.c ycc_rgb_convert with SCALEBITS = 16, two pixels per register from here on:
                    //   y = clamp(s + 128, 0, 255) = clamp(s, -128, 127) + 128 with s the IDCT's result (packsswb), and the
                    //   +128 as well as the (x - 128) of the chroma terms live in the additive constants, so
                    //   bits 31..16 of one 24-bit mad ARE "chroma term + 128" as an int16 (|.| < 2^9);
                    //   v_sat_pk_u8_i16 is the final clamp to 0..255 and the byte packing in one instruction.
                    unsigned s0[4], s1[4], s2[4];
#pragma unroll
                    for (int c = 0; c < 8; c += 2) {
                        const unsigned yp = clamp_s8_pair(rows[i][c >> 1]);
                        // (names as for RGB output; for BGR the windows and constants have traded places: tr is then the blue term)
                        const unsigned tr = hi_pair(mad24(cr[c], mr, kr), mad24(cr[c + 1], mr, kr));
                        const unsigned tb = hi_pair(mad24(cb[c], mb, kb), mad24(cb[c + 1], mb, kb));
                        const unsigned tg = hi_pair(mad24(cr[c], gr, mad24(cb[c], gb, kg)), mad24(cr[c + 1], gr, mad24(cb[c + 1], gb, kg)));
                        s0[c >> 1] = pk_add16(yp, tr);
                        s1[c >> 1] = pk_add16(yp, tg);
                        s2[c >> 1] = pk_add16(yp, tb);
                    }
                    if (planar) {
                        q[0] = sat_pk4(s0[0], s0[1]);
                        q[1] = sat_pk4(s0[2], s0[3]);
                        q[2] = sat_pk4(s1[0], s1[1]);
                        q[3] = sat_pk4(s1[2], s1[3]);
                        q[4] = sat_pk4(s2[0], s2[1]);
                        q[5] = sat_pk4(s2[2], s2[3]);
                    } else {
#pragma unroll
                        for (int h = 0; h < 2; h++) {  // pixels 4h .. 4h+3 -> three dwords
                            const unsigned rg01 = sat_pk4(s0[2 * h], s1[2 * h]);          // R0 R1 G0 G1
                            const unsigned rg23 = sat_pk4(s0[2 * h + 1], s1[2 * h + 1]);  // R2 R3 G2 G3
                            const unsigned b03 = sat_pk4(s2[2 * h], s2[2 * h + 1]);       // B0 B1 B2 B3
                            q[3 * h] = __builtin_amdgcn_perm(b03, rg01, 0x01040200u);      // R0 G0 B0 R1
                            const unsigned m = __builtin_amdgcn_perm(b03, rg01, 0x00000503u);  // G1 B1 . .
                            q[3 * h + 1] = __builtin_amdgcn_perm(rg23, m, 0x06040100u);    // G1 B1 R2 G2
                            q[3 * h + 2] = __builtin_amdgcn_perm(b03, rg23, 0x07030106u);  // B2 R3 G3 B3
                        }
                    }
                }
            }
            if (!packed_done) {
                int R[8], G[8], B[8];
                // the luma samples as bytes (v_sat_pk_u8_i16 of value + 128, as in K1)
                const uint2 ypx = make_uint2(sat_pk4(pk_add16(rows[i][0], 0x00800080u), pk_add16(rows[i][1], 0x00800080u)),
                                             sat_pk4(pk_add16(rows[i][2], 0x00800080u), pk_add16(rows[i][3], 0x00800080u)));
                if constexpr (HS == 0) {
#pragma unroll
                    for (int c = 0; c < 8; c++) R[c] = G[c] = B[c] = byte_of(ypx, c);
                } else {
                    // Adobe RGB JPEG: the three components already are R, G, B (same window rows as above)
                    constexpr int kNear[4] = {1, 1, 2, 2}, kFar[4] = {0, 2, 1, 3};
                    const int near = VS == 2 ? kNear[i] : i, far = VS == 2 ? kFar[i] : i;
                    const bool odd_row = (i & 1) != (int)p;
                    upsample_row<HS, VS>(cbw[near], cbw[far], odd_row, fancy, G);
                    upsample_row<HS, VS>(crw[near], crw[far], odd_row, fancy, B);
#pragma unroll
                    for (int c = 0; c < 8; c++) R[c] = byte_of(ypx, c);
                }
                if (bgr) {
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        const int t = R[c];
                        R[c] = B[c];
                        B[c] = t;
                    }
                }
                if (planar) {
                    q[0] = pack4(R[0], R[1], R[2], R[3]);
                    q[1] = pack4(R[4], R[5], R[6], R[7]);
                    q[2] = pack4(G[0], G[1], G[2], G[3]);
                    q[3] = pack4(G[4], G[5], G[6], G[7]);
                    q[4] = pack4(B[0], B[1], B[2], B[3]);
                    q[5] = pack4(B[4], B[5], B[6], B[7]);
                } else {
                    q[0] = pack4(R[0], G[0], B[0], R[1]);
                    q[1] = pack4(G[1], B[1], R[2], G[2]);
                    q[2] = pack4(B[2], R[3], G[3], B[3]);
                    q[3] = pack4(R[4], G[4], B[4], R[5]);
                    q[4] = pack4(G[5], B[5], R[6], G[6]);
                    q[5] = pack4(B[6], R[7], G[7], B[7]);
                }
            }
            // row offsets: (wave-uniform 64-bit part) + (per-lane 32-bit part) keeps 64-bit multiplies off the vector ALU
            if (planar) {
                uint8_t* p0 = im.out[0] + (size_t)y0 * im.out_pitch[0] + (__umul24((unsigned)r, im.out_pitch[0]) + (unsigned)x0);
                uint8_t* p1 = im.out[1] + (size_t)y0 * im.out_pitch[1] + (__umul24((unsigned)r, im.out_pitch[1]) + (unsigned)x0);
                uint8_t* p2 = im.out[2] + (size_t)y0 * im.out_pitch[2] + (__umul24((unsigned)r, im.out_pitch[2]) + (unsigned)x0);
                if (full && (((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) & 7) == 0) {
                    *reinterpret_cast<uint2*>(p0) = make_uint2(q[0], q[1]);
                    *reinterpret_cast<uint2*>(p1) = make_uint2(q[2], q[3]);
                    *reinterpret_cast<uint2*>(p2) = make_uint2(q[4], q[5]);
                } else {
#pragma unroll
                    for (int c = 0; c < 8; c++)
                        if (x0 + c < W) {
                            p0[c] = (uint8_t)(q[c >> 2] >> (8 * (c & 3)));
                            p1[c] = (uint8_t)(q[2 + (c >> 2)] >> (8 * (c & 3)));
                            p2[c] = (uint8_t)(q[4 + (c >> 2)] >> (8 * (c & 3)));
                        }
                }
            } else {
                if (staged) {
                    uint2* t = reinterpret_cast<uint2*>(lds_wave + (narrow ? ((blk >> 4) * 8 + r) * kNarrowRowBytes + (blk & 15) * 24 : r * kOutRowBytes + blk * 24));
                    t[0] = make_uint2(q[0], q[1]);
                    t[1] = make_uint2(q[2], q[3]);
                    t[2] = make_uint2(q[4], q[5]);
                } else {
                    uint8_t* o = im.out[0] + (size_t)y0 * im.out_pitch[0] + (__umul24((unsigned)r, im.out_pitch[0]) + (unsigned)x0 * 3u);
                    if (full && ((uintptr_t)o & 7) == 0) {
                        uint2* t = reinterpret_cast<uint2*>(o);
                        t[0] = make_uint2(q[0], q[1]);
                        t[1] = make_uint2(q[2], q[3]);
                        t[2] = make_uint2(q[4], q[5]);
                    } else {
#pragma unroll
                        for (int c = 0; c < 8; c++)
                            if (x0 + c < W) {
#pragma unroll
                                for (int k = 0; k < 3; k++) o[3 * c + k] = (uint8_t)(q[(3 * c + k) >> 2] >> (8 * ((3 * c + k) & 3)));
                            }
                    }
                }
            }
        }
    }

    const int y0w = by_wave * 8;
    if (!staged || by_wave >= bh || y0w >= H) return;  // wave-uniform
    wave_lds_fence();
    // the wave's tile: pixel rows y0w .. y0w + 7 (narrow: + 15), bytes [bx0*24, bx0*24 + row_bytes) of each row
    const int tile_row_bytes = narrow ? kNarrowRowBytes : kBlocksPerWave * 24, lds_row_bytes = narrow ? kNarrowRowBytes : kOutRowBytes;
    const int row_bytes = min((W - bx0 * 8) * 3, tile_row_bytes);
    const int nrows = min(min(narrow ? 16 : 8, H - y0w), (bh - by_wave) * 8);
    const int chunks_per_row = tile_row_bytes >> 4;  // 48 or 24
    const int recip = narrow ? 2731 : 1366;          // g / chunks_per_row == (g * recip) >> 16 for g < 384
    uint8_t* out_base = im.out[0] + (size_t)y0w * im.out_pitch[0] + (size_t)bx0 * 24;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int g = k * 64 + lane;  // 16-byte chunk of the 6,144-byte tile
        const int r = (g * recip) >> 16, off = (g - r * chunks_per_row) * 16;
        if (r < nrows && off < row_bytes) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(lds_wave + r * lds_row_bytes + off);
            uint8_t* dst = out_base + (__umul24((unsigned)r, im.out_pitch[0]) + (unsigned)off);
            if (off + 16 <= row_bytes) {
#ifdef HJ_ABLATE_STORE
                if (v.x == 0x12345678u && v.y == 0x9abcdef0u)  // practically never: keeps the value live, drops the traffic
#endif
                typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;
#ifdef HJ_DEC_PLAIN_STORES
                *(HJ_GLOBAL u32x4_unaligned*)dst = v;
#else
                __builtin_nontemporal_store(v, (HJ_GLOBAL u32x4_unaligned*)dst);
#endif
            } else {
                const unsigned w[4] = {v.x, v.y, v.z, v.w};
                for (int j = 0; j < row_bytes - off; j++) dst[j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
            }
        }
    }
}

template <int HS, int VS, int LAYOUT>
__global__ __launch_bounds__(kThreads, HJ_MIN_WAVES_LUMA) void luma_color_kernel(const DecodeImage* __restrict__ images, const WorkUnit* __restrict__ units)
{
    __shared__ __attribute__((aligned(16))) char lds[4 * kLdsLumaWaveBytes];
    const WorkUnit u = units[blockIdx.x];
    luma_color_body<HS, VS, LAYOUT>(images[u.image], u, lds);
}

#ifndef HJ_MIN_WAVES_FUSED
#define HJ_MIN_WAVES_FUSED 4
#endif
template <int HS, int VS, int LAYOUT>
__global__ __launch_bounds__(kThreads, HJ_MIN_WAVES_FUSED) void luma_color_fused_kernel(const DecodeImage* __restrict__ images, const WorkUnit* __restrict__ units,
                                                                                        HuffImage* __restrict__ himages)
{
    __shared__ __attribute__((aligned(16))) char lds[4 * kLdsLumaWaveBytes];
    __shared__ FusedShared fs;
    extern __shared__ uint16_t dyn_pool[];
    const WorkUnit u = units[blockIdx.x];
    const DecodeImage& im = images[u.image];
    HuffImage& hi = himages[im.huff_index];
    fused_stage(hi, (HJ_LDS uint16_t*)dyn_pool, fs);
    luma_color_body<HS, VS, LAYOUT, true>(im, u, lds, &hi, (HJ_LDS uint16_t*)dyn_pool, &fs);
}

// ------------------------------------------------------------------------------------------------
// K3: generic colour stage for layouts the fused kernel does not cover (4:1:1, 4:1:0, ...): every component is in a
// plane already; libjpeg upsamples those by plain replication (jdsample.c int_upsample).  One thread = one pixel.
// The unit table is reused: block_base = pixel row, comp unused.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void generic_color_kernel(const DecodeImage* __restrict__ images, const WorkUnit* __restrict__ units)
{
    const WorkUnit u = units[blockIdx.x];
    const DecodeImage& im = images[u.image];
    const int W = im.width, H = im.height;
    const int y = u.block_base;
    if (y >= H) return;
    const int fmt = im.out_format;
    const bool planar = fmt == kOutPlanarRGB || fmt == kOutPlanarBGR;
    const bool bgr = fmt == kOutInterleavedBGR || fmt == kOutPlanarBGR;
    // Only three-component images take this path (gray goes through luma_color_kernel<0,0>), so the component
    // index is a compile-time constant everywhere below.
    const int hmax = im.hmax, vmax = im.vmax;
    const uint8_t* row0 = im.comp[0].plane + (size_t)(y * im.comp[0].v / vmax) * im.comp[0].plane_pitch;
    const uint8_t* row1 = im.comp[1].plane + (size_t)(y * im.comp[1].v / vmax) * im.comp[1].plane_pitch;
    const uint8_t* row2 = im.comp[2].plane + (size_t)(y * im.comp[2].v / vmax) * im.comp[2].plane_pitch;
    const int h0 = im.comp[0].h, h1 = im.comp[1].h, h2 = im.comp[2].h;
    for (int x = threadIdx.x; x < W; x += kThreads) {
        int s[3];
        s[0] = row0[x * h0 / hmax];
        s[1] = row1[x * h1 / hmax];
        s[2] = row2[x * h2 / hmax];
        int R, G, B;
        if (im.color_model == 1) {
            int rr = (s[2] * 91881 + (32768 - 128 * 91881)) >> 16;
            int bb = (s[1] * 116130 + (32768 - 128 * 116130)) >> 16;
            int gg = (s[1] * -22554 + s[2] * -46802 + (32768 + 128 * 22554 + 128 * 46802)) >> 16;
            R = min(max(s[0] + rr, 0), 255);
            G = min(max(s[0] + gg, 0), 255);
            B = min(max(s[0] + bb, 0), 255);
        } else {
            R = s[0];
            G = s[1];
            B = s[2];
        }
        if (bgr) {
            int t = R;
            R = B;
            B = t;
        }
        if (planar) {
            im.out[0][(size_t)y * im.out_pitch[0] + x] = (uint8_t)R;
            im.out[1][(size_t)y * im.out_pitch[1] + x] = (uint8_t)G;
            im.out[2][(size_t)y * im.out_pitch[2] + x] = (uint8_t)B;
        } else {
            uint8_t* p = im.out[0] + (size_t)y * im.out_pitch[0] + (size_t)x * 3;
            p[0] = (uint8_t)R;
            p[1] = (uint8_t)G;
            p[2] = (uint8_t)B;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Four-component frames (CMYK, YCCK): every component sits in its plane; the kernel brings each to full size the way
// libjpeg-turbo does (jdsample.c: h2v1 / h2v2 / h1v2 triangle filters when fancy upsampling is on and they apply, else
// replication), turns a YCCK frame into CMYK (jdcolor.c ycck_cmyk_convert) and then applies the reference extension's own
// CMYK -> RGB step (extensions/libjpeg_turbo/jpeg_mem.cpp:292-337): with an Adobe marker r = k*c/255, without one
// r = (255-k)*(255-c)/255; P_Y = (uint8)(0.299f r + 0.587f g + 0.114f b).  One thread = one pixel; rare path.
// ------------------------------------------------------------------------------------------------
template <int C>
__device__ __forceinline__ int upsampled_sample(const DecodeImage& im, int x, int y, bool fancy)
{
    const DecodeComponent& k = im.comp[C];
    const int fx = (int)im.hmax / (int)k.h, fy = (int)im.vmax / (int)k.v;
    const int dw = k.samp_w, dh = k.samp_h;
    const uint8_t* plane = k.plane;
    const size_t pitch = k.plane_pitch;
    if (fx == 1 && fy == 1) return plane[(size_t)y * pitch + x];
    if (fx == 2 && fy == 1 && fancy && dw > 2) {
        const uint8_t* p = plane + (size_t)y * pitch;
        const int i = x >> 1;
        if (x == 0) return p[0];
        if (x == 2 * dw - 1) return p[dw - 1];
        return (x & 1) ? (3 * p[i] + p[i + 1] + 2) >> 2 : (3 * p[i] + p[i - 1] + 1) >> 2;
    }
    if (fx == 2 && fy == 2 && fancy && dw > 2) {
        const int r0 = y >> 1, r1 = min(max((y & 1) ? r0 + 1 : r0 - 1, 0), dh - 1);
        const uint8_t* p0 = plane + (size_t)r0 * pitch;
        const uint8_t* p1 = plane + (size_t)r1 * pitch;
        const int i = x >> 1, cs = 3 * p0[i] + p1[i];
        if (x == 0) return (cs * 4 + 8) >> 4;
        if (x == 2 * dw - 1) return (cs * 4 + 7) >> 4;
        return (x & 1) ? (3 * cs + (3 * p0[i + 1] + p1[i + 1]) + 7) >> 4 : (3 * cs + (3 * p0[i - 1] + p1[i - 1]) + 8) >> 4;
    }
    if (fx == 1 && fy == 2 && fancy) {
        const int r0 = y >> 1, r1 = min(max((y & 1) ? r0 + 1 : r0 - 1, 0), dh - 1);
        return (3 * plane[(size_t)r0 * pitch + x] + plane[(size_t)r1 * pitch + x] + ((y & 1) ? 2 : 1)) >> 2;
    }
    return plane[(size_t)(y / fy) * pitch + x / fx];
}

__global__ __launch_bounds__(kThreads) void cmyk_color_kernel(const DecodeImage* __restrict__ images, const WorkUnit* __restrict__ units)
{
    const WorkUnit u = units[blockIdx.x];
    const DecodeImage& im = images[u.image];
    const int W = im.width, H = im.height;
    const int y = u.block_base;
    if (y >= H) return;
    const int fmt = im.out_format;
    const bool planar = fmt == kOutPlanarRGB || fmt == kOutPlanarBGR;
    const bool bgr = fmt == kOutInterleavedBGR || fmt == kOutPlanarBGR;
    const bool fancy = (im.flags & kFlagFancyUpsampling) != 0, adobe = (im.flags & kFlagAdobeMarker) != 0;
    const bool ycck = im.color_model == 4;
    for (int x = threadIdx.x; x < W; x += kThreads) {
        int c = upsampled_sample<0>(im, x, y, fancy), m = upsampled_sample<1>(im, x, y, fancy), ye = upsampled_sample<2>(im, x, y, fancy);
        const int k = upsampled_sample<3>(im, x, y, fancy);
        if (ycck) {
            const int rr = (ye * 91881 + (32768 - 128 * 91881)) >> 16;
            const int bb = (m * 116130 + (32768 - 128 * 116130)) >> 16;
            const int gg = (m * -22554 + ye * -46802 + (32768 + 128 * 22554 + 128 * 46802)) >> 16;
            const int r = min(max(c + rr, 0), 255), g = min(max(c + gg, 0), 255), b = min(max(c + bb, 0), 255);
            c = 255 - r;
            m = 255 - g;
            ye = 255 - b;
        }
        int R, G, B;
        if (adobe) {
            R = (k * c) / 255;
            G = (k * m) / 255;
            B = (k * ye) / 255;
        } else {
            R = (255 - k) * (255 - c) / 255;
            G = (255 - k) * (255 - m) / 255;
            B = (255 - k) * (255 - ye) / 255;
        }
        if (fmt == kOutY) {
            // three products and two sums, each rounded to float on its own, as the reference's x86 build computes them
            // (a fused multiply-add would round differently)
            float yf;
            {
#pragma clang fp contract(off)
                const float pr = 0.299f * (float)R, pg = 0.587f * (float)G, pb = 0.114f * (float)B;
                yf = (pr + pg) + pb;
            }
            im.out[0][(size_t)y * im.out_pitch[0] + x] = (uint8_t)yf;
            continue;
        }
        if (bgr) {
            const int t = R;
            R = B;
            B = t;
        }
        if (planar) {
            im.out[0][(size_t)y * im.out_pitch[0] + x] = (uint8_t)R;
            im.out[1][(size_t)y * im.out_pitch[1] + x] = (uint8_t)G;
            im.out[2][(size_t)y * im.out_pitch[2] + x] = (uint8_t)B;
        } else {
            uint8_t* p = im.out[0] + (size_t)y * im.out_pitch[0] + (size_t)x * 3;
            p[0] = (uint8_t)R;
            p[1] = (uint8_t)G;
            p[2] = (uint8_t)B;
        }
    }
}


// Geometry pass: region of interest + EXIF orientation (DecodeBatch plans it for the images that ask for it; the pixel
// kernels have written those images into an intermediate buffer in stored-image coordinates).
// Upright pixel (ox, oy) comes from region pixel (u, v):
//   1: (ox, oy)            2: (rw-1-ox, oy)        3: (rw-1-ox, rh-1-oy)   4: (ox, rh-1-oy)
//   5: (oy, ox)            6: (oy, rh-1-ox)        7: (rw-1-oy, rh-1-ox)   8: (rw-1-oy, ox)
// (EXIF 2.32 orientation tag; 5 = transpose, 6 = stored picture must be turned 90 degrees clockwise, 7 = transverse,
//  8 = 270 degrees clockwise).  One workgroup = kTransformRowsPerUnit output rows; lanes walk along output rows, so the
// writes are coalesced whatever the reads look like.
__global__ __launch_bounds__(kThreads) void transform_kernel(const TransformImage* __restrict__ images, const WorkUnit* __restrict__ units)
{
    const WorkUnit wu = units[blockIdx.x];
    const TransformImage& t = images[wu.image];
    const int rw = t.rw, rh = t.rh, ow = t.out_w, o = t.orientation;
    const int row_end = min((int)wu.block_base + kTransformRowsPerUnit, t.out_h);
    for (int oy = (int)wu.block_base; oy < row_end; oy++) {
        for (int ox = threadIdx.x; ox < ow; ox += kThreads) {
            int u, v;
            switch (o) {
            case 2: u = rw - 1 - ox; v = oy; break;
            case 3: u = rw - 1 - ox; v = rh - 1 - oy; break;
            case 4: u = ox; v = rh - 1 - oy; break;
            case 5: u = oy; v = ox; break;
            case 6: u = oy; v = rh - 1 - ox; break;
            case 7: u = rw - 1 - oy; v = rh - 1 - ox; break;
            case 8: u = rw - 1 - oy; v = ox; break;
            default: u = ox; v = oy; break;
            }
            const int sx = t.x0 + u, sy = t.y0 + v;
            if (t.bpp == 3) {
                const uint8_t* s = t.src[0] + (size_t)sy * t.src_pitch[0] + (size_t)sx * 3;
                uint8_t* d = t.dst[0] + (size_t)oy * t.dst_pitch[0] + (size_t)ox * 3;
                d[0] = s[0];
                d[1] = s[1];
                d[2] = s[2];
            } else {
                for (int p = 0; p < t.nplanes; p++)
                    t.dst[p][(size_t)oy * t.dst_pitch[p] + ox] = t.src[p][(size_t)sy * t.src_pitch[p] + sx];
            }
        }
    }
}

}  // namespace

int launch_idct_plane(const DecodeImage* images, const WorkUnit* units, int nunits, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(idct_plane_kernel, dim3(nunits), dim3(kThreads), 0, (hipStream_t)stream, images, units);
    return (int)hipGetLastError();
}

int launch_idct_plane_fused(const DecodeImage* images, const WorkUnit* units, int nunits, HuffImage* himages, unsigned pool_bytes, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(idct_plane_fused_kernel, dim3(nunits), dim3(kThreads), pool_bytes, (hipStream_t)stream, images, units, himages);
    return (int)hipGetLastError();
}

template <int LAYOUT>
static int launch_luma_fused_t(int hs, int vs, const DecodeImage* images, const WorkUnit* units, int nunits, HuffImage* himages, unsigned pool_bytes, hipStream_t s)
{
    if (hs == 0) {
        if constexpr (LAYOUT != kLayoutAny) return (int)hipErrorInvalidValue;
        else hipLaunchKernelGGL((luma_color_fused_kernel<0, 0, kLayoutAny>), dim3(nunits), dim3(kThreads), pool_bytes, s, images, units, himages);
    } else if (hs == 1 && vs == 1)
        hipLaunchKernelGGL((luma_color_fused_kernel<1, 1, LAYOUT>), dim3(nunits), dim3(kThreads), pool_bytes, s, images, units, himages);
    else if (hs == 2 && vs == 1)
        hipLaunchKernelGGL((luma_color_fused_kernel<2, 1, LAYOUT>), dim3(nunits), dim3(kThreads), pool_bytes, s, images, units, himages);
    else if (hs == 2 && vs == 2)
        hipLaunchKernelGGL((luma_color_fused_kernel<2, 2, LAYOUT>), dim3(nunits), dim3(kThreads), pool_bytes, s, images, units, himages);
    else if (hs == 1 && vs == 2)
        hipLaunchKernelGGL((luma_color_fused_kernel<1, 2, LAYOUT>), dim3(nunits), dim3(kThreads), pool_bytes, s, images, units, himages);
    else
        return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

int launch_luma_color_fused(int layout, int hs, int vs, const DecodeImage* images, const WorkUnit* units, int nunits, HuffImage* himages, unsigned pool_bytes,
                            void* stream)
{
    if (nunits <= 0) return 0;
    switch (layout) {
    case 0: return launch_luma_fused_t<kLayoutAny>(hs, vs, images, units, nunits, himages, pool_bytes, (hipStream_t)stream);
    case 1: return launch_luma_fused_t<kLayoutInterleaved>(hs, vs, images, units, nunits, himages, pool_bytes, (hipStream_t)stream);
    case 2: return launch_luma_fused_t<kLayoutPlanar>(hs, vs, images, units, nunits, himages, pool_bytes, (hipStream_t)stream);
    default: return (int)hipErrorInvalidValue;
    }
}

template <int LAYOUT>
static int launch_luma_color_t(int hs, int vs, const DecodeImage* images, const WorkUnit* units, int nunits, hipStream_t s)
{
    if (hs == 0) {
        if constexpr (LAYOUT != kLayoutAny) return (int)hipErrorInvalidValue;  // gray sources have no colour conversion to specialise
        else hipLaunchKernelGGL((luma_color_kernel<0, 0, kLayoutAny>), dim3(nunits), dim3(kThreads), 0, s, images, units);
    } else if (hs == 1 && vs == 1)
        hipLaunchKernelGGL((luma_color_kernel<1, 1, LAYOUT>), dim3(nunits), dim3(kThreads), 0, s, images, units);
    else if (hs == 2 && vs == 1)
        hipLaunchKernelGGL((luma_color_kernel<2, 1, LAYOUT>), dim3(nunits), dim3(kThreads), 0, s, images, units);
    else if (hs == 2 && vs == 2)
        hipLaunchKernelGGL((luma_color_kernel<2, 2, LAYOUT>), dim3(nunits), dim3(kThreads), 0, s, images, units);
    else if (hs == 1 && vs == 2)
        hipLaunchKernelGGL((luma_color_kernel<1, 2, LAYOUT>), dim3(nunits), dim3(kThreads), 0, s, images, units);
    else
        return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

int launch_luma_color(int layout, int hs, int vs, const DecodeImage* images, const WorkUnit* units, int nunits, void* stream)
{
    if (nunits <= 0) return 0;
    switch (layout) {
    case 0: return launch_luma_color_t<kLayoutAny>(hs, vs, images, units, nunits, (hipStream_t)stream);
    case 1: return launch_luma_color_t<kLayoutInterleaved>(hs, vs, images, units, nunits, (hipStream_t)stream);
    case 2: return launch_luma_color_t<kLayoutPlanar>(hs, vs, images, units, nunits, (hipStream_t)stream);
    default: return (int)hipErrorInvalidValue;
    }
}

int launch_generic_color(const DecodeImage* images, const WorkUnit* units, int nunits, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(generic_color_kernel, dim3(nunits), dim3(kThreads), 0, (hipStream_t)stream, images, units);
    return (int)hipGetLastError();
}

int launch_cmyk_color(const DecodeImage* images, const WorkUnit* units, int nunits, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(cmyk_color_kernel, dim3(nunits), dim3(kThreads), 0, (hipStream_t)stream, images, units);
    return (int)hipGetLastError();
}

int launch_transform(const TransformImage* images, const WorkUnit* units, int nunits, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(transform_kernel, dim3(nunits), dim3(kThreads), 0, (hipStream_t)stream, images, units);
    return (int)hipGetLastError();
}

}  // namespace hipjpeg
