// encode_kernels.hip -- gfx950 kernel of the JPEG encode device stage: colour conversion + chroma downsampling +
// forward DCT + quantization, i.e. what nvjpegEncodeImage does on the GPU for the reference
// (extensions/nvjpeg/cuda_encoder.cpp:362-372).  Huffman coding and marker writing stay on the host (entropy_encode.cpp).
//
// Mapping: a workgroup owns a tile of 32x8 luma blocks (256x64 pixels); one LANE owns one 8x8 pixel block.
//   phase A  each lane loads its 8x8 RGB pixels (24-byte rows, coalesced across the 32 lanes of a tile row), converts to
//            YCbCr with libjpeg's 16-bit fixed-point weights, runs the luma FDCT + quantizer in registers, and
//            box-downsamples its Cb/Cr patch into an LDS chroma tile;
//   phase B  lanes pick up whole 8x8 chroma blocks from that LDS tile, FDCT + quantize them.
// Quantized blocks are staged through LDS so a wave writes 8 KB of coefficients contiguously (16 B per lane).
// Coefficients are written in ZIGZAG order (what the host Huffman coder consumes sequentially).
//
// Arithmetic restates libjpeg-turbo's jccolor.c / jcsample.c / jfdctint.c / jcdctmgr.c; compared bit-for-bit with the
// CPU oracle (and, through it, with libjpeg-turbo's bitstreams) in tests/.
#include <hip/hip_runtime.h>

#include "encode_kernels.h"
#include "encode_layout.h"

namespace hipjpeg {

namespace {

constexpr int kThreads = 256;
constexpr int kTileBX = 32, kTileBY = 8;  // luma blocks per workgroup tile
constexpr int kLdsBlockStride = 144;      // 128 B block + 16 B pad (same conflict-free stride as the decoder)

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;

constexpr int F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633;
constexpr int F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;

// zigzag index -> natural index (compile-time: every use below has a constant subscript after unrolling)
__device__ constexpr int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                        41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                        15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// jfdctint.c, one 1-D pass.  PASS1: outputs scaled up by 2^PASS1_BITS (rows); else final descale (columns).
// Inputs are |x| <= 128 (pass 1) or <= 2^13 (pass 2) so every product fits the 24-bit multiplier exactly.
template <bool PASS1>
__device__ __forceinline__ void fdct8(int (&d)[8])
{
    int t0 = d[0] + d[7], t7 = d[0] - d[7];
    int t1 = d[1] + d[6], t6 = d[1] - d[6];
    int t2 = d[2] + d[5], t5 = d[2] - d[5];
    int t3 = d[3] + d[4], t4 = d[3] - d[4];
    int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    constexpr int S = PASS1 ? 11 : 15;  // CONST_BITS - PASS1_BITS / CONST_BITS + PASS1_BITS
    if (PASS1) {
        d[0] = (t10 + t11) << 2;
        d[4] = (t10 - t11) << 2;
    } else {
        d[0] = descale(t10 + t11, 2);
        d[4] = descale(t10 - t11, 2);
    }
    int z1 = __mul24(t12 + t13, F_0_541);
    d[2] = descale(z1 + __mul24(t13, F_0_765), S);
    d[6] = descale(z1 + __mul24(t12, -F_1_847), S);
    z1 = t4 + t7;
    int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    int z5 = __mul24(z3 + z4, F_1_175);
    t4 = __mul24(t4, F_0_298);
    t5 = __mul24(t5, F_2_053);
    t6 = __mul24(t6, F_3_072);
    t7 = __mul24(t7, F_1_501);
    z1 = __mul24(z1, -F_0_899);
    z2 = __mul24(z2, -F_2_562);
    z3 = __mul24(z3, -F_1_961) + z5;
    z4 = __mul24(z4, -F_0_390) + z5;
    d[7] = descale(t4 + z1 + z3, S);
    d[5] = descale(t5 + z2 + z4, S);
    d[3] = descale(t6 + z2 + z3, S);
    d[1] = descale(t7 + z1 + z4, S);
}

// s[r][c] = level-shifted samples (sample - 128).  Result: quantized coefficients packed in zigzag order, 2 per dword.
__device__ __forceinline__ void fdct_quantize(int (&s)[8][8], const EncodeQuant& q, u32x4 (&packed)[8])
{
#pragma unroll
    for (int r = 0; r < 8; r++) fdct8<true>(s[r]);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        int col[8];
#pragma unroll
        for (int r = 0; r < 8; r++) col[r] = s[r][c];
        fdct8<false>(col);
#pragma unroll
        for (int r = 0; r < 8; r++) s[r][c] = col[r];
    }
    // jcdctmgr.c quantize: divisor 8*q, round half away from zero.  The division is a multiply-high by a per-entry magic
    // number prepared on the host; exact for every numerator below 2^17 (FDCT output + half divisor is below 2^16).
    int zz[64];
#pragma unroll
    for (int k = 0; k < 64; k++) {
        const int n = kZigzag[k];
        const int v = s[n >> 3][n & 7];
        const unsigned a = (unsigned)(v < 0 ? -v : v) + q.half[k];
        const int m = (int)__umulhi(a << 4, q.magic[k]);
        zz[k] = v < 0 ? -m : m;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        packed[j].x = (unsigned)(zz[8 * j + 0] & 0xFFFF) | ((unsigned)zz[8 * j + 1] << 16);
        packed[j].y = (unsigned)(zz[8 * j + 2] & 0xFFFF) | ((unsigned)zz[8 * j + 3] << 16);
        packed[j].z = (unsigned)(zz[8 * j + 4] & 0xFFFF) | ((unsigned)zz[8 * j + 5] << 16);
        packed[j].w = (unsigned)(zz[8 * j + 6] & 0xFFFF) | ((unsigned)zz[8 * j + 7] << 16);
    }
}

// Lane writes its block to its slot of the wave's LDS staging area; after the barrier the wave streams the 64 slots out
// to `dst` (consecutive blocks of one block row) with coalesced 16-byte stores.  Slots flagged invalid are skipped.
__device__ __forceinline__ void stage_block(char* lds_wave, int lane, const u32x4 (&packed)[8])
{
#pragma unroll
    for (int j = 0; j < 8; j++) *reinterpret_cast<u32x4*>(lds_wave + lane * kLdsBlockStride + j * 16) = packed[j];
}

__global__ __launch_bounds__(kThreads) void forward_kernel(const EncodeImage* __restrict__ images, const EncodeUnit* __restrict__ units)
{
    __shared__ __attribute__((aligned(16))) char lds_coef[4 * 64 * kLdsBlockStride];  // 36,864 B
    __shared__ __attribute__((aligned(16))) unsigned char lds_chroma[2][kTileBY * 8][kTileBX * 8 + 16];  // up to 4:4:4: 2 x 64 x 272 B

    const EncodeUnit u = units[blockIdx.x];
    const EncodeImage& im = images[u.image];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = im.width, H = im.height;
    const int hs = im.hs, vs = im.vs;  // luma sampling factors = chroma downsampling factors
    const bool color = im.ncomp == 3;

    // ---- phase A: the lane's luma block.  Tile rows are 32 blocks wide: lanes 0..31 of a wave = one block row.
    const int lbx = tid & (kTileBX - 1), lby = tid >> 5;
    const int bx = u.tile_bx * kTileBX + lbx, by = u.tile_by * kTileBY + lby;
    const int x0 = bx * 8, y0 = by * 8;
    const int fmt = im.in_format;
    const bool planar = fmt == kInPlanarRGB || fmt == kInPlanarBGR;
    const bool bgr = fmt == kInInterleavedBGR || fmt == kInPlanarBGR;
    const bool interior = x0 + 8 <= W;

    int ys[8][8];
    // chroma accumulators for this lane's patch: (8/hs) x (8/vs) samples per component, kept as running sums
    const int cw = 8 / hs, ch = 8 / vs;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int y = min(y0 + r, H - 1);  // rows past the image replicate the last row (jcprepct.c expand_bottom_edge)
        int R[8], G[8], B[8];
        if (fmt == kInGray) {
            const unsigned char* p = im.in[0] + (size_t)y * im.in_pitch[0];
#pragma unroll
            for (int c = 0; c < 8; c++) R[c] = G[c] = B[c] = p[min(x0 + c, W - 1)];
        } else if (planar) {
            const unsigned char* p0 = im.in[0] + (size_t)y * im.in_pitch[0];
            const unsigned char* p1 = im.in[1] + (size_t)y * im.in_pitch[1];
            const unsigned char* p2 = im.in[2] + (size_t)y * im.in_pitch[2];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int x = min(x0 + c, W - 1);
                R[c] = p0[x];
                G[c] = p1[x];
                B[c] = p2[x];
            }
        } else {
            const unsigned char* p = im.in[0] + (size_t)y * im.in_pitch[0];
            const unsigned char* q = p + (size_t)x0 * 3;
            if (interior && ((uintptr_t)q & 7) == 0) {
                const uint2* v = reinterpret_cast<const uint2*>(q);
                const uint2 a = v[0], b = v[1], cc = v[2];
                const unsigned w[6] = {a.x, a.y, b.x, b.y, cc.x, cc.y};
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    R[c] = (w[(3 * c) >> 2] >> (8 * ((3 * c) & 3))) & 0xFF;
                    G[c] = (w[(3 * c + 1) >> 2] >> (8 * ((3 * c + 1) & 3))) & 0xFF;
                    B[c] = (w[(3 * c + 2) >> 2] >> (8 * ((3 * c + 2) & 3))) & 0xFF;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int x = min(x0 + c, W - 1);  // columns past the image replicate the last pixel (expand_right_edge)
                    R[c] = p[3 * x];
                    G[c] = p[3 * x + 1];
                    B[c] = p[3 * x + 2];
                }
            }
        }
        if (bgr) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int t = R[c];
                R[c] = B[c];
                B[c] = t;
            }
        }
        int cb[8], cr[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            // jccolor.c rgb_ycc_convert, SCALEBITS 16.  For gray input R=G=B so Y == the sample (weights sum to 65536).
            ys[r][c] = ((__mul24(R[c], 19595) + __mul24(G[c], 38470) + __mul24(B[c], 7471) + 32768) >> 16) - 128;
            cb[c] = (__mul24(R[c], -11059) + __mul24(G[c], -21709) + __mul24(B[c], 32768) + ((128 << 16) + 32767)) >> 16;
            cr[c] = (__mul24(R[c], 32768) + __mul24(G[c], -27439) + __mul24(B[c], -5329) + ((128 << 16) + 32767)) >> 16;
        }
        if (color) {
            // full-resolution chroma row into the LDS tile; the downsample happens when phase B gathers (keeps every
            // sampling layout on one code path)
            unsigned char* d0 = &lds_chroma[0][lby * 8 + r][lbx * 8];
            unsigned char* d1 = &lds_chroma[1][lby * 8 + r][lbx * 8];
            *reinterpret_cast<uint2*>(d0) = make_uint2((unsigned)cb[0] | (cb[1] << 8) | (cb[2] << 16) | ((unsigned)cb[3] << 24),
                                                       (unsigned)cb[4] | (cb[5] << 8) | (cb[6] << 16) | ((unsigned)cb[7] << 24));
            *reinterpret_cast<uint2*>(d1) = make_uint2((unsigned)cr[0] | (cr[1] << 8) | (cr[2] << 16) | ((unsigned)cr[3] << 24),
                                                       (unsigned)cr[4] | (cr[5] << 8) | (cr[6] << 16) | ((unsigned)cr[7] << 24));
        }
    }
    (void)cw;
    (void)ch;

    // luma FDCT + quantize; real blocks only (dummy blocks completing the last MCU are synthesized by the host coder)
    {
        u32x4 packed[8];
        fdct_quantize(ys, im.quant[0], packed);
        stage_block(lds_coef + wave * 64 * kLdsBlockStride, lane, packed);
    }
    __syncthreads();
    {
        // wave `wave` holds tile block rows 2*wave and 2*wave+1 (32 blocks each = 4 KB contiguous in the luma grid)
        const int real_w = im.real_w[0], real_h = im.real_h[0];
        int16_t* base = im.coef[0];
        const char* lw = lds_coef + wave * 64 * kLdsBlockStride;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int g = k * 64 + lane;     // 16-byte chunk index within the wave's 64 blocks
            const int blk = g >> 3;          // 0..63
            const int row = 2 * wave + (blk >> 5), col = blk & 31;
            const int gbx = u.tile_bx * kTileBX + col, gby = u.tile_by * kTileBY + row;
            if (gbx < real_w && gby < real_h) {
                u32x4 v = *reinterpret_cast<const u32x4*>(lw + blk * kLdsBlockStride + (g & 7) * 16);
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(base + ((size_t)gby * im.blocks_w[0] + gbx) * 64) + (g & 7));
            }
        }
    }
    if (!color) return;

    // ---- phase B: chroma blocks of the tile.  Tile chroma grid: (32/hs) x (8/vs) blocks per component.
    const int cbw = kTileBX / hs, cbh = kTileBY / vs;  // chroma blocks per tile row / column
    const int nchroma = cbw * cbh * 2;
    const int last_row = (H + vs - 1) / vs - 1;  // last real downsampled row (rows below replicate it: jcprepct.c)
    for (int base_idx = 0; base_idx < nchroma; base_idx += kThreads) {
        const int idx = base_idx + tid;
        const bool active = idx < nchroma;
        const int comp = active ? idx / (cbw * cbh) : 0;  // 0 = Cb, 1 = Cr
        const int rem = active ? idx - comp * (cbw * cbh) : 0;
        const int cby = rem / cbw, cbx = rem - cby * cbw;
        const int gcx = u.tile_bx * cbw + cbx, gcy = u.tile_by * cbh + cby;
        __syncthreads();  // staging area reuse
        if (active) {
            int s[8][8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                // global downsampled row, clamped to the last real one; then back to tile-local full-res rows
                int grow = min(gcy * 8 + r, last_row);
                int lrow = grow * vs - u.tile_by * kTileBY * 8;  // first full-res source row inside the tile
                lrow = max(lrow, 0);
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int lx = (cbx * 8 + c) * hs;  // first full-res source column inside the tile
                    int v;
                    if (hs == 1 && vs == 1) {
                        v = lds_chroma[comp][lrow][lx];
                    } else if (hs == 2 && vs == 1) {
                        v = (lds_chroma[comp][lrow][lx] + lds_chroma[comp][lrow][lx + 1] + (c & 1)) >> 1;  // h2v1_downsample bias 0,1,0,1
                    } else if (hs == 2 && vs == 2) {
                        v = (lds_chroma[comp][lrow][lx] + lds_chroma[comp][lrow][lx + 1] + lds_chroma[comp][lrow + 1][lx] +
                             lds_chroma[comp][lrow + 1][lx + 1] + 1 + (c & 1)) >> 2;  // h2v2_downsample bias 1,2,1,2
                    } else {
                        int sum = 0;  // int_downsample: box average, rounding at numpix/2
                        for (int j = 0; j < vs; j++)
                            for (int i = 0; i < hs; i++) sum += lds_chroma[comp][lrow + j][lx + i];
                        v = (sum + (hs * vs) / 2) / (hs * vs);
                    }
                    s[r][c] = v - 128;
                }
            }
            u32x4 packed[8];
            fdct_quantize(s, im.quant[1], packed);
            // chroma blocks are few: write them straight out (16 B pieces, one block per lane)
            if (gcx < (int)im.real_w[1 + comp] && gcy < (int)im.real_h[1 + comp]) {
                u32x4* dst = reinterpret_cast<u32x4*>(im.coef[1 + comp] + ((size_t)gcy * im.blocks_w[1 + comp] + gcx) * 64);
#pragma unroll
                for (int j = 0; j < 8; j++) dst[j] = packed[j];
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// forward_pair_kernel: the same arithmetic with TWO LANES per 8x8 block (the decoder's mapping, decode_kernels.hip) for the
// inputs nearly every caller has: interleaved RGB/BGR, three components, 4:2:0 / 4:2:2 / 4:4:4.  The one-lane-per-block
// kernel above holds a whole block in registers (240 VGPRs, two waves per SIMD) and stays for every other layout.
//   lane p of a pair loads and colour-converts four pixel rows of the block -- lane 0 rows 0,1,2,3, lane 1 rows 7,6,5,4 --
//   downsamples its chroma patch into the LDS chroma tile, runs the FDCT row pass on its four rows and leaves them in the
//   block's LDS slot as int16; after the hand-off it reads columns 4p..4p+3 of all eight rows back and runs the column pass on
//   int16 pairs (fdct8_pk16).  Quantized coefficients go back into the slot in natural order; the wave's copy-out gathers them in
//   zigzag order (each lane always fetches the same eight positions) and stores 16 bytes per lane, fully coalesced.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pair_lds_fence()
{
    // LDS operations of one wave execute in order; only the compiler has to be kept from reordering across the hand-off
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ unsigned pack16(int a, int b) { return __builtin_amdgcn_perm((unsigned)b, (unsigned)a, 0x05040100u); }  // a[15:0] | b[15:0] << 16
__device__ __forceinline__ unsigned hi16_pair(int a, int b) { return __builtin_amdgcn_perm((unsigned)b, (unsigned)a, 0x07060302u); }  // a[31:16] | b[31:16] << 16

using lds_char = __attribute__((address_space(3))) char;

// ---- FDCT passes on int16 pairs ------------------------------------------------------------------------------------------
// jfdctint.c's pass 2 is a linear map with integer coefficients in front of its descale (every input meets ONE constant on the
// way to an output), and its inputs -- the row pass's outputs, at most 8 x 128 x 4 = 4,096 in magnitude -- always fit int16 (so
// do the sums and differences of two of them: the libjpeg SIMD builds rely on the same bound).  So: the mirrored sums and
// differences t0..t7 as packed 16-bit adds on pairs of rows, then every output as two v_dot2_i32_i16 (two products and the
// accumulate, rounding term included, in one instruction) -- 4 + 4 + 16 + 8 shifts = 32 instructions per column where the
// butterfly form takes 54 with its unpacking.  Exact: regrouping integer sums does not change them (mod 2^32, no overflow here).
using i16x2 = __attribute__((ext_vector_type(2))) short;
constexpr unsigned pk16(int lo, int hi) { return ((unsigned)lo & 0xFFFFu) | (((unsigned)hi & 0xFFFFu) << 16); }
__device__ __forceinline__ unsigned pk_add16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (i16x2)(__builtin_bit_cast(i16x2, a) + __builtin_bit_cast(i16x2, b))); }
__device__ __forceinline__ unsigned pk_sub16(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (i16x2)(__builtin_bit_cast(i16x2, a) - __builtin_bit_cast(i16x2, b))); }
// a.lo * k.lo + a.hi * k.hi + acc (three-operand form pinned: hipcc prefers v_dot2c, whose accumulator is the destination, and pays
// a v_mov per product for it)
__device__ __forceinline__ int dot2(unsigned a, unsigned k, int acc)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(acc));
    return r;
}
__device__ __forceinline__ int dot2_plus2(unsigned a, unsigned k)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 2" : "=v"(r) : "v"(a), "s"(k));
    return r;
}
// even outputs from (t0,t1), (t2,t3); odd outputs from (t7,t6), (t5,t4) -- fdct8's names
constexpr int kFa = F_0_541, kFb = F_0_541 + F_0_765, kFe = F_0_541 - F_1_847;
constexpr int kA44 = F_0_298 - F_0_899 - F_1_961 + F_1_175, kA45 = F_1_175, kA46 = F_1_175 - F_1_961, kA47 = F_1_175 - F_0_899;
constexpr int kA54 = F_1_175, kA55 = F_2_053 - F_2_562 - F_0_390 + F_1_175, kA56 = F_1_175 - F_2_562, kA57 = F_1_175 - F_0_390;
constexpr int kA64 = F_1_175 - F_1_961, kA65 = F_1_175 - F_2_562, kA66 = F_3_072 - F_2_562 - F_1_961 + F_1_175, kA67 = F_1_175;
constexpr int kA74 = F_1_175 - F_0_899, kA75 = F_1_175 - F_0_390, kA76 = F_1_175, kA77 = F_1_501 - F_0_899 - F_0_390 + F_1_175;

__device__ __forceinline__ int dot2_plus0(unsigned a, unsigned k)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "s"(k));
    return r;
}
// One 1-D pass from the four input pairs P0 = (x0,x1), P1 = (x2,x3), P2 = (x7,x6), P3 = (x5,x4): d[k] = output k.
// PASS1: the row pass (outputs scaled up by 2^PASS1_BITS, rnd = 1 << 10); else the column pass (final descale, rnd = 1 << 14).
template <bool PASS1>
__device__ __forceinline__ void fdct8_pk16(unsigned P0, unsigned P1, unsigned P2, unsigned P3, int rnd, int (&d)[8])
{
    constexpr int S = PASS1 ? 11 : 15;
    const unsigned S01 = pk_add16(P0, P2), D01 = pk_sub16(P0, P2);  // (t0,t1), (t7,t6)
    const unsigned S23 = pk_add16(P1, P3), D23 = pk_sub16(P1, P3);  // (t2,t3), (t5,t4)
    if constexpr (PASS1) {
        d[0] = dot2(S01, pk16(4, 4), dot2_plus0(S23, pk16(4, 4)));
        d[4] = dot2(S01, pk16(4, -4), dot2_plus0(S23, pk16(-4, 4)));
    } else {
        d[0] = dot2(S01, pk16(1, 1), dot2_plus2(S23, pk16(1, 1))) >> 2;
        d[4] = dot2(S01, pk16(1, -1), dot2_plus2(S23, pk16(-1, 1))) >> 2;
    }
    d[2] = dot2(S01, pk16(kFb, kFa), dot2(S23, pk16(-kFa, -kFb), rnd)) >> S;
    d[6] = dot2(S01, pk16(kFa, kFe), dot2(S23, pk16(-kFe, -kFa), rnd)) >> S;
    d[7] = dot2(D01, pk16(kA47, kA46), dot2(D23, pk16(kA45, kA44), rnd)) >> S;
    d[5] = dot2(D01, pk16(kA57, kA56), dot2(D23, pk16(kA55, kA54), rnd)) >> S;
    d[3] = dot2(D01, pk16(kA67, kA66), dot2(D23, pk16(kA65, kA64), rnd)) >> S;
    d[1] = dot2(D01, pk16(kA77, kA76), dot2(D23, pk16(kA75, kA74), rnd)) >> S;
}

// FDCT row pass of one block row from its sample pairs (level-shifted, int16) and hand-off: the row goes into the block's LDS slot
// as int16.
__device__ __forceinline__ void row_pass_store_pk16(unsigned P0, unsigned P1, unsigned P2, unsigned P3, lds_char* slot, int row)
{
    int s[8];
    fdct8_pk16<true>(P0, P1, P2, P3, 1 << 10, s);
    const u32x4 v = {pack16(s[0], s[1]), pack16(s[2], s[3]), pack16(s[4], s[5]), pack16(s[6], s[7])};
    *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(slot + row * 16) = v;
}

// After every row of the block is in the slot: column pass of columns 4p..4p+3, quantization, and the quantized coefficients back
// into the slot in natural order (int16).  The caller fences afterwards.
// qmagic / qhalf16: tables in LDS, COLUMN-major over the natural block (index column * 8 + row), half16 = half << 4.
__device__ __forceinline__ void column_pass_quantize(lds_char* slot, bool p, const __attribute__((address_space(3))) unsigned* qmagic,
                                                     const __attribute__((address_space(3))) unsigned* qhalf16)
{
    using lds_u32x2 = __attribute__((address_space(3))) u32x2;
    using lds_u32x4 = __attribute__((address_space(3))) u32x4;
    pair_lds_fence();
    u32x2 rd[8];
    lds_char* mine = slot + (p ? 8 : 0);
#pragma unroll
    for (int n = 0; n < 8; n++) rd[n] = *reinterpret_cast<const lds_u32x2*>(mine + n * 16);
    pair_lds_fence();  // the partner has read too before anything below overwrites the slot (same wave: program order)
    const int rnd15 = 1 << 14;
    // two columns at a time; the tables are stored column-major so the eight quantizers of a column are two 16-byte reads
#pragma unroll
    for (int half = 0; half < 2; half++) {
        int r[2][8];
        unsigned w[8];
#pragma unroll
        for (int n = 0; n < 8; n++) w[n] = half ? rd[n].y : rd[n].x;  // row n: column 4p + 2 half in the low, the next one in the high half
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            const unsigned sel = jj ? 0x07060302u : 0x05040100u;  // x.half | y.half << 16
            int d[8];
            fdct8_pk16<false>(__builtin_amdgcn_perm(w[1], w[0], sel), __builtin_amdgcn_perm(w[3], w[2], sel), __builtin_amdgcn_perm(w[6], w[7], sel),
                              __builtin_amdgcn_perm(w[4], w[5], sel), rnd15, d);
            const int tcol = ((p ? 4 : 0) + 2 * half + jj) * 8;
            const u32x4 ma = *reinterpret_cast<const lds_u32x4*>(qmagic + tcol), mb = *reinterpret_cast<const lds_u32x4*>(qmagic + tcol + 4);
            const u32x4 ha = *reinterpret_cast<const lds_u32x4*>(qhalf16 + tcol), hb = *reinterpret_cast<const lds_u32x4*>(qhalf16 + tcol + 4);
            const unsigned m8[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w}, h8[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                // jcdctmgr.c quantize: divisor 8*q, round half away from zero (see fdct_quantize above)
                const int v = d[k];
                const int sgn = v >> 31;
                const unsigned a16 = (((unsigned)((v ^ sgn) - sgn)) << 4) + h8[k];
                const int m = (int)__umulhi(a16, m8[k]);
                r[jj][k] = (m ^ sgn) - sgn;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            *reinterpret_cast<__attribute__((address_space(3))) unsigned*>(slot + k * 16 + (p ? 8 : 0) + half * 4) = pack16(r[0][k], r[1][k]);
    }
}

// byte offsets (inside a natural-order int16 block) of the eight coefficients that make up 16-byte piece `piece` of the
// zigzag-ordered block
__device__ const unsigned short kZigzagPieceOffsets[8][8] = {
    {0, 2, 16, 32, 18, 4, 6, 20},       {34, 48, 64, 50, 36, 22, 8, 10},    {24, 38, 52, 66, 80, 96, 82, 68},
    {54, 40, 26, 12, 14, 28, 42, 56},   {70, 84, 98, 112, 114, 100, 86, 72}, {58, 44, 30, 46, 60, 74, 88, 102},
    {116, 118, 104, 90, 76, 62, 78, 92}, {106, 120, 122, 108, 94, 110, 124, 126}};

// gathers piece (lane & 7) of block slot `slot` in zigzag order; off = the piece's eight byte offsets, two per register
__device__ __forceinline__ u32x4 gather_zigzag_piece(const lds_char* slot, const uint4& off)
{
    using lds_u16 = __attribute__((address_space(3))) unsigned short;
    const unsigned o[4] = {off.x, off.y, off.z, off.w};
    unsigned h[8];
#pragma unroll
    for (int t = 0; t < 8; t++) h[t] = *reinterpret_cast<const lds_u16*>(slot + ((t & 1) ? (o[t >> 1] >> 16) : (o[t >> 1] & 0xFFFF)));
    return u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
}

#ifndef HJ_PAIR_WAVES
#define HJ_PAIR_WAVES 4
#endif
constexpr int kPairSlotStride = 144;  // 128 B block + 16 B pad

// PLANAR: three separate input planes (P_RGB / P_BGR: CHW tensors) instead of interleaved pixels -- eight bytes per plane and row
// per lane, consecutive blocks consecutive addresses; the arithmetic does not know the difference
template <int HS, int VS, bool PLANAR>
// (4:4:4: the full-resolution chroma tile limits it to three waves per SIMD)
__global__ __launch_bounds__(kThreads, (HS == 1 ? 3 : HJ_PAIR_WAVES)) void forward_pair_kernel(const EncodeImage* __restrict__ images, const EncodeUnit* __restrict__ units)
{
    constexpr int kChromaW = kTileBX * 8 / HS, kChromaH = kTileBY * 8 / VS;  // downsampled chroma tile
    __shared__ __attribute__((aligned(16))) char lds_slots[4 * 32 * kPairSlotStride];             // 18,432 B
    __shared__ __attribute__((aligned(16))) unsigned lds_quant[4][64];                             // luma magic, half16, chroma magic, half16
    __shared__ __attribute__((aligned(16))) unsigned char lds_chroma[2][kChromaH][kChromaW + 16];  // 4:2:0: 9,216 B ... 4:4:4: 34,816 B

    const EncodeUnit u = units[blockIdx.x];
    const EncodeImage& im = images[u.image];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool p = lane & 1;
    const int blk = lane >> 1;
    const int W = im.width, H = im.height;
    {
        const int t = tid >> 6, k = tid & 63;
        lds_quant[t][k] = (t & 1) ? im.qnat[t >> 1].half16[k] : im.qnat[t >> 1].magic[k];
    }
    const uint4 zoff = *reinterpret_cast<const uint4*>(&kZigzagPieceOffsets[lane & 7][0]);
    // colour weights: for BGR input the first and third byte swap roles -- wave-uniform scalars, no per-pixel work
    const bool bgr = im.in_format == (PLANAR ? kInPlanarBGR : kInInterleavedBGR);
    const int y0w = bgr ? 7471 : 19595, y2w = bgr ? 19595 : 7471;
    const int cb0w = bgr ? 32768 : -11059, cb2w = bgr ? -11059 : 32768;
    const int cr0w = bgr ? -5329 : 32768, cr2w = bgr ? 32768 : -5329;
    const int kyc = 32768 - (128 << 16);  // + ONE_HALF, level shift folded in
    const int kcc = (128 << 16) + 32767;  // CBCR_OFFSET + ONE_HALF - 1
    __syncthreads();  // quant tables

    lds_char* wave_slots = (lds_char*)lds_slots + wave * 32 * kPairSlotStride;
    lds_char* slot = wave_slots + blk * kPairSlotStride;
    const auto* qtab = (const __attribute__((address_space(3))) unsigned*)&lds_quant[0][0];
    const unsigned char* in0 = im.in[0];
    const unsigned pitch = im.in_pitch[0];
    const unsigned char* in1 = PLANAR ? im.in[1] : nullptr;
    const unsigned char* in2 = PLANAR ? im.in[2] : nullptr;
    const unsigned pitch1 = PLANAR ? im.in_pitch[1] : 0u, pitch2 = PLANAR ? im.in_pitch[2] : 0u;

    // ---- phase A: wave w takes block rows 2w and 2w+1 of the tile, 32 blocks each
#pragma unroll 1
    for (int it = 0; it < 2; it++) {
        const int lby = wave * 2 + it;
        const int bx = u.tile_bx * kTileBX + blk, by = u.tile_by * kTileBY + lby;
        const int x0 = bx * 8, y0 = by * 8;
        const bool interior = x0 + 8 <= W;
        // the lane's four pixel rows (24 bytes each), loads first
        unsigned w[4][6];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = p ? 7 - i : i;
            const int y = min(y0 + r, H - 1);  // rows past the image replicate the last row (jcprepct.c expand_bottom_edge)
            const unsigned char* rowp = in0 + (size_t)y * pitch;
            typedef u32x2 __attribute__((aligned(1))) u32x2_unaligned;  // any byte alignment: gfx950 fetches an unaligned 8-byte piece in one go
            if constexpr (PLANAR) {
                // w[i][2k], w[i][2k+1] = the row's eight bytes of plane k
                const unsigned char* rows3[3] = {rowp, in1 + (size_t)y * pitch1, in2 + (size_t)y * pitch2};
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    if (interior) {
                        const u32x2 a = *(const __attribute__((address_space(1))) u32x2_unaligned*)(rows3[k] + x0);
                        w[i][2 * k] = a.x;
                        w[i][2 * k + 1] = a.y;
                    } else {
                        w[i][2 * k] = w[i][2 * k + 1] = 0;
#pragma unroll
                        for (int c = 0; c < 8; c++)  // columns past the image replicate the last pixel (expand_right_edge)
                            w[i][2 * k + (c >> 2)] |= (unsigned)rows3[k][min(x0 + c, W - 1)] << (8 * (c & 3));
                    }
                }
            } else if (interior) {
                // (explicitly global: through a generic pointer these become FLAT loads, which also count against the LDS counter)
                const auto* v = (const __attribute__((address_space(1))) u32x2_unaligned*)(rowp + (size_t)x0 * 3);
                // plain loads: the three 8-byte pieces of a lane's 24 bytes are three instructions over the same cache lines (a wave's
                // 768-byte row segment, a third of it per instruction); as non-temporal loads each of them fetched the lines again --
                // 1.056 ms against 0.867 (tools/ab_enc_flag.sh "-DHJ_ENC_NT_LOADS")
#ifdef HJ_ENC_NT_LOADS
                const u32x2 a = __builtin_nontemporal_load(v), b = __builtin_nontemporal_load(v + 1), c = __builtin_nontemporal_load(v + 2);
#else
                const u32x2 a = v[0], b = v[1], c = v[2];
#endif
                w[i][0] = a.x; w[i][1] = a.y; w[i][2] = b.x; w[i][3] = b.y; w[i][4] = c.x; w[i][5] = c.y;
            } else {
#pragma unroll
                for (int k = 0; k < 6; k++) w[i][k] = 0;
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const int x = min(x0 + c, W - 1);  // columns past the image replicate the last pixel (expand_right_edge)
#pragma unroll
                    for (int t = 0; t < 3; t++) w[i][(3 * c + t) >> 2] |= (unsigned)rowp[3 * x + t] << (8 * ((3 * c + t) & 3));
                }
            }
        }
        int cbs[4], crs[4];  // 4:2:0: horizontal pair sums of the previous row
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = p ? 7 - i : i;
            int yacc[8], cb[8], cr[8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int c0 = PLANAR ? (int)((w[i][c >> 2] >> (8 * (c & 3))) & 0xFF) : (int)((w[i][(3 * c) >> 2] >> (8 * ((3 * c) & 3))) & 0xFF);
                const int c1 = PLANAR ? (int)((w[i][2 + (c >> 2)] >> (8 * (c & 3))) & 0xFF) : (int)((w[i][(3 * c + 1) >> 2] >> (8 * ((3 * c + 1) & 3))) & 0xFF);
                const int c2 = PLANAR ? (int)((w[i][4 + (c >> 2)] >> (8 * (c & 3))) & 0xFF) : (int)((w[i][(3 * c + 2) >> 2] >> (8 * ((3 * c + 2) & 3))) & 0xFF);
                // jccolor.c rgb_ycc_convert, SCALEBITS 16
                yacc[c] = __mul24(c0, y0w) + __mul24(c1, 38470) + __mul24(c2, y2w) + kyc;  // bits 31..16: the level-shifted luma sample
                cb[c] = (__mul24(c0, cb0w) + __mul24(c1, -21709) + __mul24(c2, cb2w) + kcc) >> 16;
                cr[c] = (__mul24(c0, cr0w) + __mul24(c1, -27439) + __mul24(c2, cr2w) + kcc) >> 16;
            }
            row_pass_store_pk16(hi16_pair(yacc[0], yacc[1]), hi16_pair(yacc[2], yacc[3]), hi16_pair(yacc[7], yacc[6]), hi16_pair(yacc[5], yacc[4]), slot, r);
            // chroma into the (downsampled) LDS tile: jcsample.c h2v2_downsample (bias 1,2,1,2), h2v1_downsample (bias 0,1,0,1),
            // fullsize_downsample
            using lds_u32 = __attribute__((address_space(3))) unsigned;
            using lds_u32x2 = __attribute__((address_space(3))) u32x2;
            if constexpr (HS == 2 && VS == 2) {
                if (i & 1) {
                    unsigned o0 = 0, o1 = 0;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        o0 |= (unsigned)((cbs[c] + cb[2 * c] + cb[2 * c + 1] + 1 + (c & 1)) >> 2) << (8 * c);
                        o1 |= (unsigned)((crs[c] + cr[2 * c] + cr[2 * c + 1] + 1 + (c & 1)) >> 2) << (8 * c);
                    }
                    const int crow = lby * 4 + (p ? 3 - (i >> 1) : (i >> 1));
                    *(lds_u32*)&lds_chroma[0][crow][blk * 4] = o0;
                    *(lds_u32*)&lds_chroma[1][crow][blk * 4] = o1;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        cbs[c] = cb[2 * c] + cb[2 * c + 1];
                        crs[c] = cr[2 * c] + cr[2 * c + 1];
                    }
                }
            } else if constexpr (HS == 2 && VS == 1) {
                unsigned o0 = 0, o1 = 0;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    o0 |= (unsigned)((cb[2 * c] + cb[2 * c + 1] + (c & 1)) >> 1) << (8 * c);
                    o1 |= (unsigned)((cr[2 * c] + cr[2 * c + 1] + (c & 1)) >> 1) << (8 * c);
                }
                *(lds_u32*)&lds_chroma[0][lby * 8 + r][blk * 4] = o0;
                *(lds_u32*)&lds_chroma[1][lby * 8 + r][blk * 4] = o1;
            } else {
                *(lds_u32x2*)&lds_chroma[0][lby * 8 + r][blk * 8] =
                    u32x2{(unsigned)cb[0] | (cb[1] << 8) | (cb[2] << 16) | ((unsigned)cb[3] << 24), (unsigned)cb[4] | (cb[5] << 8) | (cb[6] << 16) | ((unsigned)cb[7] << 24)};
                *(lds_u32x2*)&lds_chroma[1][lby * 8 + r][blk * 8] =
                    u32x2{(unsigned)cr[0] | (cr[1] << 8) | (cr[2] << 16) | ((unsigned)cr[3] << 24), (unsigned)cr[4] | (cr[5] << 8) | (cr[6] << 16) | ((unsigned)cr[7] << 24)};
            }
        }
        column_pass_quantize(slot, p, qtab, qtab + 64);
        pair_lds_fence();
        // copy-out: the wave's 32 blocks are 4 KB contiguous in the luma grid
        if (by < (int)im.real_h[0]) {
            int16_t* rowbase = im.coef[0] + ((size_t)by * im.blocks_w[0] + (size_t)u.tile_bx * kTileBX) * 64;
            const int nvalid = min((int)im.real_w[0] - (int)u.tile_bx * kTileBX, kTileBX);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int g = k * 64 + lane, b = g >> 3;
                if (b < nvalid) {
                    const u32x4 v = gather_zigzag_piece(wave_slots + b * kPairSlotStride, zoff);
                    __builtin_nontemporal_store(v, (__attribute__((address_space(1))) u32x4*)rowbase + g);
                }
            }
        }
        pair_lds_fence();  // slots are reused by the next block row
    }
    __syncthreads();  // chroma tile complete

    // ---- phase B: the tile's chroma blocks, two lanes each: (32/HS) x (8/VS) blocks per component
    constexpr int cbw = kTileBX / HS, cbh = kTileBY / VS, per_comp = cbw * cbh, nchroma = 2 * per_comp;
    const int last_row = (H + VS - 1) / VS - 1;  // last real downsampled row (rows below replicate it: jcprepct.c)
#pragma unroll 1
    for (int base = 0; base < nchroma; base += kThreads / 2) {
        const int idx = base + (tid >> 1);  // < nchroma: every count is a multiple of 128
        const int comp = idx / per_comp, rem = idx - comp * per_comp;
        const int cby = rem / cbw, cbx = rem - cby * cbw;
        const int gcy = u.tile_by * cbh + cby;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = p ? 7 - i : i;
            const int lrow = min(gcy * 8 + r, last_row) - (int)u.tile_by * kChromaH;
            const u32x2 v = *(const __attribute__((address_space(3))) u32x2*)&lds_chroma[comp][lrow][cbx * 8];
            // bytes b0..b7 -> (b0,b1), (b2,b3), (b7,b6), (b5,b4) as int16 pairs, level shift as one packed subtraction each
            constexpr unsigned kShift = 0x00800080u;
            const unsigned P0 = pk_sub16(__builtin_amdgcn_perm(0u, v.x, 0x0c010c00u), kShift), P1 = pk_sub16(__builtin_amdgcn_perm(0u, v.x, 0x0c030c02u), kShift);
            const unsigned P2 = pk_sub16(__builtin_amdgcn_perm(0u, v.y, 0x0c020c03u), kShift), P3 = pk_sub16(__builtin_amdgcn_perm(0u, v.y, 0x0c000c01u), kShift);
            row_pass_store_pk16(P0, P1, P2, P3, slot, r);
        }
        column_pass_quantize(slot, p, qtab + 128, qtab + 192);
        pair_lds_fence();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int g = k * 64 + lane, b = g >> 3;  // block b of this wave's 32
            const int bidx = base + wave * 32 + b;
            const int bcomp = bidx / per_comp, brem = bidx - bcomp * per_comp;
            const int bcy = brem / cbw, bcx = brem - bcy * cbw;
            const int ggx = u.tile_bx * cbw + bcx, ggy = u.tile_by * cbh + bcy;
            if (ggx < (int)im.real_w[1 + bcomp] && ggy < (int)im.real_h[1 + bcomp]) {
                const u32x4 v = gather_zigzag_piece(wave_slots + b * kPairSlotStride, zoff);
                auto* dst = (__attribute__((address_space(1))) u32x4*)(im.coef[1 + bcomp] + ((size_t)ggy * im.blocks_w[1 + bcomp] + ggx) * 64) + (lane & 7);
                __builtin_nontemporal_store(v, dst);
            }
        }
        pair_lds_fence();
    }
}

}  // namespace

// Pre-subsampled planar YCbCr input (NVIMGCODEC_SAMPLEFORMAT_P_YUV; the reference hands such images to nvjpegEncodeYUV,
// extensions/nvjpeg/cuda_encoder.cpp:362-368): no colour conversion, no downsampling -- every plane is a component as it goes
// into the stream.  One lane per block of the component's real block grid (unit: pad = component, tile_bx = first block);
// samples past the plane's edge replicate the last column / row, as libjpeg pads (jcprepct.c, jcsample.c expand_right_edge).
__global__ __launch_bounds__(kThreads) void forward_planes_kernel(const EncodeImage* __restrict__ images, const EncodeUnit* __restrict__ units)
{
    const EncodeUnit u = units[blockIdx.x];
    const EncodeImage& im = images[u.image];
    const int c = (int)u.pad;  // 0, 1, 2 (uniform)
    const unsigned char* plane = c == 0 ? im.in[0] : c == 1 ? im.in[1] : im.in[2];
    const unsigned pitch = c == 0 ? im.in_pitch[0] : c == 1 ? im.in_pitch[1] : im.in_pitch[2];
    const int real_w = (int)(c == 0 ? im.real_w[0] : c == 1 ? im.real_w[1] : im.real_w[2]);
    const int real_h = (int)(c == 0 ? im.real_h[0] : c == 1 ? im.real_h[1] : im.real_h[2]);
    const int grid_w = (int)(c == 0 ? im.blocks_w[0] : c == 1 ? im.blocks_w[1] : im.blocks_w[2]);
    int16_t* coef = c == 0 ? im.coef[0] : c == 1 ? im.coef[1] : im.coef[2];
    // the plane's own size: luma = the picture, chroma = ceil(picture / sampling factor)
    const int pw = c == 0 ? (int)im.width : ((int)im.width + (int)im.hs - 1) / (int)im.hs;
    const int ph = c == 0 ? (int)im.height : ((int)im.height + (int)im.vs - 1) / (int)im.vs;
    const int b = (int)u.tile_bx + (int)threadIdx.x;
    if (b >= real_w * real_h) return;
    const int by = b / real_w, bx = b - by * real_w;
    int s[8][8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const unsigned char* row = plane + (size_t)min(by * 8 + r, ph - 1) * pitch;
#pragma unroll
        for (int k = 0; k < 8; k++) s[r][k] = (int)row[min(bx * 8 + k, pw - 1)] - 128;
    }
    u32x4 packed[8];
    if (c == 0)
        fdct_quantize(s, im.quant[0], packed);
    else
        fdct_quantize(s, im.quant[1], packed);
    u32x4* dst = reinterpret_cast<u32x4*>(coef + ((size_t)by * grid_w + bx) * 64);
#pragma unroll
    for (int j = 0; j < 8; j++) dst[j] = packed[j];
}

int launch_forward(const EncodeImage* images, const EncodeUnit* units, int nunits, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(forward_kernel, dim3(nunits), dim3(kThreads), 0, (hipStream_t)stream, images, units);
    return (int)hipGetLastError();
}

int launch_forward_planes(const EncodeImage* images, const EncodeUnit* units, int nunits, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(forward_planes_kernel, dim3(nunits), dim3(kThreads), 0, (hipStream_t)stream, images, units);
    return (int)hipGetLastError();
}

template <bool PLANAR>
static int launch_forward_pair_t(int hs, int vs, const EncodeImage* images, const EncodeUnit* units, int nunits, hipStream_t stream)
{
    if (hs == 2 && vs == 2)
        hipLaunchKernelGGL((forward_pair_kernel<2, 2, PLANAR>), dim3(nunits), dim3(kThreads), 0, stream, images, units);
    else if (hs == 2 && vs == 1)
        hipLaunchKernelGGL((forward_pair_kernel<2, 1, PLANAR>), dim3(nunits), dim3(kThreads), 0, stream, images, units);
    else if (hs == 1 && vs == 1)
        hipLaunchKernelGGL((forward_pair_kernel<1, 1, PLANAR>), dim3(nunits), dim3(kThreads), 0, stream, images, units);
    else
        return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

int launch_forward_pair(int hs, int vs, bool planar, const EncodeImage* images, const EncodeUnit* units, int nunits, void* stream)
{
    if (nunits <= 0) return 0;
    return planar ? launch_forward_pair_t<true>(hs, vs, images, units, nunits, (hipStream_t)stream)
                  : launch_forward_pair_t<false>(hs, vs, images, units, nunits, (hipStream_t)stream);
}

}  // namespace hipjpeg
