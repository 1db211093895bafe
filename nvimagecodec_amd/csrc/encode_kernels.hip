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

}  // namespace

int launch_forward(const EncodeImage* images, const EncodeUnit* units, int nunits, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(forward_kernel, dim3(nunits), dim3(kThreads), 0, (hipStream_t)stream, images, units);
    return (int)hipGetLastError();
}

}  // namespace hipjpeg
