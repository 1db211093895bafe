// encode_layout.h -- host/device shared structures of the encode device stage (plain C++).
//
// HBM layout: input pixels as the caller gave them (interleaved RGB/BGR, planar RGB/BGR or one gray plane, any pitch);
// output = quantized coefficient blocks, int16[64] per block IN ZIGZAG ORDER, blocks in raster order over the MCU-padded
// grid of each component (blocks outside the real width_in_blocks x height_in_blocks area are left untouched: the host
// entropy coder synthesizes libjpeg's "dummy blocks" itself).
#pragma once
#include <cstdint>

namespace hipjpeg {

enum InFormat : uint32_t {
    kInInterleavedRGB = 0,
    kInInterleavedBGR = 1,
    kInPlanarRGB = 2,
    kInPlanarBGR = 3,
    kInGray = 4,
    kInPlanarYUV = 5,  // Y, Cb, Cr planes as they go into the stream: luma at full size, chroma already downsampled
};

// Quantizer of one table, indexed by ZIGZAG position k: divisor = 8*q[k]; half = divisor/2; magic = floor(2^28/divisor)+1
// so that  floor(n / divisor) == (n * magic) >> 28  for every n < 2^17.
struct alignas(16) EncodeQuant {
    uint32_t magic[64];
    uint32_t half[64];
};

// The same quantizer indexed column-major over the natural block (column * 8 + row), half pre-shifted by 4 (forward_pair_kernel)
struct alignas(16) EncodeQuantNatural {
    uint32_t magic[64];
    uint32_t half16[64];
};

struct alignas(16) EncodeImage {
    const uint8_t* in[4];
    int16_t* coef[4];
    uint32_t in_pitch[4];
    uint32_t blocks_w[4], blocks_h[4];  // allocation grid (MCU padded)
    uint32_t real_w[4], real_h[4];      // width_in_blocks / height_in_blocks: blocks that carry real samples
    uint32_t width, height, ncomp, in_format;
    uint32_t hs, vs, pad0, pad1;        // luma sampling factors (chroma is 1x1)
    EncodeQuant quant[2];               // [0] luma table, [1] chroma table
    EncodeQuantNatural qnat[2];
};

// One workgroup = one tile of 32 x 8 luma blocks.
struct EncodeUnit {
    uint32_t image, tile_bx, tile_by, pad;
};

}  // namespace hipjpeg
