// device_layout.h -- structures shared by the host code and the HIP kernels (plain C++, no HIP types).
//
// HBM layout of the decode path
// -----------------------------
//   coefficient blocks  int16[64] per 8x8 block, 128 B, blocks in raster order over the MCU-padded component grid.
//                       Inside a block the coefficient at (row r, col c) sits at index c*8 + r, i.e. one 16-byte chunk per
//                       IDCT column; the quant table uses the same order.
//   component planes    u8, pitch = blocks_w*8 rounded up to 16 (+16 slack), only for components that the fused kernel
//                       reads as chroma (or for the generic path).
//   output              whatever the caller described: interleaved RGB/BGR (3 B/px) or separate planes, any pitch.
//   descriptors         DecodeImage[n] + WorkUnit[] tables, uploaded with the coefficients in the same H2D copy.
#pragma once
#include <cstdint>

namespace hipjpeg {

enum OutFormat : uint8_t {
    kOutInterleavedRGB = 0,  // I_RGB
    kOutInterleavedBGR = 1,  // I_BGR
    kOutPlanarRGB = 2,       // P_RGB
    kOutPlanarBGR = 3,       // P_BGR
    kOutY = 4,               // P_Y  (luma / gray only)
    kOutPlanarYUV = 5,       // P_YUV / P_UNCHANGED: raw component planes at component resolution
};

enum ImageFlags : uint8_t {
    kFlagFancyUpsampling = 1,  // libjpeg do_fancy_upsampling
    kFlagExactMul32 = 2,       // coefficient range too wide for 24-bit multiplies in IDCT pass 1
};

struct DecodeImage {
    const int16_t* coef[4];
    uint8_t* plane[4];  // intermediate component planes (may be null)
    uint8_t* out[4];    // output planes; interleaved formats use out[0] only
    uint32_t plane_pitch[4];
    uint32_t out_pitch[4];
    uint16_t qt[4][64];  // per component, column-major like the coefficients
    uint16_t width, height;
    uint16_t blocks_w[4], blocks_h[4];
    uint16_t samp_w[4], samp_h[4];
    uint8_t ncomp, hmax, vmax, color_model;  // color_model: hipjpeg::ColorModel
    uint8_t h[4], v[4];
    uint8_t out_format, flags, pad0, pad1;
};

// One workgroup's worth of work: 256 consecutive blocks (raster order) of one component of one image.
struct WorkUnit {
    uint32_t image;       // index into DecodeImage[]
    uint32_t block_base;  // first block handled by this workgroup
    uint32_t comp;        // component index
    uint32_t mode;        // kernel specific
};

enum PlaneUnitMode : uint32_t {
    kToPlane = 0,   // write whole blocks into DecodeImage::plane[comp]
    kToOutput = 1,  // write into DecodeImage::out[mode>>8], cropped to samp_w x samp_h
};

constexpr int kBlocksPerUnit = 256;

}  // namespace hipjpeg
