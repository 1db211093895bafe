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
    kFlagAdobeMarker = 4,      // the file carries an Adobe APP14 segment (selects the reference's CMYK -> RGB formula)
};

// Per-component part of the descriptor.  Kept as one aligned record per component (rather than parallel arrays inside
// DecodeImage) so that a kernel indexing it with a run-time component number computes ONE 16-byte-aligned base and then
// uses constant offsets: hipcc otherwise folds byte-array indices into scalar-load base addresses, and an unaligned
// s_load base reads the wrong dwords on gfx950.
struct alignas(16) DecodeComponent {
    const int16_t* coef;  // coefficient blocks
    uint8_t* plane;       // intermediate plane (may be null)
    uint32_t plane_pitch;
    uint16_t blocks_w, blocks_h;  // MCU-padded block grid
    uint16_t samp_w, samp_h;      // true component size in samples
    uint16_t h, v;                // sampling factors
    // Where the DC coefficient of block b is.  Host entropy stage: inside the block (dc_stride == 64, dc unused); GPU entropy
    // stage: dc[b] in a compact plane of DC values (dc_stride == 1), position 0 of the blocks holds zero.
    const int16_t* dc;
    uint32_t dc_stride, pad0;
    // Zero-run-compressed staging (host entropy stage, entropy_decode.h): non-null = `coef` is the start of the picture's SPARSE stream and
    // block_off[b] the byte offset of block b's record in it (0 = all-zero block); null = dense blocks at coef + 64 b.
    const uint32_t* block_off;
    uint64_t pad2;
    // Quantizers as the kernels consume them, int16 pairs in the layout of a block's 16-byte column chunk, so that ONE
    // v_pk_mul_lo_u16 dequantizes two coefficients (the low 16 bits of the product, like the SIMD routine's pmullw):
    // qpk[p][j*4 + i] = q(row 2i, column 4p+j) in the low half, q(row 2i+1, column 4p+j) in the high half.
    uint32_t qpk[2][16];
};

struct alignas(16) DecodeImage {
    DecodeComponent comp[4];
    uint8_t* out[4];  // output planes; interleaved formats use out[0] only
    uint32_t out_pitch[4];
    uint32_t width, height;
    uint32_t ncomp, hmax, vmax, color_model;  // color_model: hipjpeg::ColorModel
    uint32_t out_format, flags;
    uint32_t huff_index, pad1[3];  // GPU entropy stage: index of the image's HuffImage (the FUSED kernels decode from it)
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

// Geometry pass (region of interest + EXIF orientation): copies from the intermediate full-frame picture the pixel kernels
// wrote into the caller's buffer.  One descriptor per transformed image; work units = (descriptor index, first output row).
struct alignas(16) TransformImage {
    const uint8_t* src[3];  // intermediate planes, addressed in stored-image coordinates
    uint8_t* dst[3];
    uint32_t src_pitch[3];
    uint32_t dst_pitch[3];
    int32_t x0, y0;         // region origin in the stored image
    int32_t rw, rh;         // region size (before orientation)
    int32_t out_w, out_h;   // output size (after orientation)
    int32_t orientation;    // EXIF 1..8
    int32_t nplanes, bpp;   // planes, bytes per pixel of each (3 interleaved / 1)
    int32_t pad;
};
constexpr int kTransformRowsPerUnit = 8;

constexpr int kBlocksPerUnit = 128;  // idct_plane_kernel: 256 lanes, two lanes per block
constexpr int kLumaTileW = 32, kLumaTileH = 4;  // luma_color_kernel tile in blocks (one block row per wave)

}  // namespace hipjpeg
