// entropy_decode.h -- host-side Huffman entropy decode (baseline, extended sequential, progressive).
//
// This is the "host stage" of the hybrid decoder: what nvjpegDecodeJpegHost does in the reference's GPU plugin
// (extensions/nvjpeg/cuda_decoder.cpp:527-530).  It turns the entropy-coded segments into dense quantized DCT
// coefficient blocks, written directly in the layout the HIP kernels consume (see device_layout.h):
//   block b of component c = 64 int16 at coef[c] + 64*b, blocks in raster order over the MCU-padded grid,
//   coefficient (row r, col c) of a block stored at index c*8 + r   (column-major: one 16-byte chunk per column,
//   so a GPU lane fetches a whole IDCT column with one 128-bit load).
#pragma once
#include <cstddef>
#include <cstdint>

#include "jpeg_syntax.h"

namespace hipjpeg {

enum EntropyStatus : int {
    kEntropyOk = 0,
    kEntropyCorrupt = -1,    // invalid Huffman code / coefficient index out of range / bad restart marker
    kEntropyTruncated = -2,  // ran out of bits before the scan was complete
    kEntropyMissingTable = -3,
};

// zigzag index -> position inside a device-layout block (transposed natural order)
extern const uint8_t kZigzagDevice[64];

// coef[c] must hold comp[c].blocks_w * comp[c].blocks_h * 64 int16.  Blocks never touched by any scan are zeroed.
// coef_or[c] (optional) receives the bitwise OR of |coefficient| over component c: an upper bound of its magnitudes.
EntropyStatus decode_coefficients(const uint8_t* data, size_t size, const FrameInfo& frame, int16_t* const coef[4],
                                  uint32_t coef_or[4] = nullptr);

}  // namespace hipjpeg
