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

// ---- zero-run-compressed staging (SURVEY 7 "hard part 3": dense int16 coefficients are 6.27 MB per 1080p picture, 12-20 x the file) ----
// For sequential frames whose blocks are each coded by exactly one scan (one interleaved scan over all components, or a one-component
// frame) the host entropy stage can hand the device a SPARSE picture instead: per block a record of its non-zero coefficients,
//     [n : u8]  [DC : i16 LE]  n x { position : u8 (device-layout index 1..63), value : i16 LE }        (3 + 3 n bytes)
// in scan order behind a table of per-block byte offsets (uint32, relative to the start of the picture's stream; one table per component over
// its MCU-padded raster grid, tables back to back; 0 = block never coded = all zero).  The pixel kernels expand the records into the LDS
// slots they stage dense blocks in (decode_kernels.hip sparse_expand_slots).  ~1.9 MB instead of 6.27 MB for a q90 1080p 4:2:0 photograph.
bool sparse_staging_applies(const FrameInfo& frame);
// worst case of the stream for `frame` in bytes (offset tables + every coefficient of every block non-zero)
size_t sparse_stream_capacity(const FrameInfo& frame);
// Decodes into out[0 .. *size) (capacity sparse_stream_capacity(frame)).  Same verdicts as decode_coefficients.
EntropyStatus decode_coefficients_sparse(const uint8_t* data, size_t size, const FrameInfo& frame, uint8_t* out, size_t* out_size);

}  // namespace hipjpeg
