// gpu_huffman_encode.h -- GPU entropy coder of the encode path: baseline Huffman coding with the Annex-K tables, byte
// stuffing and file assembly on the device (gpu_huffman_encode.hip).  Counterpart of the decode side's gpu_huffman.h:
// the quantized coefficients never leave HBM, only finished JPEG files cross PCIe.
//
// Every 8x8 block is independent once the bit length of everything in front of it is known, so the coder is three maps and
// two scans:
//   length   one lane per block (scan order): code length of the block (DC difference + run/size symbols + value bits)
//   scan     per image: exclusive prefix sum of the lengths -> bit offset of every block, total bits
//   -- the host reads the totals, lays out one zeroed bit buffer per image --
//   write    one lane per block: emit the bits at the block's offset -- assembled per workgroup in an LDS window with LDS
//            atomics, copied out as whole words, only the window's two end words OR-ed into memory atomically; the lane
//            of an image's last block pads the final byte with ones (jchuff.c flush_bits)
//   count    per 4 KB chunk of the bit buffer: number of 0xFF bytes (each needs a stuffed 0x00 behind it)
//   layout   per image: prefix sum of those counts -> where each chunk lands; file length; files packed back to back
//   expand   per chunk: stuffed bytes to their final place; the first chunk also lays down SOI..SOS, the last one EOI
// Bit-exact with the host coder (entropy_encode.cpp) and therefore with libjpeg-turbo's output for the same coefficients,
// including libjpeg's "dummy block" rule for the MCU padding (jccoefct.c compress_data).
#pragma once
#include <cstdint>

#include "entropy_encode.h"

namespace hipjpeg {

constexpr int kHencChunk = 4096;  // bytes of unstuffed data per workgroup of the count / expand kernels

struct alignas(16) HencImage {
    const int16_t* coef[3];  // zigzag-ordered blocks over the MCU-padded grid (encode_layout.h)
    uint32_t blocks_w[3], real_w[3], real_h[3];
    uint32_t mcus_x, mcus_y, ncomp, hs, vs, bpm;
    uint32_t total_blocks;   // mcus_x * mcus_y * bpm
    uint32_t first_block;    // index of the image's first block in the batch-wide per-block arrays
    // filled in after the scan (second upload)
    uint8_t* raw;            // zero-initialised bit buffer: ceil(total_bits / 8) rounded up to whole words (+ slack)
    const uint8_t* header;   // SOI .. SOS
    uint32_t raw_bytes, header_bytes;
    uint32_t first_chunk;    // index of the image's first chunk in the batch-wide per-chunk arrays
    uint32_t num_chunks;
    // optimized Huffman tables (nvimgcodecJpegEncodeParams_t::optimized_huffman, reference extensions/nvjpeg/cuda_encoder.cpp:348-357):
    // hist = where the histogram kernel adds this image's symbol counts ([2][2][256] uint32: table, DC / AC, symbol), null for images
    // coded with the Annex-K tables; tables = the image's own code tables once the host has built them from the counts (null: the
    // batch's standard tables)
    uint32_t* hist;
    const StandardCodeTables* tables;
};

struct HencUnit {
    uint32_t image, first;  // first block (length / write kernels) or chunk index (count / expand kernels)
};

// stream = hipStream_t as void*; all launches are asynchronous
// Symbol statistics of the images that want their own tables (jchuff.c's gather-statistics pass, one lane per block): same units as the
// length kernel; images without HencImage::hist are skipped.
int launch_henc_hist(const HencImage* images, const HencUnit* units, int nunits, void* stream);
int launch_henc_length(const HencImage* images, const HencUnit* units, int nunits, const StandardCodeTables* tables, uint16_t* block_bits, void* stream);
int launch_henc_scan(const HencImage* images, int nimages, const uint16_t* block_bits, uint32_t* block_off, uint32_t* total_bits, void* stream);
int launch_henc_write(const HencImage* images, const HencUnit* units, int nunits, const StandardCodeTables* tables, const uint32_t* block_off,
                      const uint16_t* block_bits, void* stream);
int launch_henc_zero(void* p, size_t bytes, void* stream);  // bytes rounded up to 16; p 16-byte aligned
int launch_henc_count(const HencImage* images, const HencUnit* chunk_units, int nchunks, uint32_t* chunk_ff, void* stream);
int launch_henc_layout(const HencImage* images, int nimages, const uint32_t* chunk_ff, uint32_t* chunk_out, uint32_t* final_len,
                       unsigned long long* final_off, void* stream);
int launch_henc_expand(const HencImage* images, const HencUnit* chunk_units, int nchunks, const uint32_t* chunk_out, const uint32_t* final_len,
                       const unsigned long long* final_off, uint8_t* arena, void* stream);

}  // namespace hipjpeg
