// jpeg_syntax.h -- JPEG marker-segment parser for the HIP JPEG extension (host side).
//
// Plays the role nvjpegJpegStreamParse plays in the reference's GPU plugin
// (extensions/nvjpeg/cuda_decoder.cpp:503-504) and mirrors what the framework-side header parser reports
// (src/parsers/jpeg.cpp:202-361): frame geometry, sampling factors, tables, scan list.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/hipjpeg.h"

namespace hipjpeg {

enum class ColorModel : int { Gray = 0, YCbCr = 1, RGB = 2, CMYK = 3, YCCK = 4 };

enum ParseStatus : int {
    kParseOk = 0,
    kParseBadStream = -1,    // not a JPEG / malformed segment
    kParseUnsupported = -2,  // valid JPEG we do not decode (arithmetic, lossless, 12-bit, hierarchical)
    kParseTruncated = -3,
};

struct HuffSpec {
    uint8_t bits[17] = {0};  // bits[l] = number of codes of length l (1..16)
    uint8_t vals[256] = {0};
    bool present = false;
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int blocks_w = 0, blocks_h = 0;  // allocation grid, padded to whole MCUs
    int samp_w = 0, samp_h = 0;      // true component size: ceil(W*h/hmax) x ceil(H*v/vmax)
};

struct ScanHeader {
    int ncomp = 0;
    int comp_index[4] = {0, 0, 0, 0};
    int td[4] = {0, 0, 0, 0}, ta[4] = {0, 0, 0, 0};
    int ss = 0, se = 63, ah = 0, al = 0;
    size_t data_begin = 0;  // offset of the first entropy-coded byte
    size_t data_end = 0;    // offset one past the last entropy-coded byte (next non-RST marker's 0xFF, or stream end)
    // Tables in force when this scan starts (DHT/DQT/DRI may be redefined between scans)
    HuffSpec dc[4], ac[4];
    int restart_interval = 0;
    bool plain_stuffing = true;  // the segment consists of data bytes, FF 00 pairs and RSTn markers in their proper order
                                 // (no fill bytes, no lone FF at the end of the input)
    std::vector<uint32_t> rst_after;  // per RSTn marker: offset of the byte behind it in the destuffed segment
    // per kScanChunkBytes-byte chunk of [data_begin, data_end): how many of its bytes byte-stuffing removal drops (the 00 behind an
    // FF, both bytes of an RSTn marker).  The marker walk sees every FF anyway; the GPU entropy stage takes these counts instead of
    // counting again (gpu_huffman.hip destuff_count_kernel).  Meaningful for plain_stuffing scans only.
    std::vector<uint32_t> chunk_drops;
};
constexpr size_t kScanChunkBytes = 16384;

struct FrameInfo {
    int width = 0, height = 0, precision = 8, ncomp = 0;
    int sof = 0;  // 0xC0 / 0xC1 / 0xC2
    int hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    bool saw_jfif = false, saw_adobe = false;
    int adobe_transform = -1;
    ColorModel color = ColorModel::YCbCr;
    Component comp[4];
    uint16_t qtab[4][64];       // quant table captured per COMPONENT at its first scan, natural order
    bool qtab_16bit[4] = {false, false, false, false};
    std::vector<ScanHeader> scans;
    bool progressive() const { return sof == 0xC2; }
    size_t total_blocks() const
    {
        size_t n = 0;
        for (int c = 0; c < ncomp; c++) n += (size_t)comp[c].blocks_w * comp[c].blocks_h;
        return n;
    }
};

// Parses every marker segment up to EOI (or end of data).  `headers_only` stops at the first SOS.
ParseStatus parse_jpeg(const uint8_t* data, size_t size, FrameInfo* out, bool headers_only = false);

extern const uint8_t kZigzagNatural[64];  // zigzag index -> natural (row-major) position

hipjpegChromaSubsampling_t classify_subsampling(const FrameInfo& f);

}  // namespace hipjpeg
