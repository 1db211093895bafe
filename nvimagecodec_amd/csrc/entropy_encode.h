// entropy_encode.h -- host-side baseline and progressive Huffman coder + JFIF marker writer for the encode path
// (the part of nvjpegEncodeImage / nvjpegEncodeRetrieveBitstream that stays on the CPU; reference call sites
// extensions/nvjpeg/cuda_encoder.cpp:336-381).  Bitstream conventions follow libjpeg (jcmarker.c / jchuff.c / jccoefct.c)
// so that, with the Annex-K tables, the output is byte-identical to libjpeg-turbo's for the same coefficients.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace hipjpeg {

struct EncodeGeometry {
    int width = 0, height = 0, ncomp = 3;
    int hs = 2, vs = 2;                   // luma sampling factors; chroma components are 1x1
    int blocks_w[3] = {0, 0, 0}, blocks_h[3] = {0, 0, 0};  // MCU-padded grid
    int real_w[3] = {0, 0, 0}, real_h[3] = {0, 0, 0};      // width_in_blocks / height_in_blocks
    int mcus_x = 0, mcus_y = 0;
};

// jcparam.c jpeg_set_quality(force_baseline): Annex-K tables scaled by quality, natural order.
void quality_tables(int quality, uint16_t lum[64], uint16_t chr[64]);
// Fills the grid fields from width/height/ncomp/hs/vs.
void compute_geometry(EncodeGeometry* g);

struct EntropyEncodeOptions {
    int restart_interval = 0;     // in MCUs, 0 = none
    bool optimized_huffman = false;  // two-pass optimal tables (jchuff.c jpeg_gen_optimal_table); default = Annex-K tables
    bool progressive = false;        // SOF2: jcparam.c jpeg_simple_progression's scan script coded as jcphuff.c does, per-scan optimal
                                     // tables (libjpeg forces them in progressive mode)
};

// coef[c]: zigzag-ordered int16[64] blocks over the MCU-padded grid; only the real_w x real_h area is read.
// Appends a complete JFIF file to `out`.
void encode_jfif(const EncodeGeometry& g, const uint16_t qlum[64], const uint16_t qchr[64], const int16_t* const coef[3],
                 const EntropyEncodeOptions& opt, std::vector<uint8_t>* out);

// For the GPU entropy coder (gpu_huffman_encode.hip): the Annex-K tables as (code, length) per symbol, [0] luma [1] chroma,
// and the bytes of SOI .. SOS for those tables (no DRI).
struct StandardCodeTables {
    uint16_t dc_code[2][16];
    uint16_t ac_code[2][256];
    uint8_t dc_size[2][16];
    uint8_t ac_size[2][256];
};
void standard_code_tables(StandardCodeTables* t);
// Optimized tables for the GPU coder: from symbol counts gathered on the device (counts[t][0][category] for DC table t, counts[t][1][run/size
// symbol] for AC table t; t = 0 luma, 1 chroma) to the (code, length) tables and the bytes of SOI .. SOS with the matching DHT segments --
// jchuff.c jpeg_gen_optimal_table, the very routine the host coder's optimized_huffman path uses, so both paths write the same file.
void optimal_code_tables(const uint32_t counts[2][2][256], const EncodeGeometry& g, const uint16_t qlum[64], const uint16_t qchr[64],
                         StandardCodeTables* t, std::vector<uint8_t>* headers);
void write_standard_headers(const EncodeGeometry& g, const uint16_t qlum[64], const uint16_t qchr[64], std::vector<uint8_t>* out);

}  // namespace hipjpeg
