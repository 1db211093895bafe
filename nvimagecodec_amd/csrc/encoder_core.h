// encoder_core.h -- host-side planning, launch and entropy stage of one encode batch
// (counterpart of decoder_core.h; reference flow extensions/nvjpeg/cuda_encoder.cpp:284-396).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/hipjpeg.h"
#include "decoder_core.h"
#include "encode_layout.h"
#include "entropy_encode.h"
#include "gpu_huffman_encode.h"

namespace hipjpeg {

struct PlannedEncode {
    hipjpegStatus_t status = HIPJPEG_STATUS_SUCCESS;
    EncodeGeometry geom;
    hipjpegEncodeParams_t params{};
    uint16_t qlum[64], qchr[64];
    size_t coef_offset[3] = {0, 0, 0};  // byte offsets inside the coefficient area
    std::vector<uint8_t> bitstream;          // host entropy coder's output
    const uint8_t* gpu_bitstream = nullptr;  // GPU entropy coder's output (inside the batch's pinned arena), or null
    size_t gpu_bitstream_len = 0;
    const uint8_t* file() const { return gpu_bitstream ? gpu_bitstream : bitstream.data(); }
    size_t file_size() const { return gpu_bitstream ? gpu_bitstream_len : bitstream.size(); }
};

class EncodeBatch {
public:
    EncodeBatch(int device_id, const MemoryHooks* hooks);
    ~EncodeBatch();
    // Parse params, lay out device memory, upload descriptors and launch the forward kernel (asynchronous on `stream`).
    hipjpegStatus_t device_stage(const hipjpegEncodeInput_t* inputs, const hipjpegEncodeParams_t* params, int n, hipjpegStatus_t* statuses,
                                 void* stream);
    hipjpegStatus_t relaunch(void* stream);
    // Coefficients D2H (on the stream used by device_stage), wait, then Huffman + markers for image i.
    hipjpegStatus_t fetch_coefficients();
    void entropy_stage(int i);
    // GPU entropy coder (gpu_huffman_encode.h) for every image it can take -- Annex-K tables, no restart markers; blocking.
    // todo[i] = true afterwards for the images that still need the host coder (entropy_stage).
    hipjpegStatus_t gpu_entropy_stage(std::vector<char>* todo);
    uint64_t gpu_entropy_images() const { return gpu_entropy_images_; }
    int size() const { return (int)images_.size(); }
    PlannedEncode& image(int i) { return images_[i]; }
    const int16_t* host_coef(int i, int c) const;
    int num_units() const { return (int)units_.size(); }
    uint64_t pixel_bytes() const { return pixel_bytes_; }
    uint64_t coef_bytes() const { return coef_bytes_; }

private:
    int device_id_;
    Buffer pinned_desc_, device_, pinned_coef_;
    std::vector<PlannedEncode> images_;
    std::vector<EncodeImage> desc_;
    std::vector<EncodeUnit> units_;            // every tile of the batch, grouped by kernel flavour
    // [0] one-lane-per-block kernel, [1..3] forward_pair_kernel 4:2:0 / 4:2:2 / 4:4:4 on interleaved input, [4] planar YCbCr,
    // [5..7] forward_pair_kernel on planar RGB / BGR input
    static constexpr int kUnitLists = 8;
    std::vector<EncodeUnit> unit_lists_[kUnitLists];
    size_t unit_first_[kUnitLists] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t units_offset_ = 0, coef_offset_ = 0, desc_bytes_ = 0, coef_total_ = 0;
    uint64_t pixel_bytes_ = 0, coef_bytes_ = 0;
    void* stream_ = nullptr;
    void* event_ = nullptr;
    bool launched_ = false, fetched_ = false;
    Buffer henc_dev_, henc_dev2_, henc_pinned_, henc_out_;
    uint64_t gpu_entropy_images_ = 0;
};

hipjpegStatus_t subsampling_factors(int subsampling, int* ncomp, int* hs, int* vs);

}  // namespace hipjpeg
