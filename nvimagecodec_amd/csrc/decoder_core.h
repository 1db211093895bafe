// decoder_core.h -- host-side planning, staging and launch of one decode batch.
//
// A DecodeBatch owns: a pinned staging area [DecodeImage[] | WorkUnit tables | coefficient blocks], its device
// mirror (same offsets, one hipMemcpyAsync), and a device-only arena for intermediate chroma planes.  It is the
// MI355X counterpart of the reference's per-thread {pinned buffer, device buffer, stream} resources
// (extensions/nvjpeg/cuda_decoder.h:54-75), but sized for a whole batch so the device stage is one launch per kernel.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/hipjpeg.h"
#include "device_layout.h"
#include "decode_kernels.h"
#include "gpu_huffman.h"
#include "progressive_gpu.h"
#include "jpeg_syntax.h"
#include "thread_pool.h"

namespace hipjpeg {

// Custom allocation hooks (the plugin forwards nvimgcodecDeviceAllocator_t / nvimgcodecPinnedAllocator_t here).
struct MemoryHooks {
    int (*device_malloc)(void* ctx, void** ptr, size_t size, void* stream) = nullptr;
    int (*device_free)(void* ctx, void* ptr, size_t size, void* stream) = nullptr;
    void* device_ctx = nullptr;
    int (*pinned_malloc)(void* ctx, void** ptr, size_t size, void* stream) = nullptr;
    int (*pinned_free)(void* ctx, void* ptr, size_t size, void* stream) = nullptr;
    void* pinned_ctx = nullptr;
};

class Buffer {
public:
    enum Kind { kDevice, kPinned };
    Buffer(Kind kind, const MemoryHooks* hooks) : kind_(kind), hooks_(hooks) {}
    ~Buffer() { release(); }
    Buffer(const Buffer&) = delete;
    Buffer& operator=(const Buffer&) = delete;
    // grow-only; contents are NOT preserved
    hipjpegStatus_t reserve(size_t bytes);
    void release();
    uint8_t* data() const { return ptr_; }
    size_t capacity() const { return cap_; }
    bool custom() const { return custom_; }  // allocated through the caller's hooks

private:
    Kind kind_;
    const MemoryHooks* hooks_;
    uint8_t* ptr_ = nullptr;
    size_t cap_ = 0;
    bool custom_ = false;
};

enum KernelVariant { kVarGray = 0, kVar11 = 1, kVar21 = 2, kVar22 = 3, kVar12 = 4, kNumLumaVariants = 5 };

struct PlannedImage {
    FrameInfo frame;
    hipjpegStatus_t status = HIPJPEG_STATUS_SUCCESS;
    const uint8_t* data = nullptr;
    size_t size = 0;
    size_t coef_offset[4] = {0, 0, 0, 0};  // byte offset of component c inside the staging area
    int variant = -1;                      // KernelVariant, or -1 = generic colour path, -2 = planes-to-output only, -3 = CMYK / YCCK
    uint32_t coef_or[4] = {0, 0, 0, 0};    // OR of |coefficient| per component (from the entropy stage)
    uint32_t ac_bound[4] = {32768, 32768, 32768, 32768};  // upper bound of |AC coefficient| per component (packed IDCT pass 1 decision); default: any int16, -32768 included
    // GPU entropy decoding (flag HIPJPEG_FLAG_GPU_HUFFMAN and an eligible stream): the host only destuffs the scan
    bool gpu_entropy = false;
    int huff_index = -1;          // index into the HuffImage array
    size_t stream_offset = 0;     // staging offsets of the destuffed stream and the 8 expanded tables
    size_t tables_offset = 0;
    bool has_transform = false;  // region of interest and/or EXIF orientation (geometry pass)
    hipjpegTransform_t transform = {0, 0, 0, 0, 1};
    int xform_index = -1;
    size_t pool_words = 0;      // lookup-table entries of the scan (GPU entropy path)
    size_t raw_offset = 0;      // staged copy of the scan's entropy-coded bytes
    uint32_t first_chunk = 0;   // first destuff chunk (batch-wide numbering)
    size_t boundary_offset = 0;   // restart boundaries + per-subsequence boundary index (staging area)
    uint32_t num_boundaries = 0;
    size_t block_pos_offset = 0;  // bytes into the block-position scratch
    size_t dc_diff_offset = 0;
    size_t dc_plane_offset[4] = {0, 0, 0, 0};  // bytes into the same scratch: compact DC planes per component  // bytes into the DC-difference scratch
    uint32_t stream_bytes = 0;
    // progressive scans on the GPU entropy stage (progressive_gpu_core.h): gpu_entropy is set as well (device-only coefficient
    // arena, compact DC planes); every scan has a HuffImage of its own for the destuff kernels
    bool gpu_prog = false;
    bool sparse = false;        // host entropy stage wrote the picture's zero-run-compressed stream (entropy_decode.h) instead of dense blocks
    size_t host_coef_offset = 0, host_coef_bytes = 0;  // where this picture's host-decoded coefficients (sparse stream or dense blocks) landed
    bool input_pinned = false;  // the caller's bitstream memory is page-locked: the scan's bytes are fetched from there, no staging copy
    const uint8_t* input_device_view = nullptr;  // ... `data` as the device addresses it (hipPointerGetAttributes)
    int prog_index = -1;               // index into the ProgImage array
    uint32_t prog_huff_first = 0;      // HuffImage index of scan 0, relative to the first progressive one
    size_t prog_raw_offset[kProgMaxScans] = {0};     // staged copy of each scan's entropy-coded bytes (staging area)
    size_t prog_stream_offset[kProgMaxScans] = {0};  // destuffed streams (device-only scratch)
    uint32_t prog_first_chunk[kProgMaxScans] = {0};
    size_t prog_pos_offset[kProgMaxScans] = {0};     // block positions of the AC scans (device-only scratch)
};

class DecodeBatch {
public:
    DecodeBatch(int device_id, const MemoryHooks* hooks);
    ~DecodeBatch();

    // Phase 0: parse headers, choose kernels, lay out staging memory.  Per-image problems land in statuses[i].
    // `formats` (optional) gives one output format per image; otherwise `format` applies to all.
    hipjpegStatus_t plan(const uint8_t* const* data, const size_t* lengths, int n, const hipjpegOutput_t* outputs,
                         hipjpegOutputFormat_t format, unsigned flags, hipjpegStatus_t* statuses,
                         const hipjpegOutputFormat_t* formats = nullptr, ForkJoinPool* pool = nullptr,
                         const hipjpegTransform_t* transforms = nullptr);
    // GPU entropy stage only for images of MORE than this many pixels (width x height); smaller ones keep the host Huffman decoder.
    // The reference's nvJPEG plugin has the same switch between its HYBRID and GPU_HYBRID backends (`hybrid_huffman_threshold`,
    // extensions/nvjpeg/cuda_decoder.cpp:188-209, 512-521; default 1000 x 1000 there).  Default here 0: every eligible stream goes to
    // the GPU -- measured faster from 224 x 224 upwards.
    void set_gpu_entropy_threshold(uint64_t pixels) { gpu_entropy_min_pixels_ = pixels; }
    // Between plan() and entropy_stage(): give up image i (e.g. the caller's image descriptor is too small for it).
    void reject(int i, hipjpegStatus_t st)
    {
        if (i >= 0 && i < (int)images_.size() && images_[i].status == HIPJPEG_STATUS_SUCCESS) images_[i].status = st;
    }
    // Size of what image i writes into the caller's buffer: width x height after region of interest and orientation.
    void output_size(int i, int* w, int* h) const;
    // Phase 1: entropy-decode image i into the pinned staging area.  Thread-safe for distinct i.
    void entropy_stage(int i);
    // Phase 1b: after every entropy_stage returned: final per-image flags, drop failed images from the unit tables.
    void finalize(hipjpegStatus_t* statuses);
    // Phase 2: one async H2D copy of descriptors + coefficients.
    // kernels_on_other_stream: the copy is issued on a stream of its own; launch() then makes the kernels' stream wait for
    // it on the device (copy of batch n+1 overlaps the kernels of batch n)
    hipjpegStatus_t transfer(void* stream, bool kernels_on_other_stream = false);
    // Phase 3: kernel launches.  which = -1: all; 0 idct_plane, 1 luma_color (every variant), 2 generic_color,
    // 3 GPU entropy stage (blocks until its result status has been read back).
    // entropy_stream (optional): a second stream for the GPU entropy stage, see launch() in decoder_core.cpp
    hipjpegStatus_t launch(void* stream, int which = -1, void* entropy_stream = nullptr);
    // After launch(): waits for `stream` and settles the GPU entropy stage's verdicts (see decoder_core.cpp); image(i).status
    // is final afterwards.  A no-op for batches without GPU-decoded streams.
    hipjpegStatus_t resolve(void* stream);
    // Blocks until the kernels of the last launch() have finished (the event recorded behind them).
    hipjpegStatus_t wait_done();
    void* last_stream() const { return last_stream_; }
    int gpu_entropy_images() const { return (int)(huff_to_image_.size() + prog_to_image_.size()); }
    int last_sync_launches() const { return last_sync_launches_; }
    int host_fallback_images() const { return host_fallback_images_; }  // GPU-entropy images the host decoder took over in resolve()
    bool has_progressive() const { return !prog_to_image_.empty(); }
    uint64_t stream_bytes() const { return stream_bytes_total_; }
    int zero_copy_images() const { return zero_copy_images_; }
    uint64_t h2d_bytes() const { return h2d_used_; }  // bytes the current batch's transfer() copies to the device
    int sparse_images() const
    {
        int n = 0;
        for (const PlannedImage& im : images_) n += im.sparse ? 1 : 0;
        return n;
    }  // images of the current batch whose bitstream went to the device from the caller's own (pinned) memory
    void flavour_units(int32_t* plane_units, int32_t luma_units[kNumLumaLayouts]) const
    {
        *plane_units = (int32_t)(plane_units_.size() + fused_plane_units_.size());
        for (int e = 0; e < kNumLumaLayouts; e++) {
            luma_units[e] = 0;
            for (const auto& v : luma_units_[e]) luma_units[e] += (int32_t)v.size();
            for (const auto& v : fused_luma_units_[e]) luma_units[e] += (int32_t)v.size();
        }
    }
    int fused_units() const  // work units of the FUSED kernel builds (blocks decoded inside the pixel kernels) in the current batch
    {
        size_t n = fused_plane_units_.size();
        for (int e = 0; e < kNumLumaLayouts; e++)
            for (const auto& v : fused_luma_units_[e]) n += v.size();
        return (int)n;
    }

    int size() const { return (int)images_.size(); }
    const PlannedImage& image(int i) const { return images_[i]; }
    void stats(int32_t num_units[3], uint64_t* coef_bytes, uint64_t* output_bytes) const;

private:
    hipjpegStatus_t plan_once(const uint8_t* const* data, const size_t* lengths, int n, const hipjpegOutput_t* outputs,
                              hipjpegOutputFormat_t format, unsigned flags, hipjpegStatus_t* statuses, const hipjpegOutputFormat_t* formats,
                              ForkJoinPool* pool, const hipjpegTransform_t* transforms, const std::vector<char>& give_up);
    int device_id_;
    Buffer pinned_, device_, planes_;
    std::vector<PlannedImage> images_;
    std::vector<DecodeImage> desc_;  // host copy (device pointers inside)
    std::vector<WorkUnit> plane_units_, luma_units_[kNumLumaLayouts][kNumLumaVariants], generic_units_, cmyk_units_;  // luma: [layout of K2][sampling]
    size_t desc_offset_ = 0, units_offset_ = 0, coef_offset_ = 0, staging_bytes_ = 0, plane_bytes_ = 0;
    size_t unit_off_plane_ = 0, unit_off_luma_[kNumLumaLayouts][kNumLumaVariants] = {{0}}, unit_off_generic_ = 0, unit_off_cmyk_ = 0;
    // Images of the GPU entropy stage (baseline): their K1 / K2 units go to the FUSED kernel builds, which Huffman-decode the blocks
    // themselves (decode_kernels.hip) -- same unit geometry.  host_taken_: such images the host entropy decoder took over in resolve()
    // (damaged / periodic streams): their coefficients then lie in HBM and the plain builds run for them (launch_taken_pixels).
    std::vector<WorkUnit> fused_plane_units_, fused_luma_units_[kNumLumaLayouts][kNumLumaVariants];
    size_t unit_off_fused_plane_ = 0, unit_off_fused_luma_[kNumLumaLayouts][kNumLumaVariants] = {{0}};
    bool fused_ = false;
    uint64_t gpu_entropy_min_pixels_ = 0;
    size_t raw_region_begin_ = 0, raw_region_end_ = 0;  // the staged bitstreams inside the H2D part of the staging area
    // Host-decoded coefficients are handed out of their region [coef_offset_, h2d_bytes_) first come first served while the pool threads
    // decode (a sparse stream's size is known only then): h2d_used_ = what transfer() actually has to copy.
    bool sparse_mode_ = false;
    std::atomic<size_t> host_coef_used_{0};
    size_t h2d_used_ = 0;
    int zero_copy_images_ = 0;
    std::vector<int> host_taken_;
    void* taken_units_dev_ = nullptr;
    size_t taken_units_cap_ = 0;
    int launch_taken_pixels(void* stream, int which);
    uint64_t coef_bytes_ = 0, output_bytes_ = 0;
    bool finalized_ = false;
    // ---- GPU entropy stage
    struct EntropyLaunch {
        HuffImage* dimg;
        const HuffUnit *dunits, *dwunits, *ddc;
        const uint32_t* dlist;
        unsigned long long *states, *incoming;
        uint32_t* first_block;
        int32_t* group_sums;
        unsigned int *changed, *host_changed;
        HuffImage* himg;
        unsigned pool_bytes;
        int nunits;
    };
    EntropyLaunch entropy_launch_args();
    bool entropy_write_passes(const EntropyLaunch& L, void* stream);
    void stage_chunk_drops(const ScanHeader& sc, uint32_t first_chunk);
    hipjpegStatus_t enqueue_gpu_entropy(void* stream);
    int launch_pixel_kernels(void* stream, int which);
    bool entropy_pending_ = false, pixels_launched_ = false, copy_pending_ = false;
    void* last_stream_ = nullptr;  // stream of the last launch()
    void* copied_event_ = nullptr;
    void* entropy_event_ = nullptr;  // hipEvent_t: entropy stage finished on its own stream  // hipEvent_t: H2D copy issued on a stream other than the kernels' 
    Buffer work_;  // device only: subsequence states, first-block indices, change counter
    std::vector<HuffImage> huff_images_;
    std::vector<HuffUnit> huff_units_, huff_dc_units_;
    std::vector<uint32_t> huff_list_;
    std::vector<int> huff_to_image_;
    size_t huff_desc_offset_ = 0, huff_units_offset_ = 0, huff_dc_units_offset_ = 0, huff_list_offset_ = 0, h2d_bytes_ = 0;
    size_t gpu_coef_begin_ = 0, gpu_coef_bytes_ = 0, total_subseq_ = 0, max_huff_units_ = 0, max_pool_words_ = 0;
    size_t work_first_block_ = 0, work_changed_ = 0, work_incoming_ = 0, work_tail_ = 0, work_dc_diff_ = 0, work_block_pos_ = 0, work_drops_ = 0, work_streams_ = 0, work_group_sums_ = 0;
    size_t huff_chunk_units_offset_ = 0, huff_wunits_offset_ = 0, max_huff_wunits_ = 0;
    size_t huff_drops_offset_ = 0;  // per destuff chunk: bytes to drop, counted by the parser's marker walk (ScanHeader::chunk_drops)
    std::atomic<bool> host_drops_missing_{false};  // set by a host task that found a scan without counts: the device counts instead
    std::vector<TransformImage> xform_desc_;
    std::vector<WorkUnit> xform_units_;
    size_t xform_desc_offset_ = 0, xform_units_offset_ = 0;
    std::vector<HuffUnit> huff_wunits_;  // block kernel: kHuffMcusPerWg MCUs per workgroup
    std::vector<HuffUnit> huff_chunk_units_;  // offsets into work_
    // progressive images of the batch
    std::vector<ProgImage> prog_images_;
    std::vector<int> prog_to_image_;
    std::vector<HuffUnit> prog_units_;  // replay kernel: {ProgImage index, component << 28 | first block}
    size_t prog_desc_offset_ = 0, prog_units_offset_ = 0, max_prog_units_ = 0, prog_scan_total_ = 0, work_prog_pos_ = 0;
    unsigned prog_slot_words_ = 0;
    hipjpegStatus_t enqueue_progressive(void* stream);
    uint64_t stream_bytes_total_ = 0;
    int last_sync_launches_ = 0, host_fallback_images_ = 0;
    ForkJoinPool* pool_ = nullptr;  // the pool plan() was given (resolve() decodes handed-over images on it)
    unsigned sync_rounds_total_ = 0, sync_rounds_max_ = 0;  // correction rounds of the first sync launch (sum over workgroups, maximum)
    bool entropy_done_ = false;
    void* done_event_ = nullptr;  // hipEvent_t recorded after the last launch that reads this batch's buffers
    bool in_flight_ = false;
};

// status helpers
hipjpegStatus_t status_from_parse(ParseStatus s);

}  // namespace hipjpeg
