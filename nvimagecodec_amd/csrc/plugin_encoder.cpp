// plugin_encoder.cpp -- the nvImageCodec encoder plugin ("hipjpeg_encoder") of this extension.
//
// Fills nvimgcodecEncoderDesc_t (ABI: include/nvimgcodec_abi.h; reference include/nvimgcodec.h:1087-1145) the way the
// reference's nvJPEG CUDA encoder does (extensions/nvjpeg/cuda_encoder.cpp):
//   * canEncode(): acceptance rules of cuda_encoder.cpp:49-140 (baseline and progressive Huffman output, :77-82)
//   * encode(): quality = int(params->quality) (:336), encoding from the code stream's chained nvimgcodecJpegImageInfo_t (:339-346),
//     optimized_huffman from the chained nvimgcodecJpegEncodeParams_t (:348-357),
//     output subsampling from the code stream's image info (:358-361), bitstream delivered through
//     io_stream->reserve/seek/write/flush (:383-388), exactly one imageReady per sample.
// Device work for the whole batch is ONE launch (colour + downsample + FDCT + quantize); the Huffman stage runs per sample
// on the framework's executor threads.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <sstream>
#include <vector>

#include "encoder_core.h"
#include "plugin_common.h"
#include "plugin_objects.h"

namespace hipjpeg_ext {

using hipjpeg::EncodeBatch;
using hipjpeg::MemoryHooks;

namespace {

bool map_input_format(nvimgcodecSampleFormat_t f, int* fmt)
{
    switch (f) {
    case NVIMGCODEC_SAMPLEFORMAT_I_RGB: *fmt = HIPJPEG_OUTPUT_RGBI; return true;
    case NVIMGCODEC_SAMPLEFORMAT_I_BGR: *fmt = HIPJPEG_OUTPUT_BGRI; return true;
    case NVIMGCODEC_SAMPLEFORMAT_P_RGB: *fmt = HIPJPEG_OUTPUT_RGB_PLANAR; return true;
    case NVIMGCODEC_SAMPLEFORMAT_P_BGR: *fmt = HIPJPEG_OUTPUT_BGR_PLANAR; return true;
    case NVIMGCODEC_SAMPLEFORMAT_P_Y: *fmt = HIPJPEG_OUTPUT_Y; return true;
    case NVIMGCODEC_SAMPLEFORMAT_P_YUV: *fmt = HIPJPEG_OUTPUT_YUV_PLANAR; return true;
    default: return false;
    }
}

bool map_css(nvimgcodecChromaSubsampling_t c, int* css)
{
    switch (c) {
    case NVIMGCODEC_SAMPLING_444: *css = HIPJPEG_CSS_444; return true;
    case NVIMGCODEC_SAMPLING_422: *css = HIPJPEG_CSS_422; return true;
    case NVIMGCODEC_SAMPLING_420: *css = HIPJPEG_CSS_420; return true;
    case NVIMGCODEC_SAMPLING_440: *css = HIPJPEG_CSS_440; return true;
    case NVIMGCODEC_SAMPLING_411: *css = HIPJPEG_CSS_411; return true;
    case NVIMGCODEC_SAMPLING_410: *css = HIPJPEG_CSS_410; return true;
    case NVIMGCODEC_SAMPLING_GRAY: *css = HIPJPEG_CSS_GRAY; return true;
    default: return false;
    }
}

void init_info(nvimgcodecImageInfo_t* info, void* next = nullptr)
{
    memset(info, 0, sizeof *info);
    info->struct_type = NVIMGCODEC_STRUCTURE_TYPE_IMAGE_INFO;
    info->struct_size = sizeof *info;
    info->struct_next = next;
}

}  // namespace

class HipJpegEncoder {
public:
    HipJpegEncoder(const nvimgcodecFrameworkDesc_t* fw, const nvimgcodecExecutionParams_t* ep, const char* options) : fw_(fw), ep_(ep), device_(ep->device_id)
    {
        // "hipjpeg_encoder:gpu_huffman=0" keeps the entropy coder on the executor threads (default: on the GPU)
        for_each_option(options, kEncoderId, [&](const std::string& key, const std::string& value) {
            std::istringstream v(value);
            if (key == "gpu_huffman") v >> gpu_huffman_;
        });
        if (ep->device_allocator && ep->device_allocator->device_malloc && ep->device_allocator->device_free) {
            hooks_.device_malloc = reinterpret_cast<int (*)(void*, void**, size_t, void*)>(ep->device_allocator->device_malloc);
            hooks_.device_free = reinterpret_cast<int (*)(void*, void*, size_t, void*)>(ep->device_allocator->device_free);
            hooks_.device_ctx = ep->device_allocator->device_ctx;
        }
        if (ep->pinned_allocator && ep->pinned_allocator->pinned_malloc && ep->pinned_allocator->pinned_free) {
            hooks_.pinned_malloc = reinterpret_cast<int (*)(void*, void**, size_t, void*)>(ep->pinned_allocator->pinned_malloc);
            hooks_.pinned_free = reinterpret_cast<int (*)(void*, void*, size_t, void*)>(ep->pinned_allocator->pinned_free);
            hooks_.pinned_ctx = ep->pinned_allocator->pinned_ctx;
        }
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || device_ < 0 || device_ >= count) {
            HJ_LOG_ERROR(fw_, kEncoderId, "no usable HIP device " << device_);
            return;
        }
        if (hipSetDevice(device_) != hipSuccess || hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) != hipSuccess) return;
        if (hipEventCreateWithFlags(&event_, hipEventDisableTiming) != hipSuccess) return;
        batch_.reset(new EncodeBatch(device_, &hooks_));
        ok_ = true;
    }
    ~HipJpegEncoder()
    {
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return !busy_; });
        }
        batch_.reset();
        if (event_) (void)hipEventDestroy(event_);
        if (stream_) {
            (void)hipStreamSynchronize(stream_);
            (void)hipStreamDestroy(stream_);
        }
    }
    bool ok() const { return ok_; }

    nvimgcodecStatus_t canEncode(nvimgcodecProcessingStatus_t* status, nvimgcodecImageDesc_t** images, nvimgcodecCodeStreamDesc_t** code_streams,
                                 int n, const nvimgcodecEncodeParams_t* params)
    {
        if (!status || !images || !code_streams || !params) return NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER;
        for (int i = 0; i < n; i++) single_can_encode(&status[i], images[i], code_streams[i]);
        return NVIMGCODEC_STATUS_SUCCESS;
    }

    nvimgcodecStatus_t encode(nvimgcodecImageDesc_t** images, nvimgcodecCodeStreamDesc_t** code_streams, int n, const nvimgcodecEncodeParams_t* params);

private:
    struct Sample {
        nvimgcodecImageDesc_t* image;
        nvimgcodecCodeStreamDesc_t* code_stream;
        nvimgcodecProcessingStatus_t early = NVIMGCODEC_PROCESSING_STATUS_SUCCESS;
    };
    void single_can_encode(nvimgcodecProcessingStatus_t* st, nvimgcodecImageDesc_t* image, nvimgcodecCodeStreamDesc_t* cs);
    static void host_task(int tid, int idx, void* ctx);
    void sample_done();

    const nvimgcodecFrameworkDesc_t* fw_;
    const nvimgcodecExecutionParams_t* ep_;
    MemoryHooks hooks_;
    int device_;
    bool ok_ = false;
    bool gpu_huffman_ = true;       // entropy-code on the GPU what it can take (Annex-K tables, no restart markers)
    std::vector<char> host_coder_;  // per sample of the current batch: still needs the host entropy coder
    hipStream_t stream_ = nullptr;
    hipEvent_t event_ = nullptr;
    std::unique_ptr<EncodeBatch> batch_;
    std::vector<Sample> samples_;
    std::atomic<int> remaining_{0};
    std::mutex m_;
    std::condition_variable cv_;
    bool busy_ = false;
    std::mutex encode_mutex_;
};

void HipJpegEncoder::single_can_encode(nvimgcodecProcessingStatus_t* st, nvimgcodecImageDesc_t* image, nvimgcodecCodeStreamDesc_t* cs)
{
    *st = NVIMGCODEC_PROCESSING_STATUS_SUCCESS;
    nvimgcodecJpegImageInfo_t ji{NVIMGCODEC_STRUCTURE_TYPE_JPEG_IMAGE_INFO, sizeof(nvimgcodecJpegImageInfo_t), nullptr, NVIMGCODEC_JPEG_ENCODING_UNKNOWN};
    nvimgcodecImageInfo_t out_info;
    init_info(&out_info, &ji);
    if (!cs || !image || cs->getImageInfo(cs->instance, &out_info) != NVIMGCODEC_STATUS_SUCCESS) {
        *st = NVIMGCODEC_PROCESSING_STATUS_FAIL;
        return;
    }
    if (strcmp(out_info.codec_name, "jpeg") != 0) {
        *st = NVIMGCODEC_PROCESSING_STATUS_CODEC_UNSUPPORTED;
        return;
    }
    // baseline sequential and progressive Huffman output (cuda_encoder.cpp:77-82); anything else goes down the chain
    if (ji.encoding != NVIMGCODEC_JPEG_ENCODING_UNKNOWN && ji.encoding != NVIMGCODEC_JPEG_ENCODING_BASELINE_DCT &&
        ji.encoding != NVIMGCODEC_JPEG_ENCODING_PROGRESSIVE_DCT_HUFFMAN) {
        *st = NVIMGCODEC_PROCESSING_STATUS_ENCODING_UNSUPPORTED;
        return;
    }
    nvimgcodecImageInfo_t info;
    init_info(&info);
    if (image->getImageInfo(image->instance, &info) != NVIMGCODEC_STATUS_SUCCESS) {
        *st = NVIMGCODEC_PROCESSING_STATUS_FAIL;
        return;
    }
    switch (info.color_spec) {
    case NVIMGCODEC_COLORSPEC_UNCHANGED:
    case NVIMGCODEC_COLORSPEC_SRGB:
    case NVIMGCODEC_COLORSPEC_GRAY:
    case NVIMGCODEC_COLORSPEC_SYCC: break;
    default: *st |= NVIMGCODEC_PROCESSING_STATUS_COLOR_SPEC_UNSUPPORTED;
    }
    int css_in, css_out, fmt = 0;
    if (!map_css(info.chroma_subsampling, &css_in)) *st |= NVIMGCODEC_PROCESSING_STATUS_SAMPLING_UNSUPPORTED;
    if (!map_css(out_info.chroma_subsampling, &css_out)) *st |= NVIMGCODEC_PROCESSING_STATUS_SAMPLING_UNSUPPORTED;
    if (!map_input_format(info.sample_format, &fmt)) {
        *st |= NVIMGCODEC_PROCESSING_STATUS_SAMPLE_FORMAT_UNSUPPORTED;
    } else if (info.sample_format == NVIMGCODEC_SAMPLEFORMAT_P_YUV) {
        // pre-subsampled planar YCbCr goes into the stream as it is (the reference hands it to nvjpegEncodeYUV,
        // cuda_encoder.cpp:362-368): three planes in the output's own sampling -- resampling between samplings is not offered
        if (info.num_planes != 3) *st |= NVIMGCODEC_PROCESSING_STATUS_NUM_PLANES_UNSUPPORTED;
        if (info.color_spec != NVIMGCODEC_COLORSPEC_SYCC) *st |= NVIMGCODEC_PROCESSING_STATUS_COLOR_SPEC_UNSUPPORTED;
        if (info.chroma_subsampling != out_info.chroma_subsampling || out_info.chroma_subsampling == NVIMGCODEC_SAMPLING_GRAY)
            *st |= NVIMGCODEC_PROCESSING_STATUS_SAMPLING_UNSUPPORTED;
    } else if (info.sample_format == NVIMGCODEC_SAMPLEFORMAT_P_Y) {
        // same cross-checks as cuda_encoder.cpp:118-128
        if (info.chroma_subsampling != NVIMGCODEC_SAMPLING_GRAY || out_info.chroma_subsampling != NVIMGCODEC_SAMPLING_GRAY)
            *st |= NVIMGCODEC_PROCESSING_STATUS_SAMPLE_FORMAT_UNSUPPORTED | NVIMGCODEC_PROCESSING_STATUS_SAMPLING_UNSUPPORTED;
        if (info.color_spec != NVIMGCODEC_COLORSPEC_GRAY && info.color_spec != NVIMGCODEC_COLORSPEC_SYCC)
            *st |= NVIMGCODEC_PROCESSING_STATUS_SAMPLE_FORMAT_UNSUPPORTED | NVIMGCODEC_PROCESSING_STATUS_COLOR_SPEC_UNSUPPORTED;
    } else {
        if (out_info.chroma_subsampling == NVIMGCODEC_SAMPLING_GRAY) *st |= NVIMGCODEC_PROCESSING_STATUS_SAMPLING_UNSUPPORTED;
        const bool interleaved = info.sample_format == NVIMGCODEC_SAMPLEFORMAT_I_RGB || info.sample_format == NVIMGCODEC_SAMPLEFORMAT_I_BGR;
        if (interleaved && (info.num_planes != 1 || info.plane_info[0].num_channels != 3)) *st |= NVIMGCODEC_PROCESSING_STATUS_NUM_CHANNELS_UNSUPPORTED;
        if (!interleaved && info.num_planes != 3) *st |= NVIMGCODEC_PROCESSING_STATUS_NUM_PLANES_UNSUPPORTED;
    }
    for (uint32_t p = 0; p < info.num_planes && p < NVIMGCODEC_MAX_NUM_PLANES; ++p)
        if (info.plane_info[p].sample_type != NVIMGCODEC_SAMPLE_DATA_TYPE_UINT8) *st |= NVIMGCODEC_PROCESSING_STATUS_SAMPLE_TYPE_UNSUPPORTED;
}

void HipJpegEncoder::sample_done()
{
    if (remaining_.fetch_sub(1) == 1) {
        {
            std::lock_guard<std::mutex> lk(m_);
            busy_ = false;
        }
        cv_.notify_all();
    }
}

void HipJpegEncoder::host_task(int /*tid*/, int idx, void* ctx)
{
    auto* self = static_cast<HipJpegEncoder*>(ctx);
    Sample& s = self->samples_[idx];
    nvimgcodecProcessingStatus_t ps = s.early;
    // nothing may escape an executor task (the reference's pool swallows exceptions, src/thread_pool.cpp:175-185, and a sample
    // that never reports deadlocks the caller's future): whatever is thrown becomes this sample's FAIL
    try {
    if (ps == NVIMGCODEC_PROCESSING_STATUS_SUCCESS) {
        hipjpeg::PlannedEncode& im = self->batch_->image(idx);
        if (im.status != HIPJPEG_STATUS_SUCCESS) {
            ps = im.status == HIPJPEG_STATUS_UNSUPPORTED ? NVIMGCODEC_PROCESSING_STATUS_SAMPLING_UNSUPPORTED : NVIMGCODEC_PROCESSING_STATUS_FAIL;
        } else {
            if (self->host_coder_[idx]) self->batch_->entropy_stage(idx);  // else: the GPU entropy coder has produced the file already
            // cuda_encoder.cpp:383-388
            nvimgcodecIoStreamDesc_t* io = s.code_stream->io_stream;
            size_t written = 0;
            if (io->reserve(io->instance, im.file_size()) != NVIMGCODEC_STATUS_SUCCESS ||
                io->seek(io->instance, 0, SEEK_SET) != NVIMGCODEC_STATUS_SUCCESS ||
                io->write(io->instance, &written, const_cast<uint8_t*>(im.file()), im.file_size()) != NVIMGCODEC_STATUS_SUCCESS ||
                written != im.file_size() || io->flush(io->instance) != NVIMGCODEC_STATUS_SUCCESS)
                ps = NVIMGCODEC_PROCESSING_STATUS_FAIL;
        }
    }
    } catch (...) {
        ps = NVIMGCODEC_PROCESSING_STATUS_FAIL;
    }
    nvimgcodecImageDesc_t* image = s.image;
    self->sample_done();  // after this the encoder may start the next batch; `s` must not be touched any more
    image->imageReady(image->instance, ps);
}

nvimgcodecStatus_t HipJpegEncoder::encode(nvimgcodecImageDesc_t** images, nvimgcodecCodeStreamDesc_t** code_streams, int n,
                                          const nvimgcodecEncodeParams_t* params)
{
    if (!images || !code_streams || !params) return NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER;
    if (n < 1) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    std::lock_guard<std::mutex> serial(encode_mutex_);
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !busy_; });
        busy_ = true;
    }
    bool gpu_ok = false;
    int reported = 0;  // samples handed to host tasks (they report themselves)
    try {
    samples_.assign(n, Sample());
    std::vector<hipjpegEncodeInput_t> inputs(n);
    std::vector<hipjpegEncodeParams_t> eparams(n);
    memset(inputs.data(), 0, sizeof(hipjpegEncodeInput_t) * n);
    memset(eparams.data(), 0, sizeof(hipjpegEncodeParams_t) * n);
    const auto* jp = find_in_chain<nvimgcodecJpegEncodeParams_t>(params->struct_next, NVIMGCODEC_STRUCTURE_TYPE_JPEG_ENCODE_PARAMS);
    gpu_ok = hipSetDevice(device_) == hipSuccess;
    for (int i = 0; i < n; i++) {
        Sample& s = samples_[i];
        s.image = images[i];
        s.code_stream = code_streams[i];
        nvimgcodecJpegImageInfo_t ji{NVIMGCODEC_STRUCTURE_TYPE_JPEG_IMAGE_INFO, sizeof(nvimgcodecJpegImageInfo_t), nullptr,
                                     NVIMGCODEC_JPEG_ENCODING_UNKNOWN};
        nvimgcodecImageInfo_t info, out_info;
        init_info(&info);
        init_info(&out_info, &ji);
        if (s.image->getImageInfo(s.image->instance, &info) != NVIMGCODEC_STATUS_SUCCESS ||
            s.code_stream->getImageInfo(s.code_stream->instance, &out_info) != NVIMGCODEC_STATUS_SUCCESS) {
            s.early = NVIMGCODEC_PROCESSING_STATUS_FAIL;
            continue;
        }
        if (info.plane_info[0].sample_type != NVIMGCODEC_SAMPLE_DATA_TYPE_UINT8) {
            s.early = NVIMGCODEC_PROCESSING_STATUS_SAMPLE_TYPE_UNSUPPORTED;  // cuda_encoder.cpp:304-308
            continue;
        }
        int fmt = 0, css = 0;
        if (!map_input_format(info.sample_format, &fmt)) {
            s.early = NVIMGCODEC_PROCESSING_STATUS_SAMPLE_FORMAT_UNSUPPORTED;
            continue;
        }
        if (!map_css(out_info.chroma_subsampling, &css)) {
            s.early = NVIMGCODEC_PROCESSING_STATUS_SAMPLING_UNSUPPORTED;
            continue;
        }
        if (info.buffer_kind != NVIMGCODEC_IMAGE_BUFFER_KIND_STRIDED_DEVICE || !info.buffer) {
            s.early = NVIMGCODEC_PROCESSING_STATUS_FAIL;
            continue;
        }
        const uint8_t* p = static_cast<const uint8_t*>(info.buffer);
        for (uint32_t pl = 0; pl < info.num_planes && pl < 3; pl++) {
            inputs[i].plane[pl] = p;
            inputs[i].pitch[pl] = (uint32_t)info.plane_info[pl].row_stride;
            p += info.plane_info[pl].row_stride * info.plane_info[pl].height;
        }
        inputs[i].width = (int32_t)info.plane_info[0].width;
        inputs[i].height = (int32_t)info.plane_info[0].height;
        eparams[i].quality = (int32_t)params->quality;  // cuda_encoder.cpp:336
        eparams[i].subsampling = css;
        eparams[i].input_format = fmt;
        eparams[i].optimized_huffman = jp ? jp->optimized_huffman : 0;
        eparams[i].progressive = ji.encoding == NVIMGCODEC_JPEG_ENCODING_PROGRESSIVE_DCT_HUFFMAN;  // cuda_encoder.cpp:339-346
        // our stream must see the producer's pixels (cuda_encoder.cpp:311-312)
        if (gpu_ok && (hipEventRecord(event_, (hipStream_t)info.cuda_stream) != hipSuccess || hipStreamWaitEvent(stream_, event_, 0) != hipSuccess))
            gpu_ok = false;
    }
    std::vector<hipjpegStatus_t> statuses(n, HIPJPEG_STATUS_SUCCESS);
    if (gpu_ok) gpu_ok = batch_->device_stage(inputs.data(), eparams.data(), n, statuses.data(), stream_) == HIPJPEG_STATUS_SUCCESS;
    host_coder_.assign(n, 1);
    if (gpu_ok && gpu_huffman_) gpu_ok = batch_->gpu_entropy_stage(&host_coder_) == HIPJPEG_STATUS_SUCCESS;  // blocks
    bool any_host = false;
    for (int i = 0; i < n; i++) any_host = any_host || host_coder_[i];
    // the host coder needs the coefficients: blocks like cudaEventSynchronize in cuda_encoder.cpp:374-375
    if (gpu_ok && any_host) gpu_ok = batch_->fetch_coefficients() == HIPJPEG_STATUS_SUCCESS;
    } catch (...) {
        gpu_ok = false;  // an exception while marshalling or in the device stage: the whole batch is reported failed below
    }
    if (!gpu_ok) {
        // batch-level failure: every sample FAIL, an error code back (reference cuda_encoder.cpp error path)
        for (int i = 0; i < n; i++) {
            try {
                images[i]->imageReady(images[i]->instance, NVIMGCODEC_PROCESSING_STATUS_FAIL);
            } catch (...) {
            }
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            busy_ = false;
        }
        cv_.notify_all();
        HJ_LOG_ERROR(fw_, kEncoderId, "device stage of the encode batch failed on device " << device_);
        return NVIMGCODEC_STATUS_EXTENSION_EXECUTION_FAILED;
    }
    remaining_.store(n);
    nvimgcodecExecutorDesc_t* ex = ep_->executor;
    for (int i = 0; i < n; i++, reported++) {
        bool handed = false;
        if (n != 1 && ex) {
            try {
                handed = ex->launch(ex->instance, device_, i, this, &HipJpegEncoder::host_task) == NVIMGCODEC_STATUS_SUCCESS;
            } catch (...) {
                handed = false;
            }
        }
        if (!handed) host_task(0, i, this);  // host_task lets nothing escape
    }
    return NVIMGCODEC_STATUS_SUCCESS;
}

// ------------------------------------------------------------------------------------------------ plugin (factory) object
HipJpegEncoderPlugin::HipJpegEncoderPlugin(const nvimgcodecFrameworkDesc_t* framework)
    : desc_{NVIMGCODEC_STRUCTURE_TYPE_ENCODER_DESC, sizeof(nvimgcodecEncoderDesc_t), nullptr, this, kEncoderId, "jpeg",
            NVIMGCODEC_BACKEND_KIND_HYBRID_CPU_GPU, static_create, static_destroy, static_can_encode, static_encode},
      framework_(framework)
{
}

nvimgcodecStatus_t HipJpegEncoderPlugin::static_create(void* instance, nvimgcodecEncoder_t* encoder, const nvimgcodecExecutionParams_t* exec_params,
                                                       const char* options)
{
    try {
        if (!instance || !encoder || !exec_params) return NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER;
        if (exec_params->device_id == NVIMGCODEC_DEVICE_CPU_ONLY) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
        auto* self = static_cast<HipJpegEncoderPlugin*>(instance);
        std::unique_ptr<HipJpegEncoder> e(new HipJpegEncoder(self->framework_, exec_params, options));
        if (!e->ok()) return NVIMGCODEC_STATUS_EXTENSION_CUDA_CALL_ERROR;
        *encoder = reinterpret_cast<nvimgcodecEncoder_t>(e.release());
        return NVIMGCODEC_STATUS_SUCCESS;
    } catch (...) {
        return NVIMGCODEC_STATUS_EXTENSION_INTERNAL_ERROR;
    }
}

nvimgcodecStatus_t HipJpegEncoderPlugin::static_destroy(nvimgcodecEncoder_t encoder)
{
    try {
        if (!encoder) return NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER;
        delete reinterpret_cast<HipJpegEncoder*>(encoder);
        return NVIMGCODEC_STATUS_SUCCESS;
    } catch (...) {
        return NVIMGCODEC_STATUS_EXTENSION_INTERNAL_ERROR;
    }
}

nvimgcodecStatus_t HipJpegEncoderPlugin::static_can_encode(nvimgcodecEncoder_t encoder, nvimgcodecProcessingStatus_t* status,
                                                           nvimgcodecImageDesc_t** images, nvimgcodecCodeStreamDesc_t** code_streams, int batch_size,
                                                           const nvimgcodecEncodeParams_t* params)
{
    try {
        if (!encoder) return NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER;
        return reinterpret_cast<HipJpegEncoder*>(encoder)->canEncode(status, images, code_streams, batch_size, params);
    } catch (...) {
        return NVIMGCODEC_STATUS_EXTENSION_INTERNAL_ERROR;
    }
}

nvimgcodecStatus_t HipJpegEncoderPlugin::static_encode(nvimgcodecEncoder_t encoder, nvimgcodecImageDesc_t** images,
                                                       nvimgcodecCodeStreamDesc_t** code_streams, int batch_size, const nvimgcodecEncodeParams_t* params)
{
    try {
        if (!encoder) return NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER;
        return reinterpret_cast<HipJpegEncoder*>(encoder)->encode(images, code_streams, batch_size, params);
    } catch (...) {
        return NVIMGCODEC_STATUS_EXTENSION_INTERNAL_ERROR;
    }
}

}  // namespace hipjpeg_ext
