// plugin_decoder.cpp -- the nvImageCodec decoder plugin ("hipjpeg_decoder") of this extension.
//
// Fills the nvimgcodecDecoderDesc_t function table (ABI: include/nvimgcodec_abi.h; reference include/nvimgcodec.h:1150-1209)
// so that ImageGenericDecoder / DecoderWorker dispatch to it exactly as they do to the reference's nvjpeg CUDA decoder
// (extensions/nvjpeg/cuda_decoder.cpp).  Behavioural contract mirrored from there:
//   * create() refuses NVIMGCODEC_DEVICE_CPU_ONLY (cuda_decoder.cpp:272-273), honours the user's allocators (:221-237)
//   * canDecode() fills every status; acceptance rules follow cuda_decoder.cpp:52-122 minus what this decoder hands
//     to the fallback chain (12-bit, arithmetic coding, regions that leave the image)
//   * decode() is asynchronous and calls imageReady exactly once per sample; the user's stream is ordered after our
//     work with an event before imageReady(SUCCESS) (cuda_decoder.cpp:552-558)
// What is different by design: the reference issues one nvJPEG device call per image from per-thread streams; here the
// executor threads only run the Huffman host stage, and the last one to finish issues ONE H2D copy and ONE launch per
// kernel for the whole batch (three job pages rotate so the next batch's -- or the next piece's -- host stage and H2D copy
// overlap the GPU work in flight).
#include <algorithm>
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <chrono>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

#include "decoder_core.h"
#include "diagnostics.h"
#include "plugin_common.h"
#include "thread_pool.h"
#include "plugin_objects.h"

namespace hipjpeg_ext {

using hipjpeg::DecodeBatch;
using hipjpeg::MemoryHooks;

namespace {

nvimgcodecProcessingStatus_t to_processing_status(hipjpegStatus_t s)
{
    switch (s) {
    case HIPJPEG_STATUS_SUCCESS: return NVIMGCODEC_PROCESSING_STATUS_SUCCESS;
    case HIPJPEG_STATUS_UNSUPPORTED: return NVIMGCODEC_PROCESSING_STATUS_CODESTREAM_UNSUPPORTED;
    case HIPJPEG_STATUS_BAD_JPEG:
    case HIPJPEG_STATUS_TRUNCATED:
    case HIPJPEG_STATUS_CORRUPT: return NVIMGCODEC_PROCESSING_STATUS_IMAGE_CORRUPTED;
    default: return NVIMGCODEC_PROCESSING_STATUS_FAIL;
    }
}

// nvimgcodecSampleFormat_t -> our output layout (reference extensions/nvjpeg/type_convert.cpp:19-41)
bool map_sample_format(nvimgcodecSampleFormat_t f, hipjpegOutputFormat_t* out)
{
    switch (f) {
    case NVIMGCODEC_SAMPLEFORMAT_P_UNCHANGED: *out = HIPJPEG_OUTPUT_YUV_PLANAR; return true;
    case NVIMGCODEC_SAMPLEFORMAT_I_UNCHANGED: *out = HIPJPEG_OUTPUT_RGBI; return true;
    case NVIMGCODEC_SAMPLEFORMAT_P_RGB: *out = HIPJPEG_OUTPUT_RGB_PLANAR; return true;
    case NVIMGCODEC_SAMPLEFORMAT_I_RGB: *out = HIPJPEG_OUTPUT_RGBI; return true;
    case NVIMGCODEC_SAMPLEFORMAT_P_BGR: *out = HIPJPEG_OUTPUT_BGR_PLANAR; return true;
    case NVIMGCODEC_SAMPLEFORMAT_I_BGR: *out = HIPJPEG_OUTPUT_BGRI; return true;
    case NVIMGCODEC_SAMPLEFORMAT_P_Y: *out = HIPJPEG_OUTPUT_Y; return true;
    case NVIMGCODEC_SAMPLEFORMAT_P_YUV: *out = HIPJPEG_OUTPUT_YUV_PLANAR; return true;
    default: return false;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ Decoder object
class HipJpegDecoder {
public:
    HipJpegDecoder(const nvimgcodecFrameworkDesc_t* fw, const nvimgcodecExecutionParams_t* ep, const char* options);
    ~HipJpegDecoder();
    bool ok() const { return ok_; }

    nvimgcodecStatus_t canDecode(nvimgcodecProcessingStatus_t* status, nvimgcodecCodeStreamDesc_t** code_streams,
                                 nvimgcodecImageDesc_t** images, int batch_size, const nvimgcodecDecodeParams_t* params);
    nvimgcodecStatus_t decode(nvimgcodecCodeStreamDesc_t** code_streams, nvimgcodecImageDesc_t** images, int batch_size,
                              const nvimgcodecDecodeParams_t* params);

private:
    nvimgcodecStatus_t decode_chunk(nvimgcodecCodeStreamDesc_t** code_streams, nvimgcodecImageDesc_t** images, int batch_size,
                                    const nvimgcodecDecodeParams_t* params);
    struct Sample {
        nvimgcodecCodeStreamDesc_t* code_stream = nullptr;
        nvimgcodecImageDesc_t* image = nullptr;
        void* user_stream = nullptr;
        const uint8_t* data = nullptr;
        size_t size = 0;
        void* mapped = nullptr;           // non-null if io_stream->map succeeded (must be unmapped)
        std::vector<uint8_t> owned;       // bitstream copy when map() is not available
        nvimgcodecProcessingStatus_t early_status = NVIMGCODEC_PROCESSING_STATUS_SUCCESS;  // set when the sample is rejected before planning
        bool reported = false;            // imageReady has been called (exactly once per sample, whatever happens)
        // what the caller's image descriptor says it can hold
        uint32_t nplanes = 0, plane_w[3] = {0, 0, 0}, plane_h[3] = {0, 0, 0};
        size_t buffer_size = 0, buffer_needed = 0;
    };
    struct Job {
        explicit Job(int device, const MemoryHooks* hooks) : batch(device, hooks) {}
        DecodeBatch batch;
        std::vector<Sample> samples;
        std::vector<hipjpegStatus_t> statuses;
        std::atomic<int> remaining{0};
        std::mutex m;
        std::condition_variable cv;
        bool busy = false;
        bool issued = false;  // H2D copy and kernels are queued on `stream`
        HipJpegDecoder* owner = nullptr;
        hipEvent_t event = nullptr;
        hipStream_t stream = nullptr;  // H2D copy and kernels of this job: jobs overlap each other on the device
    };

    void single_can_decode(nvimgcodecProcessingStatus_t* status, nvimgcodecCodeStreamDesc_t* cs, nvimgcodecImageDesc_t* image,
                           const nvimgcodecDecodeParams_t* params);
    static void host_task(int tid, int sample_idx, void* ctx);
    void issue(Job* job);      // last host task of a job: descriptors, H2D copy, kernels -- nothing here waits for the device
    void complete(Job* job);   // completion thread: device verdicts, user-stream ordering, imageReady, page release
    void report(Job* job, int i, nvimgcodecProcessingStatus_t ps);
    void release_job(Job* job);
    void release_inputs(Job* job);
    void completion_loop();

    const nvimgcodecFrameworkDesc_t* fw_;
    const nvimgcodecExecutionParams_t* ep_;
    MemoryHooks hooks_;
    bool fancy_ = true;  // same default as the reference plugins (nvjpeg_utils.cpp:46, libjpeg_turbo_decoder.cpp:253)
    bool gpu_huffman_ = true;  // entropy-decode eligible streams on the GPU (the reference's GPU_HYBRID backend analogue)
    uint64_t hybrid_huffman_threshold_ = 0;  // ... those of more than this many pixels (cuda_decoder.cpp:188-209; 0 = all of them)
    bool fast_idct_ = false;  // asked for JDCT_FASTEST (libjpeg_turbo_decoder.cpp:250-276): not offered here, see canDecode
    bool ok_ = false;
    int device_ = 0;
    static constexpr int kJobPages = 6;  // batches (or pieces of one) in flight: each has its own page (arenas), stream and event; arenas are sized on first use
    std::unique_ptr<Job> jobs_[kJobPages];
    int next_job_ = 0;
    int pipeline_chunks_ = 0;  // option: pieces a large batch is cut into (0 = choose by size, 1 = never cut)
    std::vector<std::string> unknown_keys_, addressed_keys_;
    // Header parsing of a batch (a marker walk through every file) runs on these threads inside decode(): the framework's
    // executor only takes per-sample tasks that report through imageReady, and the batch layout needs every header first.
    std::unique_ptr<hipjpeg::ForkJoinPool> parse_pool_;
    std::mutex decode_mutex_;  // decode() may be entered from the framework's worker thread and from a fallback re-dispatch
    // Jobs whose device work is queued wait here for the completion thread: no executor thread ever blocks on the GPU, and
    // decode() has long returned when the verdicts of the GPU entropy stage arrive (the host future only means "host work
    // done", reference include/nvimgcodec.h:1455-1459; imageReady is still called once the status is final).
    std::mutex done_mutex_;
    std::condition_variable done_cv_;
    std::deque<Job*> done_queue_;
    bool stopping_ = false;
    std::thread completion_thread_;
};

HipJpegDecoder::HipJpegDecoder(const nvimgcodecFrameworkDesc_t* fw, const nvimgcodecExecutionParams_t* ep, const char* options)
    : fw_(fw), ep_(ep), device_(ep->device_id)
{
    for_each_option(options, kDecoderId, [&](const std::string& key, const std::string& value) {
        std::istringstream v(value);
        if (key == "fancy_upsampling") v >> fancy_;
        else if (key == "gpu_huffman") v >> gpu_huffman_;
        else if (key == "pipeline_chunks") v >> pipeline_chunks_;
        else if (key == "hybrid_huffman_threshold") v >> hybrid_huffman_threshold_;
        else if (key == "fast_idct") v >> fast_idct_;
        else unknown_keys_.push_back(key);
    }, &addressed_keys_);
    // keys addressed to this decoder by name that it does not know: say so (a key without a module name may be meant for another
    // plugin of the chain and is passed over in silence, as the reference's plugins do)
    for (const std::string& k : unknown_keys_)
        if (std::find(addressed_keys_.begin(), addressed_keys_.end(), k) != addressed_keys_.end())
            HJ_LOG_WARNING(fw_, kDecoderId, "unknown option '" << k << "' ignored (known: fancy_upsampling, gpu_huffman, hybrid_huffman_threshold, pipeline_chunks, fast_idct)");
    if (fast_idct_)
        HJ_LOG_WARNING(fw_, kDecoderId, "fast_idct=1: this decoder computes jpeg_idct_islow only (bit-exact with the reference's default); "
                                        "canDecode hands every sample to the next decoder of the chain");
    {
        int threads = 0;
        if (ep->executor && ep->executor->getNumThreads) threads = ep->executor->getNumThreads(ep->executor->instance);
        parse_pool_.reset(new hipjpeg::ForkJoinPool(threads > 0 ? threads : 0));
    }
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
        HJ_LOG_ERROR(fw_, kDecoderId, "no usable HIP device " << device_ << " (found " << count << ")");
        return;
    }
    if (hipSetDevice(device_) != hipSuccess) {
        HJ_LOG_ERROR(fw_, kDecoderId, "could not select HIP device " << device_);
        return;
    }
    for (auto& j : jobs_) {
        j.reset(new Job(device_, &hooks_));
        j->owner = this;
        j->batch.set_gpu_entropy_threshold(hybrid_huffman_threshold_);
        if (hipStreamCreateWithFlags(&j->stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&j->event, hipEventDisableTiming) != hipSuccess) {
            HJ_LOG_ERROR(fw_, kDecoderId, "could not create a HIP stream on device " << device_);
            return;
        }
    }
    completion_thread_ = std::thread([this] { completion_loop(); });
    ok_ = true;
}

HipJpegDecoder::~HipJpegDecoder()
{
    for (auto& j : jobs_) {
        if (!j) continue;
        std::unique_lock<std::mutex> lk(j->m);
        j->cv.wait(lk, [&] { return !j->busy; });
    }
    if (completion_thread_.joinable()) {
        {
            std::lock_guard<std::mutex> lk(done_mutex_);
            stopping_ = true;
        }
        done_cv_.notify_all();
        completion_thread_.join();
    }
    (void)hipSetDevice(device_);
    for (auto& j : jobs_) {
        if (j && j->stream) (void)hipStreamSynchronize(j->stream);
        if (j && j->event) (void)hipEventDestroy(j->event);
        if (j && j->stream) (void)hipStreamDestroy(j->stream);
        j.reset();
    }
}

// What the geometry pass has to do for one sample.  Returns bit 0: the region cannot be used, bit 1: the orientation is not
// one of the eight EXIF ones; *t = the transform to apply (x1 == 0: whole image, orientation 1: as stored).
// nvimgcodecOrientation_t <-> EXIF: reference src/parsers/exif_orientation.h:36-57 (rotated counts counter-clockwise there)
// and extensions/nvjpeg/type_convert.cpp:43-64; region = {start (y, x), end (y, x)}: cuda_decoder.cpp:469-476.
static int sample_transform(const nvimgcodecImageInfo_t& info, const nvimgcodecImageInfo_t& cs_info, const nvimgcodecDecodeParams_t* params,
                            hipjpegTransform_t* t)
{
    *t = hipjpegTransform_t{0, 0, 0, 0, 1};
    int bad = 0;
    if (params->enable_roi && info.region.ndim > 0) {
        const nvimgcodecRegion_t& r = info.region;
        const int W = (int)cs_info.plane_info[0].width, H = (int)cs_info.plane_info[0].height;
        if (r.ndim != 2 || r.start[0] < 0 || r.start[1] < 0 || r.end[0] > H || r.end[1] > W || r.end[0] <= r.start[0] || r.end[1] <= r.start[1]) {
            bad |= 1;
        } else {
            t->y0 = r.start[0];
            t->x0 = r.start[1];
            t->y1 = r.end[0];
            t->x1 = r.end[1];
        }
    }
    if (params->apply_exif_orientation) {
        const nvimgcodecOrientation_t& o = info.orientation;
        const int key = o.rotated * 4 + (o.flip_x ? 2 : 0) + (o.flip_y ? 1 : 0);
        switch (key) {
        case 0: t->orientation = 1; break;
        case 2: t->orientation = 2; break;
        case 180 * 4: t->orientation = 3; break;
        case 1: t->orientation = 4; break;
        case 90 * 4 + 1: t->orientation = 5; break;
        case 270 * 4: t->orientation = 6; break;
        case 270 * 4 + 1: t->orientation = 7; break;
        case 90 * 4: t->orientation = 8; break;
        default: bad |= 2;
        }
    }
    return bad;
}

void HipJpegDecoder::single_can_decode(nvimgcodecProcessingStatus_t* status, nvimgcodecCodeStreamDesc_t* cs, nvimgcodecImageDesc_t* image,
                                       const nvimgcodecDecodeParams_t* params)
{
    *status = NVIMGCODEC_PROCESSING_STATUS_SUCCESS;
    if (fast_idct_) {
        // JDCT_FASTEST gives other pixels than JDCT_ISLOW; the decoder that implements it (libjpeg_turbo_ext) is next in the chain
        *status = NVIMGCODEC_PROCESSING_STATUS_BACKEND_UNSUPPORTED;
        return;
    }
    nvimgcodecJpegImageInfo_t jpeg_info{NVIMGCODEC_STRUCTURE_TYPE_JPEG_IMAGE_INFO, sizeof(nvimgcodecJpegImageInfo_t), nullptr,
                                        NVIMGCODEC_JPEG_ENCODING_UNKNOWN};
    nvimgcodecImageInfo_t cs_info;
    memset(&cs_info, 0, sizeof cs_info);
    cs_info.struct_type = NVIMGCODEC_STRUCTURE_TYPE_IMAGE_INFO;
    cs_info.struct_size = sizeof cs_info;
    cs_info.struct_next = &jpeg_info;
    if (cs->getImageInfo(cs->instance, &cs_info) != NVIMGCODEC_STATUS_SUCCESS) {
        *status = NVIMGCODEC_PROCESSING_STATUS_FAIL;
        return;
    }
    if (strcmp(cs_info.codec_name, "jpeg") != 0) {
        *status = NVIMGCODEC_PROCESSING_STATUS_CODEC_UNSUPPORTED;
        return;
    }
    const nvimgcodecJpegImageInfo_t* ji = find_in_chain<nvimgcodecJpegImageInfo_t>(cs_info.struct_next, NVIMGCODEC_STRUCTURE_TYPE_JPEG_IMAGE_INFO);
    if (ji && ji->encoding != NVIMGCODEC_JPEG_ENCODING_UNKNOWN && ji->encoding != NVIMGCODEC_JPEG_ENCODING_BASELINE_DCT &&
        ji->encoding != NVIMGCODEC_JPEG_ENCODING_EXTENDED_SEQUENTIAL_DCT_HUFFMAN && ji->encoding != NVIMGCODEC_JPEG_ENCODING_PROGRESSIVE_DCT_HUFFMAN) {
        *status = NVIMGCODEC_PROCESSING_STATUS_ENCODING_UNSUPPORTED;
        return;
    }
    // 12-bit streams belong to another decoder in the chain.  Four-component streams (CMYK / YCCK) are taken like the
    // reference's plugins take them (extensions/nvjpeg/cuda_decoder.cpp:85-89; pixels as extensions/libjpeg_turbo/
    // jpeg_mem.cpp:292-337 makes them), except as raw component planes.
    if (cs_info.num_planes > 0 && cs_info.plane_info[0].sample_type != NVIMGCODEC_SAMPLE_DATA_TYPE_UINT8) {
        *status = NVIMGCODEC_PROCESSING_STATUS_CODESTREAM_UNSUPPORTED;
        return;
    }
    const bool four_components = cs_info.color_spec == NVIMGCODEC_COLORSPEC_CMYK || cs_info.color_spec == NVIMGCODEC_COLORSPEC_YCCK;
    // the framework's parser names no sampling for four components (reference src/parsers/jpeg.cpp:70-114 returns UNSUPPORTED)
    if (!four_components && cs_info.chroma_subsampling == NVIMGCODEC_SAMPLING_UNSUPPORTED) *status |= NVIMGCODEC_PROCESSING_STATUS_SAMPLING_UNSUPPORTED;

    nvimgcodecImageInfo_t info;
    memset(&info, 0, sizeof info);
    info.struct_type = NVIMGCODEC_STRUCTURE_TYPE_IMAGE_INFO;
    info.struct_size = sizeof info;
    if (image->getImageInfo(image->instance, &info) != NVIMGCODEC_STATUS_SUCCESS) {
        *status = NVIMGCODEC_PROCESSING_STATUS_FAIL;
        return;
    }
    switch (info.color_spec) {
    case NVIMGCODEC_COLORSPEC_UNCHANGED:
    case NVIMGCODEC_COLORSPEC_SRGB:
    case NVIMGCODEC_COLORSPEC_GRAY:
    case NVIMGCODEC_COLORSPEC_SYCC: break;
    default: *status |= NVIMGCODEC_PROCESSING_STATUS_COLOR_SPEC_UNSUPPORTED;
    }
    hipjpegOutputFormat_t fmt = HIPJPEG_OUTPUT_RGBI;
    const bool fmt_ok = map_sample_format(info.sample_format, &fmt);
    if (!fmt_ok) {
        *status |= NVIMGCODEC_PROCESSING_STATUS_SAMPLE_FORMAT_UNSUPPORTED;
    } else {
        const bool interleaved = fmt == HIPJPEG_OUTPUT_RGBI || fmt == HIPJPEG_OUTPUT_BGRI;
        const bool planar_rgb = fmt == HIPJPEG_OUTPUT_RGB_PLANAR || fmt == HIPJPEG_OUTPUT_BGR_PLANAR;
        if (interleaved) {
            if (info.num_planes != 1) *status |= NVIMGCODEC_PROCESSING_STATUS_NUM_PLANES_UNSUPPORTED;
            if (info.plane_info[0].num_channels != 3) *status |= NVIMGCODEC_PROCESSING_STATUS_NUM_CHANNELS_UNSUPPORTED;
        } else if (planar_rgb) {
            if (info.num_planes != 3) *status |= NVIMGCODEC_PROCESSING_STATUS_NUM_PLANES_UNSUPPORTED;
        } else if (fmt == HIPJPEG_OUTPUT_Y) {
            if (info.num_planes != 1) *status |= NVIMGCODEC_PROCESSING_STATUS_NUM_PLANES_UNSUPPORTED;
            if (info.plane_info[0].num_channels != 1) *status |= NVIMGCODEC_PROCESSING_STATUS_NUM_CHANNELS_UNSUPPORTED;
        } else if (four_components) {
            *status |= NVIMGCODEC_PROCESSING_STATUS_SAMPLE_FORMAT_UNSUPPORTED;  // P_YUV / P_UNCHANGED of a CMYK / YCCK stream
        } else if (info.num_planes != cs_info.num_planes) {
            *status |= NVIMGCODEC_PROCESSING_STATUS_NUM_PLANES_UNSUPPORTED;
        }
    }
    for (uint32_t p = 0; p < info.num_planes && p < NVIMGCODEC_MAX_NUM_PLANES; ++p)
        if (info.plane_info[p].sample_type != NVIMGCODEC_SAMPLE_DATA_TYPE_UINT8) *status |= NVIMGCODEC_PROCESSING_STATUS_SAMPLE_TYPE_UNSUPPORTED;
    // Region of interest and EXIF orientation run as a geometry pass on the device (decoder_core.cpp).  What it cannot do
    // is reported here, which sends the sample to the next decoder in the priority chain (reference
    // src/decoder_worker.cpp:275-296): regions that are not 2-D or leave the image (extensions/libjpeg_turbo/
    // libjpeg_turbo_decoder.cpp:355-370), orientations outside the eight EXIF ones (extensions/nvjpeg/type_convert.cpp:43-64),
    // and either of them on subsampled planes.
    hipjpegTransform_t t;
    const int geometry = sample_transform(info, cs_info, params, &t);
    if (geometry & 1) *status |= NVIMGCODEC_PROCESSING_STATUS_ROI_UNSUPPORTED;
    if (geometry & 2) *status |= NVIMGCODEC_PROCESSING_STATUS_ORIENTATION_UNSUPPORTED;
    if (!geometry && fmt_ok && fmt == HIPJPEG_OUTPUT_YUV_PLANAR && (t.orientation != 1 || t.x1 != 0)) {
        if (t.x1 != 0) *status |= NVIMGCODEC_PROCESSING_STATUS_ROI_UNSUPPORTED;
        if (t.orientation != 1) *status |= NVIMGCODEC_PROCESSING_STATUS_ORIENTATION_UNSUPPORTED;
    }
}

nvimgcodecStatus_t HipJpegDecoder::canDecode(nvimgcodecProcessingStatus_t* status, nvimgcodecCodeStreamDesc_t** code_streams,
                                             nvimgcodecImageDesc_t** images, int batch_size, const nvimgcodecDecodeParams_t* params)
{
    if (!status || !code_streams || !images || !params) return NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER;
    for (int i = 0; i < batch_size; i++) {
        if (!code_streams[i] || !images[i]) {
            status[i] = NVIMGCODEC_PROCESSING_STATUS_FAIL;
            continue;
        }
        single_can_decode(&status[i], code_streams[i], images[i], params);
    }
    return NVIMGCODEC_STATUS_SUCCESS;
}

void HipJpegDecoder::release_inputs(Job* job)
{
    for (Sample& s : job->samples) {
        if (s.mapped) {
            nvimgcodecIoStreamDesc_t* io = s.code_stream->io_stream;
            io->unmap(io->instance, s.mapped, s.size);
            s.mapped = nullptr;
        }
    }
}

// Exactly one imageReady per sample: every path that reports goes through here.
void HipJpegDecoder::report(Job* job, int i, nvimgcodecProcessingStatus_t ps)
{
    Sample& s = job->samples[i];
    if (s.reported) return;
    s.reported = true;
    if (ps != NVIMGCODEC_PROCESSING_STATUS_SUCCESS)
        HJ_LOG_WARNING(fw_, kDecoderId, "sample " << i << " not decoded, processing status 0x" << std::hex << ps);
    s.image->imageReady(s.image->instance, ps);
}

void HipJpegDecoder::release_job(Job* job)
{
    {
        std::lock_guard<std::mutex> lk(job->m);
        job->busy = false;
    }
    job->cv.notify_all();
}

// Runs on the framework's executor threads (or inline for a single sample).  Nothing may escape: the executor swallows
// exceptions (reference src/thread_pool.cpp:175-185), and a sample that never reports deadlocks the caller's future.
void HipJpegDecoder::host_task(int /*tid*/, int sample_idx, void* ctx)
{
    Job* job = static_cast<Job*>(ctx);
    try {
        hipjpeg::ScopedRange range("hipjpeg_decoder host task");
        job->batch.entropy_stage(sample_idx);
    } catch (...) {
        job->batch.reject(sample_idx, HIPJPEG_STATUS_INTERNAL_ERROR);
    }
    if (job->remaining.fetch_sub(1) == 1) job->owner->issue(job);
}

// Runs on whichever thread finished the last host task of the job: queue the device work, hand the job to the completion
// thread.  Does not wait for the device.
void HipJpegDecoder::issue(Job* job)
{
    bool ok = false;
    try {
        ok = hipSetDevice(device_) == hipSuccess;
        job->batch.finalize(job->statuses.data());
        if (ok) ok = job->batch.transfer(job->stream) == HIPJPEG_STATUS_SUCCESS;
        if (ok) ok = job->batch.launch(job->stream) == HIPJPEG_STATUS_SUCCESS;
    } catch (...) {
        ok = false;
    }
    job->issued = ok;
    {
        std::lock_guard<std::mutex> lk(done_mutex_);
        done_queue_.push_back(job);
    }
    done_cv_.notify_one();
}

void HipJpegDecoder::completion_loop()
{
    (void)hipSetDevice(device_);
    for (;;) {
        Job* job = nullptr;
        {
            std::unique_lock<std::mutex> lk(done_mutex_);
            done_cv_.wait(lk, [&] { return stopping_ || !done_queue_.empty(); });
            if (done_queue_.empty()) return;  // stopping, nothing left
            job = done_queue_.front();
            done_queue_.pop_front();
        }
        complete(job);
    }
}

// GPU-decoded streams: their verdicts come back from the device (resolve() waits for this job's kernels; the other job
// pages keep going meanwhile).  Jobs that went through the host entropy stage only do not wait here at all.
void HipJpegDecoder::complete(Job* job)
{
    const int n = (int)job->samples.size();
    bool gpu_ok = job->issued;
    try {
        if (gpu_ok) gpu_ok = job->batch.resolve(job->stream) == HIPJPEG_STATUS_SUCCESS;
        if (gpu_ok) gpu_ok = hipEventRecord(job->event, job->stream) == hipSuccess;
        for (int i = 0; i < n; i++) job->statuses[i] = job->batch.image(i).status;  // incl. what the GPU entropy stage reported
    } catch (...) {
        gpu_ok = false;
    }
    if (!gpu_ok) (void)hipStreamSynchronize(job->stream);  // whatever was queued must not outlive the caller's buffers
    try {
        release_inputs(job);
    } catch (...) {
    }
    // the consumers' streams must not run ahead of the decode (reference cuda_decoder.cpp:552-556): one wait per DISTINCT stream of
    // the job -- a batch's samples nearly always share one, and 256 runtime calls per batch were a quarter of this thread's time
    void* waited[4] = {nullptr, nullptr, nullptr, nullptr};
    bool waited_ok[4] = {false, false, false, false};
    int nwaited = 0;
    auto order_after_decode = [&](void* user_stream) {
        for (int k = 0; k < nwaited; k++)
            if (waited[k] == user_stream) return waited_ok[k];
        const bool ok = hipStreamWaitEvent((hipStream_t)user_stream, job->event, 0) == hipSuccess;
        if (nwaited < 4) {
            waited[nwaited] = user_stream;
            waited_ok[nwaited++] = ok;
        }
        return ok;
    };
    for (int i = 0; i < n; i++) {
        Sample& s = job->samples[i];
        nvimgcodecProcessingStatus_t ps = s.early_status;
        if (ps == NVIMGCODEC_PROCESSING_STATUS_SUCCESS) {
            ps = gpu_ok ? to_processing_status(job->statuses[i]) : (nvimgcodecProcessingStatus_t)NVIMGCODEC_PROCESSING_STATUS_FAIL;
            if (ps == NVIMGCODEC_PROCESSING_STATUS_SUCCESS && !order_after_decode(s.user_stream)) ps = NVIMGCODEC_PROCESSING_STATUS_FAIL;
        }
        try {
            report(job, i, ps);
        } catch (...) {  // a framework whose imageReady throws (double set) must not take the page down with it
        }
    }
    release_job(job);
}

// A large batch is cut into up to three pieces, each a job of its own (own page, own stream): the host stage and the H2D copy of
// one piece run beside the kernels of the piece before it, exactly as they do between consecutive decode() calls.  Samples
// report through imageReady piece by piece.
nvimgcodecStatus_t HipJpegDecoder::decode(nvimgcodecCodeStreamDesc_t** code_streams, nvimgcodecImageDesc_t** images, int batch_size,
                                          const nvimgcodecDecodeParams_t* params)
{
    if (!code_streams || !images || !params) return NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER;
    if (batch_size < 1) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    std::lock_guard<std::mutex> serial(decode_mutex_);
    int chunks = pipeline_chunks_ > 0 ? pipeline_chunks_ : (batch_size >= 192 ? 3 : batch_size >= 96 ? 2 : 1);
    if (pipeline_chunks_ <= 0) {
        // a caller that keeps several decode() calls outstanding already has what the pieces are for -- the host stage and copy of
        // one call beside the kernels of the call before -- and whole batches use the latency-bound entropy kernels better than thirds
        // of them do: with a page still busy the batch goes out in one piece
        for (auto& j : jobs_) {
            std::lock_guard<std::mutex> lk(j->m);
            if (j->busy) chunks = 1;
        }
    }
    chunks = std::max(1, std::min(chunks, std::min(batch_size, kJobPages)));
    nvimgcodecStatus_t result = NVIMGCODEC_STATUS_SUCCESS;
    for (int c = 0; c < chunks; c++) {
        const int lo = (int)((long long)batch_size * c / chunks), hi = (int)((long long)batch_size * (c + 1) / chunks);
        const nvimgcodecStatus_t st = decode_chunk(code_streams + lo, images + lo, hi - lo, params);
        if (st != NVIMGCODEC_STATUS_SUCCESS) result = st;  // that piece's samples have been reported failed; the others go on
    }
    return result;
}

nvimgcodecStatus_t HipJpegDecoder::decode_chunk(nvimgcodecCodeStreamDesc_t** code_streams, nvimgcodecImageDesc_t** images, int batch_size,
                                                const nvimgcodecDecodeParams_t* params)
{
    static const bool timing = getenv("HIPJPEG_DEBUG_TIMING") != nullptr;  // debug aid: phase times of decode() on stderr
    const auto t_enter = std::chrono::steady_clock::now();
    hipjpeg::ScopedRange range("hipjpeg_decoder decode (marshal + plan + schedule)");
    Job* job = jobs_[next_job_].get();
    next_job_ = (next_job_ + 1) % kJobPages;
    {
        std::unique_lock<std::mutex> lk(job->m);
        job->cv.wait(lk, [&] { return !job->busy; });
        job->busy = true;
    }
    const int n = batch_size;
    // From here on the page is ours and every sample owes the framework one imageReady.  Whatever is thrown before the host
    // tasks are scheduled ends in the handler at the bottom: all samples FAIL, page released, error code returned
    // (reference extensions/nvjpeg/cuda_decoder.cpp:602-608).
    bool scheduled = false;
    try {
        job->issued = false;
        job->samples.assign(n, Sample());
        for (int i = 0; i < n; i++) {
            job->samples[i].code_stream = code_streams[i];
            job->samples[i].image = images[i];
        }
        job->statuses.assign(n, HIPJPEG_STATUS_SUCCESS);
        std::vector<const uint8_t*> data(n, nullptr);
        std::vector<size_t> sizes(n, 0);
        std::vector<hipjpegOutput_t> outs(n);
        std::vector<hipjpegOutputFormat_t> formats(n, HIPJPEG_OUTPUT_RGBI);
        std::vector<hipjpegTransform_t> geometry(n, hipjpegTransform_t{0, 0, 0, 0, 1});
        bool any_geometry = false;
        memset(outs.data(), 0, sizeof(hipjpegOutput_t) * n);

        auto marshal = [&](int i) {
            Sample& s = job->samples[i];
            hipjpeg::fault_point("marshal");
            nvimgcodecImageInfo_t info;
            memset(&info, 0, sizeof info);
            info.struct_type = NVIMGCODEC_STRUCTURE_TYPE_IMAGE_INFO;
            info.struct_size = sizeof info;
            if (s.image->getImageInfo(s.image->instance, &info) != NVIMGCODEC_STATUS_SUCCESS) {
                s.early_status = NVIMGCODEC_PROCESSING_STATUS_FAIL;
                return;
            }
            s.user_stream = info.cuda_stream;
            if (info.buffer_kind != NVIMGCODEC_IMAGE_BUFFER_KIND_STRIDED_DEVICE || !info.buffer) {
                // the framework bounces host buffers for GPU backends (reference src/work.h:144-169); a host pointer here is a caller bug
                s.early_status = NVIMGCODEC_PROCESSING_STATUS_FAIL;
                return;
            }
            if (!map_sample_format(info.sample_format, &formats[i])) {
                s.early_status = NVIMGCODEC_PROCESSING_STATUS_SAMPLE_FORMAT_UNSUPPORTED;
                return;
            }
            if ((params->enable_roi && info.region.ndim > 0) || params->apply_exif_orientation) {
                nvimgcodecImageInfo_t cs_info;
                memset(&cs_info, 0, sizeof cs_info);
                cs_info.struct_type = NVIMGCODEC_STRUCTURE_TYPE_IMAGE_INFO;
                cs_info.struct_size = sizeof cs_info;
                if (s.code_stream->getImageInfo(s.code_stream->instance, &cs_info) != NVIMGCODEC_STATUS_SUCCESS) {
                    s.early_status = NVIMGCODEC_PROCESSING_STATUS_FAIL;
                    return;
                }
                const int bad = sample_transform(info, cs_info, params, &geometry[i]);
                if (bad) {  // canDecode said so already; a caller that insists gets the same verdict (cuda_decoder.cpp:452-461)
                    s.early_status = (bad & 1) ? NVIMGCODEC_PROCESSING_STATUS_ROI_UNSUPPORTED : NVIMGCODEC_PROCESSING_STATUS_ORIENTATION_UNSUPPORTED;
                    return;
                }
                any_geometry = any_geometry || geometry[i].x1 != 0 || geometry[i].orientation != 1;
            }
            // planes are laid out back to back inside `buffer` (reference cuda_decoder.cpp:532-538)
            uint8_t* p = static_cast<uint8_t*>(info.buffer);
            s.nplanes = std::min<uint32_t>(info.num_planes, 3u);
            s.buffer_size = info.buffer_size;
            for (uint32_t pl = 0; pl < s.nplanes; pl++) {
                if (info.plane_info[pl].row_stride > 0xFFFFFFFFull) {  // the kernels address rows with 32-bit pitches
                    s.early_status = NVIMGCODEC_PROCESSING_STATUS_FAIL;
                    return;
                }
                outs[i].plane[pl] = p;
                outs[i].pitch[pl] = (uint32_t)info.plane_info[pl].row_stride;
                s.plane_w[pl] = info.plane_info[pl].width;
                s.plane_h[pl] = info.plane_info[pl].height;
                p += info.plane_info[pl].row_stride * info.plane_info[pl].height;
            }
            // bitstream: zero-copy map when the stream offers it, else read into our own buffer (cuda_decoder.cpp:480-500)
            nvimgcodecIoStreamDesc_t* io = s.code_stream->io_stream;
            size_t size = 0;
            if (io->size(io->instance, &size) != NVIMGCODEC_STATUS_SUCCESS || size == 0) {
                s.early_status = NVIMGCODEC_PROCESSING_STATUS_IMAGE_CORRUPTED;
                return;
            }
            void* mapped = nullptr;
            if (io->map(io->instance, &mapped, 0, size) == NVIMGCODEC_STATUS_SUCCESS && mapped) {
                s.mapped = mapped;
                s.data = static_cast<const uint8_t*>(mapped);
            } else {
                s.owned.resize(size);
                size_t got = 0;
                io->seek(io->instance, 0, SEEK_SET);
                if (io->read(io->instance, &got, s.owned.data(), size) != NVIMGCODEC_STATUS_SUCCESS || got != size) {
                    s.early_status = NVIMGCODEC_PROCESSING_STATUS_IMAGE_CORRUPTED;
                    return;
                }
                s.data = s.owned.data();
            }
            s.size = size;
            data[i] = s.data;
            sizes[i] = size;
        };
        for (int i = 0; i < n; i++) {
            try {  // one sample's trouble (a throwing getImageInfo, no memory for its bitstream copy) stays that sample's
                marshal(i);
            } catch (...) {
                job->samples[i].early_status = NVIMGCODEC_PROCESSING_STATUS_FAIL;
                data[i] = nullptr;
                sizes[i] = 0;
            }
        }

        const auto t_marshal = std::chrono::steady_clock::now();
        const bool planned = hipSetDevice(device_) == hipSuccess &&
                             job->batch.plan(data.data(), sizes.data(), n, outs.data(), HIPJPEG_OUTPUT_RGBI,
                                             (fancy_ ? HIPJPEG_FLAG_FANCY_UPSAMPLING : 0u) | (gpu_huffman_ ? HIPJPEG_FLAG_GPU_HUFFMAN : 0u),
                                             job->statuses.data(), formats.data(), parse_pool_.get(),
                                             any_geometry ? geometry.data() : nullptr) == HIPJPEG_STATUS_SUCCESS;
        if (!planned) throw std::runtime_error("could not plan the decode batch");

        // The caller's image descriptor must be able to hold what will be written: plane sizes and buffer_size against the
        // decoded size after region of interest and orientation (nvJPEG checks its output descriptor for the reference).
        for (int i = 0; i < n; i++) {
            Sample& s = job->samples[i];
            if (s.early_status != NVIMGCODEC_PROCESSING_STATUS_SUCCESS || job->statuses[i] != HIPJPEG_STATUS_SUCCESS) continue;
            const hipjpeg::PlannedImage& im = job->batch.image(i);
            int ow = 0, oh = 0;
            job->batch.output_size(i, &ow, &oh);
            bool fits = true;
            size_t need = 0;
            const int planes = formats[i] == HIPJPEG_OUTPUT_YUV_PLANAR ? im.frame.ncomp
                               : (formats[i] == HIPJPEG_OUTPUT_RGB_PLANAR || formats[i] == HIPJPEG_OUTPUT_BGR_PLANAR) ? 3 : 1;
            if ((int)s.nplanes < planes) fits = false;
            for (int pl = 0; pl < planes && fits; pl++) {
                const int pw = formats[i] == HIPJPEG_OUTPUT_YUV_PLANAR ? im.frame.comp[pl].samp_w : ow;
                const int ph = formats[i] == HIPJPEG_OUTPUT_YUV_PLANAR ? im.frame.comp[pl].samp_h : oh;
                if ((int64_t)s.plane_w[pl] < pw || (int64_t)s.plane_h[pl] < ph) fits = false;
                need += (size_t)outs[i].pitch[pl] * s.plane_h[pl];
            }
            if (fits && s.buffer_size != 0 && s.buffer_size < need) fits = false;
            if (!fits) {
                job->batch.reject(i, HIPJPEG_STATUS_INVALID_ARGUMENT);
                job->statuses[i] = HIPJPEG_STATUS_INVALID_ARGUMENT;
            }
        }

        const auto t_plan = std::chrono::steady_clock::now();
        job->remaining.store(n);
        scheduled = true;  // from the first launch on, the host tasks own the samples
        nvimgcodecExecutorDesc_t* ex = ep_->executor;
        if (n == 1 || !ex) {
            for (int i = 0; i < n; i++) host_task(0, i, job);  // single image: run inline like the reference (:565-566)
        } else {
            for (int i = 0; i < n; i++) {
                bool handed = false;
                try {
                    handed = ex->launch(ex->instance, device_, i, job, &HipJpegDecoder::host_task) == NVIMGCODEC_STATUS_SUCCESS;
                } catch (...) {
                    handed = false;
                }
                if (!handed) host_task(0, i, job);  // host_task itself lets nothing escape
            }
        }
        if (timing) {
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            const auto t_end = std::chrono::steady_clock::now();
            fprintf(stderr, "[hipjpeg] plugin decode(%d): wait for a job page + marshal %.2f ms, plan %.2f ms, hand to executor %.2f ms\n", n,
                    ms(t_enter, t_marshal), ms(t_marshal, t_plan), ms(t_plan, t_end));
        }
        return NVIMGCODEC_STATUS_SUCCESS;
    } catch (...) {
        if (scheduled) return NVIMGCODEC_STATUS_SUCCESS;  // every sample is in the hands of a host task: they report
        try {
            release_inputs(job);
        } catch (...) {
        }
        for (int i = 0; i < n && i < (int)job->samples.size(); i++) {
            try {
                if (job->samples[i].image) report(job, i, NVIMGCODEC_PROCESSING_STATUS_FAIL);
            } catch (...) {
            }
        }
        if ((int)job->samples.size() < n)  // not even the sample table could be built: report straight from the arguments
            for (int i = (int)job->samples.size(); i < n; i++) images[i]->imageReady(images[i]->instance, NVIMGCODEC_PROCESSING_STATUS_FAIL);
        release_job(job);
        HJ_LOG_ERROR(fw_, kDecoderId, "decode batch of " << n << " samples failed before it was scheduled on device " << device_);
        return NVIMGCODEC_STATUS_EXTENSION_EXECUTION_FAILED;
    }
}

// ------------------------------------------------------------------------------------------------ plugin (factory) object
HipJpegDecoderPlugin::HipJpegDecoderPlugin(const nvimgcodecFrameworkDesc_t* framework)
    : desc_{NVIMGCODEC_STRUCTURE_TYPE_DECODER_DESC, sizeof(nvimgcodecDecoderDesc_t), nullptr, this, kDecoderId, "jpeg",
            NVIMGCODEC_BACKEND_KIND_HYBRID_CPU_GPU, static_create, static_destroy, static_can_decode, static_decode},
      framework_(framework)
{
}

nvimgcodecStatus_t HipJpegDecoderPlugin::static_create(void* instance, nvimgcodecDecoder_t* decoder, const nvimgcodecExecutionParams_t* exec_params,
                                                       const char* options)
{
    try {
        if (!instance || !decoder || !exec_params) return NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER;
        auto* self = static_cast<HipJpegDecoderPlugin*>(instance);
        if (exec_params->device_id == NVIMGCODEC_DEVICE_CPU_ONLY) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
        std::unique_ptr<HipJpegDecoder> d(new HipJpegDecoder(self->framework_, exec_params, options));
        if (!d->ok()) return NVIMGCODEC_STATUS_EXTENSION_CUDA_CALL_ERROR;  // no silent CPU fallback: without a GPU this decoder does not exist
        *decoder = reinterpret_cast<nvimgcodecDecoder_t>(d.release());
        return NVIMGCODEC_STATUS_SUCCESS;
    } catch (...) {
        return NVIMGCODEC_STATUS_EXTENSION_INTERNAL_ERROR;
    }
}

nvimgcodecStatus_t HipJpegDecoderPlugin::static_destroy(nvimgcodecDecoder_t decoder)
{
    try {
        if (!decoder) return NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER;
        delete reinterpret_cast<HipJpegDecoder*>(decoder);
        return NVIMGCODEC_STATUS_SUCCESS;
    } catch (...) {
        return NVIMGCODEC_STATUS_EXTENSION_INTERNAL_ERROR;
    }
}

nvimgcodecStatus_t HipJpegDecoderPlugin::static_can_decode(nvimgcodecDecoder_t decoder, nvimgcodecProcessingStatus_t* status,
                                                           nvimgcodecCodeStreamDesc_t** code_streams, nvimgcodecImageDesc_t** images, int batch_size,
                                                           const nvimgcodecDecodeParams_t* params)
{
    try {
        if (!decoder) return NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER;
        return reinterpret_cast<HipJpegDecoder*>(decoder)->canDecode(status, code_streams, images, batch_size, params);
    } catch (...) {
        return NVIMGCODEC_STATUS_EXTENSION_INTERNAL_ERROR;
    }
}

nvimgcodecStatus_t HipJpegDecoderPlugin::static_decode(nvimgcodecDecoder_t decoder, nvimgcodecCodeStreamDesc_t** code_streams,
                                                       nvimgcodecImageDesc_t** images, int batch_size, const nvimgcodecDecodeParams_t* params)
{
    try {
        if (!decoder) return NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER;
        return reinterpret_cast<HipJpegDecoder*>(decoder)->decode(code_streams, images, batch_size, params);
    } catch (...) {
        // never let an exception cross the C boundary; nothing was scheduled if we got here before the launch loop
        return NVIMGCODEC_STATUS_EXTENSION_INTERNAL_ERROR;
    }
}

}  // namespace hipjpeg_ext
