// hipjpeg_api.cpp -- the C-ABI declared in include/hipjpeg.h (decode side).
#include <hip/hip_runtime_api.h>

#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "../../include/hipjpeg.h"
#include "decoder_core.h"
#include "diagnostics.h"
#include "encoder_core.h"
#include "entropy_decode.h"
#include "gpu_huffman_host.h"
#include "progressive_gpu_host.h"
#include "thread_pool.h"

#include <chrono>
#include <future>
#include <cstdio>
#include <cstdlib>

using namespace hipjpeg;

namespace {
// No C++ exception may cross the C boundary (the reference wraps every entry point the same way, e.g.
// extensions/libjpeg_turbo/libjpeg_turbo_decoder.cpp:226-236): whatever is thrown below becomes a status code.
template <class F>
hipjpegStatus_t guarded(F&& body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return HIPJPEG_STATUS_ALLOC_FAILED;
    } catch (...) {
        return HIPJPEG_STATUS_INTERNAL_ERROR;
    }
}
}  // namespace

struct hipjpegHandle {
    int device_id = 0;
    MemoryHooks hooks;
    std::unique_ptr<ForkJoinPool> pool;
    // three batch pages, used in turn: the host stage and H2D copy of batches k+1 and k+2 proceed while the device is still
    // consuming batch k (same idea as the reference's two pinned pages per thread, cuda_decoder.h:50-53; the third page
    // keeps the copy engine a whole batch ahead of the kernels)
    // (hipjpegSetPipelineDepth: up to kMaxPages, for work whose batches leave most of the chip idle -- progressive scans)
    static constexpr int kMaxPages = 8;
    int pages = 3;
    std::unique_ptr<DecodeBatch> batches[kMaxPages];
    int current = 0;
    DecodeBatch& cur() { return *batches[current]; }
    // pipelined submission (hipjpegDecodeBatchSubmit / Wait): pages in flight, oldest first, with the stream each runs on
    std::vector<hipjpegTransform_t> transforms;  // geometry for the next batch (hipjpegDecodeBatchSetTransforms)
    int submitted[kMaxPages] = {-1, -1, -1, -1, -1, -1, -1, -1};
    void* submitted_stream[kMaxPages] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int num_submitted = 0;
    hipStream_t copy_stream = nullptr;  // H2D copies of submitted batches: they overlap the kernels of the batch before
    hipStream_t entropy_stream = nullptr;  // GPU entropy stage of submitted batches: beside the pixel kernels of the batch before
    // batches with progressive images: the walk of a progressive scan is one wave per scan and leaves most of the chip idle, so
    // the entropy stages of consecutive batches run beside EACH OTHER, one stream per page
    hipStream_t page_entropy_stream[kMaxPages] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::unique_ptr<EncodeBatch> encode;
    EncodeBatch* encode_view = nullptr;  // the batch hipjpegEncodeGetBitstream / GetCoefficients / Stats talk about
    // pipelined encoding (hipjpegEncodeBatchSubmit / Wait): three pages, each driven by its own host thread on its own
    // stream, so that the PCIe-bound file output and the two short host round trips of one batch overlap the kernels of the
    // others (two pages fall into lock-step: both batches compute, then both wait for PCIe)
    struct EncodePage {
        std::unique_ptr<EncodeBatch> batch;
        hipStream_t stream = nullptr;
        hipEvent_t ready = nullptr;
        std::future<hipjpegStatus_t> result;
        std::vector<hipjpegEncodeInput_t> inputs;
        std::vector<hipjpegEncodeParams_t> params;
    };
    static constexpr int kEncodePages = 3;
    EncodePage encode_pages[kEncodePages];
    int encode_next = 0, encode_oldest = 0, encode_in_flight = 0;
};

extern "C" {

const char* hipjpegStatusString(hipjpegStatus_t s)
{
    switch (s) {
    case HIPJPEG_STATUS_SUCCESS: return "success";
    case HIPJPEG_STATUS_INVALID_ARGUMENT: return "invalid argument";
    case HIPJPEG_STATUS_BAD_JPEG: return "not a JPEG or malformed marker segment";
    case HIPJPEG_STATUS_UNSUPPORTED: return "JPEG feature outside this decoder's scope";
    case HIPJPEG_STATUS_TRUNCATED: return "entropy-coded data ends early";
    case HIPJPEG_STATUS_CORRUPT: return "corrupt entropy-coded data";
    case HIPJPEG_STATUS_ALLOC_FAILED: return "memory allocation failed";
    case HIPJPEG_STATUS_HIP_ERROR: return "HIP runtime error";
    case HIPJPEG_STATUS_NO_DEVICE: return "no usable HIP device";
    case HIPJPEG_STATUS_BUFFER_TOO_SMALL: return "buffer too small";
    case HIPJPEG_STATUS_INTERNAL_ERROR: return "internal error (exception caught at the C boundary)";
    }
    return "unknown status";
}

int hipjpegVersion(void) { return 200; }

// Batches in flight run their entropy stages on streams of their own (decoder pages, plugin jobs); the HIP runtime multiplexes streams
// onto GPU_MAX_HW_QUEUES hardware queues -- four by default -- and kernels of streams that share a queue run one after the other: with
// six progressive batches in flight their 80 ms walks queued up behind each other.  The runtime reads the variable when
// it initialises, i.e. at the process's first HIP call; this library is normally loaded before that (an extension module is opened when
// the instance is created, the Python package on import), so it asks for twelve queues itself unless the environment already says
// otherwise.  A process that has initialised HIP earlier keeps its setting: nothing breaks, deep pipelines of progressive batches are
// slower.  (the figure must not depend on the caller exporting the variable.)
__attribute__((constructor)) static void hipjpeg_runtime_defaults() { (void)setenv("GPU_MAX_HW_QUEUES", "12", /*overwrite=*/0); }

// The hipjpegTest* entry points exist for the test suite (fault injection, counters the tests assert on).  They answer only in a
// process that was started with HIPJPEG_ENABLE_TEST_HOOKS=1 (read once); anywhere else they refuse -- no caller of a production process can
// arm a throw inside the library.  tests/conftest.py sets the variable; bench.py and the tools do not need the hooks.
static bool test_hooks_enabled()
{
    static const bool on = [] {
        const char* e = getenv("HIPJPEG_ENABLE_TEST_HOOKS");
        return e != nullptr && e[0] == '1';
    }();
    return on;
}

hipjpegStatus_t hipjpegTestSetFault(const char* site, int countdown)
{
    return guarded([&]() -> hipjpegStatus_t {
        if (!test_hooks_enabled()) return HIPJPEG_STATUS_INVALID_ARGUMENT;
        set_fault(site, countdown);
        return HIPJPEG_STATUS_SUCCESS;
    });
}

hipjpegStatus_t hipjpegGetImageInfo(const uint8_t* data, size_t length, hipjpegImageInfo_t* info)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!data || !info) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    FrameInfo f;
    ParseStatus ps = parse_jpeg(data, length, &f, /*headers_only=*/false);
    memset(info, 0, sizeof *info);
    info->sof_marker = f.sof;
    if (ps != kParseOk) return status_from_parse(ps);
    info->width = f.width;
    info->height = f.height;
    info->num_components = f.ncomp;
    info->color_model = (int)f.color;
    info->subsampling = classify_subsampling(f);
    info->restart_interval = f.scans.empty() ? 0 : f.scans[0].restart_interval;
    info->num_scans = (int)f.scans.size();
    for (int c = 0; c < f.ncomp; c++) {
        info->h[c] = f.comp[c].h;
        info->v[c] = f.comp[c].v;
        info->blocks_w[c] = f.comp[c].blocks_w;
        info->blocks_h[c] = f.comp[c].blocks_h;
        info->samp_w[c] = f.comp[c].samp_w;
        info->samp_h[c] = f.comp[c].samp_h;
    }
    info->coef_bytes = f.total_blocks() * 128;
    return HIPJPEG_STATUS_SUCCESS;
    });
}

hipjpegStatus_t hipjpegEntropyDecodeHost(const uint8_t* data, size_t length, int16_t* coef, size_t coef_capacity_bytes,
                                         uint64_t comp_offsets[4], uint16_t qtables[256])
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!data || !coef) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    FrameInfo f;
    ParseStatus ps = parse_jpeg(data, length, &f);
    if (ps != kParseOk) return status_from_parse(ps);
    if (f.total_blocks() * 128 > coef_capacity_bytes) return HIPJPEG_STATUS_BUFFER_TOO_SMALL;
    int16_t* ptr[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t off = 0;
    for (int c = 0; c < f.ncomp; c++) {
        ptr[c] = coef + off;
        if (comp_offsets) comp_offsets[c] = off;
        off += (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h * 64;
        if (qtables)
            for (int j = 0; j < 64; j++) qtables[c * 64 + (j & 7) * 8 + (j >> 3)] = f.qtab[c][j];
    }
    switch (decode_coefficients(data, length, f, ptr)) {
    case kEntropyOk: return HIPJPEG_STATUS_SUCCESS;
    case kEntropyTruncated: return HIPJPEG_STATUS_TRUNCATED;
    case kEntropyMissingTable: return HIPJPEG_STATUS_BAD_JPEG;
    default: return HIPJPEG_STATUS_CORRUPT;
    }
    });
}

hipjpegStatus_t hipjpegEntropyDecodeHostSparse(const uint8_t* data, size_t length, uint8_t* stream, size_t capacity_bytes, size_t* stream_bytes,
                                               uint64_t table_offsets[4])
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!data || !stream || !stream_bytes) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    FrameInfo f;
    ParseStatus ps = parse_jpeg(data, length, &f);
    if (ps != kParseOk) return status_from_parse(ps);
    if (!sparse_staging_applies(f)) return HIPJPEG_STATUS_UNSUPPORTED;
    if (sparse_stream_capacity(f) > capacity_bytes) return HIPJPEG_STATUS_BUFFER_TOO_SMALL;
    size_t blocks = 0;
    for (int c = 0; c < f.ncomp; c++) {
        if (table_offsets) table_offsets[c] = blocks;
        blocks += (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h;
    }
    switch (decode_coefficients_sparse(data, length, f, stream, stream_bytes)) {
    case kEntropyOk: return HIPJPEG_STATUS_SUCCESS;
    case kEntropyTruncated: return HIPJPEG_STATUS_TRUNCATED;
    case kEntropyMissingTable: return HIPJPEG_STATUS_BAD_JPEG;
    default: return HIPJPEG_STATUS_CORRUPT;
    }
    });
}

hipjpegStatus_t hipjpegEntropyDecodeGpuAlgorithmHost(const uint8_t* data, size_t length, int16_t* coef, size_t coef_capacity_bytes,
                                                     uint64_t comp_offsets[4], int32_t* sync_passes)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!data || !coef) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    FrameInfo f;
    ParseStatus ps = parse_jpeg(data, length, &f);
    if (ps != kParseOk) return status_from_parse(ps);
    const bool progressive = gpu_progressive_eligible(f);
    if (!progressive && !gpu_entropy_eligible(f)) return HIPJPEG_STATUS_UNSUPPORTED;
    if (f.total_blocks() * 128 > coef_capacity_bytes) return HIPJPEG_STATUS_BUFFER_TOO_SMALL;
    int16_t* ptr[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t off = 0;
    for (int c = 0; c < f.ncomp; c++) {
        ptr[c] = coef + off;
        if (comp_offsets) comp_offsets[c] = off;
        off += (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h * 64;
    }
    int passes = 0;
    int rc = progressive ? emulate_gpu_progressive(data, length, f, ptr) : emulate_gpu_entropy(data, length, f, ptr, &passes);
    if (sync_passes) *sync_passes = passes;
    return rc == 0 ? HIPJPEG_STATUS_SUCCESS : (rc == 2 ? HIPJPEG_STATUS_TRUNCATED : HIPJPEG_STATUS_CORRUPT);
    });
}

hipjpegStatus_t hipjpegCreate(hipjpegHandle_t* handle, int device_id, int num_host_threads)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    *handle = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return HIPJPEG_STATUS_NO_DEVICE;
    if (device_id < 0 || device_id >= count) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    hipjpegHandle* h = new (std::nothrow) hipjpegHandle();
    if (!h) return HIPJPEG_STATUS_ALLOC_FAILED;
    h->device_id = device_id;
    h->pool.reset(new ForkJoinPool(num_host_threads));
    for (auto& b : h->batches) b.reset(new DecodeBatch(device_id, &h->hooks));
    h->encode.reset(new EncodeBatch(device_id, &h->hooks));
    h->encode_view = h->encode.get();
    *handle = h;
    return HIPJPEG_STATUS_SUCCESS;
    });
}

hipjpegStatus_t hipjpegDestroy(hipjpegHandle_t handle)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    (void)hipSetDevice(handle->device_id);
    if (handle->copy_stream) (void)hipStreamDestroy(handle->copy_stream);
    if (handle->entropy_stream) (void)hipStreamDestroy(handle->entropy_stream);
    for (hipStream_t ps : handle->page_entropy_stream)
        if (ps) (void)hipStreamDestroy(ps);
    for (auto& pg : handle->encode_pages) {
        if (pg.result.valid()) {
            try {
                (void)pg.result.get();
            } catch (...) {
            }
        }
        pg.batch.reset();
        if (pg.ready) (void)hipEventDestroy(pg.ready);
        if (pg.stream) (void)hipStreamDestroy(pg.stream);
    }
    delete handle;
    return HIPJPEG_STATUS_SUCCESS;
    });
}

hipjpegStatus_t hipjpegDecodeBatchHost(hipjpegHandle_t handle, const uint8_t* const* data, const size_t* lengths, int batch_size,
                                       const hipjpegOutput_t* outputs, hipjpegOutputFormat_t format, unsigned flags,
                                       hipjpegStatus_t* statuses)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    handle->current = (handle->current + 1) % handle->pages;
    DecodeBatch& b = handle->cur();
    static const bool timing = getenv("HIPJPEG_DEBUG_TIMING") != nullptr;  // debug aid: host-stage phase times on stderr
    const auto t0 = std::chrono::steady_clock::now();
    const bool geometry = !handle->transforms.empty();
    if (geometry && (int)handle->transforms.size() != batch_size) {
        handle->transforms.clear();
        return HIPJPEG_STATUS_INVALID_ARGUMENT;
    }
    hipjpegStatus_t st = b.plan(data, lengths, batch_size, outputs, format, flags, statuses, nullptr, handle->pool.get(),
                                geometry ? handle->transforms.data() : nullptr);
    handle->transforms.clear();
    if (st != HIPJPEG_STATUS_SUCCESS) return st;
    const auto t1 = std::chrono::steady_clock::now();
    handle->pool->parallel_for(batch_size, [&](int i, int) { b.entropy_stage(i); });
    const auto t2 = std::chrono::steady_clock::now();
    b.finalize(statuses);
    if (timing) {
        const auto t3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[hipjpeg] host stage: plan %.2f ms, entropy/staging %.2f ms, finalize %.2f ms (%d threads)\n", ms(t0, t1), ms(t1, t2), ms(t2, t3),
                handle->pool->num_threads());
    }
    return HIPJPEG_STATUS_SUCCESS;
    });
}

hipjpegStatus_t hipjpegDecodeBatchSetTransforms(hipjpegHandle_t handle, const hipjpegTransform_t* transforms, int batch_size)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle || batch_size < 0) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    if (transforms)
        handle->transforms.assign(transforms, transforms + batch_size);
    else
        handle->transforms.clear();
    return HIPJPEG_STATUS_SUCCESS;
    });
}

hipjpegStatus_t hipjpegDecodeBatchTransfer(hipjpegHandle_t handle, void* stream)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    return handle->cur().transfer(stream);
    });
}

hipjpegStatus_t hipjpegDecodeBatchDevice(hipjpegHandle_t handle, void* stream)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    return handle->cur().launch(stream);
    });
}

hipjpegStatus_t hipjpegDecodeBatchDeviceKernel(hipjpegHandle_t handle, int which, void* stream)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    return handle->cur().launch(stream, which);
    });
}

hipjpegStatus_t hipjpegDecodeBatchStats(hipjpegHandle_t handle, int32_t num_units[3], uint64_t* coef_bytes, uint64_t* output_bytes)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    handle->cur().stats(num_units, coef_bytes, output_bytes);
    return HIPJPEG_STATUS_SUCCESS;
    });
}

hipjpegStatus_t hipjpegDecodeBatch(hipjpegHandle_t handle, const uint8_t* const* data, const size_t* lengths, int batch_size,
                                   const hipjpegOutput_t* outputs, hipjpegOutputFormat_t format, unsigned flags, hipjpegStatus_t* statuses,
                                   void* stream)
{
    return guarded([&]() -> hipjpegStatus_t {
    hipjpegStatus_t st = hipjpegDecodeBatchHost(handle, data, lengths, batch_size, outputs, format, flags, statuses);
    if (st != HIPJPEG_STATUS_SUCCESS) return st;
    if ((st = hipjpegDecodeBatchTransfer(handle, stream)) != HIPJPEG_STATUS_SUCCESS) return st;
    st = hipjpegDecodeBatchDevice(handle, stream);
    // the GPU entropy stage may have found problems the host never looked at: wait for its verdicts (batches without
    // GPU-decoded streams return without waiting)
    if (st == HIPJPEG_STATUS_SUCCESS) st = handle->cur().resolve(stream);
    if (statuses)
        for (int i = 0; i < batch_size; i++) statuses[i] = handle->cur().image(i).status;
    return st;
    });
}

hipjpegStatus_t hipjpegDecodeBatchSubmit(hipjpegHandle_t handle, const uint8_t* const* data, const size_t* lengths, int batch_size,
                                         const hipjpegOutput_t* outputs, hipjpegOutputFormat_t format, unsigned flags, void* stream)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    if (handle->num_submitted >= handle->pages) return HIPJPEG_STATUS_INVALID_ARGUMENT;  // every page in flight: Wait first
    if (hipSetDevice(handle->device_id) != hipSuccess) return HIPJPEG_STATUS_NO_DEVICE;
    if (!handle->copy_stream && hipStreamCreateWithFlags(&handle->copy_stream, hipStreamNonBlocking) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
    static const bool two_streams = getenv("HIPJPEG_SINGLE_STREAM") == nullptr;  // HIPJPEG_SINGLE_STREAM=1: measurement aid
    if (two_streams && !handle->entropy_stream && hipStreamCreateWithFlags(&handle->entropy_stream, hipStreamNonBlocking) != hipSuccess)
        return HIPJPEG_STATUS_HIP_ERROR;
    hipjpegStatus_t st = hipjpegDecodeBatchHost(handle, data, lengths, batch_size, outputs, format, flags, nullptr);
    if (st != HIPJPEG_STATUS_SUCCESS) return st;
    DecodeBatch& b = handle->cur();
    // From here on the page may have device work queued (its H2D copy, entropy kernels, pixel kernels) that reads the page's pinned and
    // device arenas.  If a later step fails or throws, the page is not recorded as submitted and would be reused `pages` Submits later
    // without anybody having waited for that work: drain the streams it may sit on before the error leaves.
    hipStream_t es = handle->entropy_stream;
    auto drain = [&]() {
        if (hipSetDevice(handle->device_id) != hipSuccess) return;
        (void)hipStreamSynchronize(handle->copy_stream);
        if (handle->entropy_stream) (void)hipStreamSynchronize(handle->entropy_stream);
        if (es && es != handle->entropy_stream) (void)hipStreamSynchronize(es);
        (void)hipStreamSynchronize((hipStream_t)stream);
    };
    try {
        if ((st = b.transfer(handle->copy_stream, true)) == HIPJPEG_STATUS_SUCCESS) {
            if (two_streams && b.has_progressive()) {
                hipStream_t& ps = handle->page_entropy_stream[handle->current];
                if (!ps && hipStreamCreateWithFlags(&ps, hipStreamNonBlocking) != hipSuccess) st = HIPJPEG_STATUS_HIP_ERROR;
                es = ps;
            }
            if (st == HIPJPEG_STATUS_SUCCESS) st = b.launch(stream, -1, es);
        }
    } catch (...) {
        drain();
        throw;
    }
    if (st != HIPJPEG_STATUS_SUCCESS) {
        drain();
        return st;
    }
    handle->submitted[handle->num_submitted] = handle->current;
    handle->submitted_stream[handle->num_submitted] = stream;
    handle->num_submitted++;
    return HIPJPEG_STATUS_SUCCESS;
    });
}

hipjpegStatus_t hipjpegSetHybridHuffmanThreshold(hipjpegHandle_t handle, uint64_t pixels)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    for (auto& b : handle->batches)
        if (b) b->set_gpu_entropy_threshold(pixels);
    return HIPJPEG_STATUS_SUCCESS;
    });
}

hipjpegStatus_t hipjpegSetPipelineDepth(hipjpegHandle_t handle, int depth)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle || depth < 1 || depth > hipjpegHandle::kMaxPages || handle->num_submitted != 0) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    handle->pages = depth;
    handle->current = 0;
    return HIPJPEG_STATUS_SUCCESS;
    });
}

hipjpegStatus_t hipjpegDecodeBatchWait(hipjpegHandle_t handle, hipjpegStatus_t* statuses, int batch_size)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle || handle->num_submitted == 0) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    DecodeBatch& b = *handle->batches[handle->submitted[0]];
    void* stream = handle->submitted_stream[0];
    if (statuses && batch_size != b.size()) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    for (int k = 1; k < handle->num_submitted; k++) {
        handle->submitted[k - 1] = handle->submitted[k];
        handle->submitted_stream[k - 1] = handle->submitted_stream[k];
    }
    handle->num_submitted--;
    if (hipSetDevice(handle->device_id) != hipSuccess) return HIPJPEG_STATUS_NO_DEVICE;
    hipjpegStatus_t st = b.wait_done();
    if (st == HIPJPEG_STATUS_SUCCESS) st = b.resolve(stream);
    if (statuses)
        for (int i = 0; i < batch_size; i++) statuses[i] = b.image(i).status;
    return st;
    });
}

hipjpegStatus_t hipjpegDecodeBatchGetStatuses(hipjpegHandle_t handle, hipjpegStatus_t* statuses, int batch_size)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle || !statuses || batch_size != handle->cur().size()) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    hipjpegStatus_t st = handle->cur().resolve(handle->cur().last_stream());
    if (st != HIPJPEG_STATUS_SUCCESS) return st;
    for (int i = 0; i < batch_size; i++) statuses[i] = handle->cur().image(i).status;
    return HIPJPEG_STATUS_SUCCESS;
    });
}

hipjpegStatus_t hipjpegDecodeBatchEntropyStats(hipjpegHandle_t handle, int32_t* gpu_images, int32_t* sync_launches, uint64_t* stream_bytes)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    if (gpu_images) *gpu_images = handle->cur().gpu_entropy_images();
    if (sync_launches) *sync_launches = handle->cur().last_sync_launches();
    if (stream_bytes) *stream_bytes = handle->cur().stream_bytes();
    return HIPJPEG_STATUS_SUCCESS;
    });
}

int32_t hipjpegDecodeBatchZeroCopyImages(hipjpegHandle_t handle) { return handle ? handle->cur().zero_copy_images() : -1; }

hipjpegStatus_t hipjpegDecodeBatchTransferStats(hipjpegHandle_t handle, uint64_t* h2d_bytes, int32_t* sparse_images)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    if (h2d_bytes) *h2d_bytes = handle->cur().h2d_bytes();
    if (sparse_images) *sparse_images = handle->cur().sparse_images();
    return HIPJPEG_STATUS_SUCCESS;
    });
}

int32_t hipjpegTestScanChunkDrops(const uint8_t* data, size_t length, int scan_index, uint32_t* drops, int32_t capacity)
{
    int32_t result = -1;
    if (!test_hooks_enabled()) return result;
    (void)guarded([&]() -> hipjpegStatus_t {
        hipjpeg::FrameInfo f;
        if (hipjpeg::parse_jpeg(data, length, &f) != hipjpeg::kParseOk || scan_index < 0 || scan_index >= (int)f.scans.size()) return HIPJPEG_STATUS_BAD_JPEG;
        const auto& d = f.scans[scan_index].chunk_drops;
        for (int32_t c = 0; c < (int32_t)d.size() && c < capacity && drops; c++) drops[c] = d[c];
        result = (int32_t)d.size();
        return HIPJPEG_STATUS_SUCCESS;
    });
    return result;
}

int32_t hipjpegTestHostFallbacks(hipjpegHandle_t handle)
{
    return handle && test_hooks_enabled() ? handle->cur().host_fallback_images() : -1;
}

hipjpegStatus_t hipjpegTestKernelFlavours(hipjpegHandle_t handle, int32_t plane_units[1], int32_t luma_units[3])
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle || !plane_units || !luma_units || !test_hooks_enabled()) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    handle->cur().flavour_units(plane_units, luma_units);
    return HIPJPEG_STATUS_SUCCESS;
    });
}

int32_t hipjpegTestFusedUnits(hipjpegHandle_t handle) { return handle && test_hooks_enabled() ? handle->cur().fused_units() : -1; }

// ---------------------------------------------------------------- encode
hipjpegStatus_t hipjpegEncodeBatchDevice(hipjpegHandle_t handle, const hipjpegEncodeInput_t* inputs, const hipjpegEncodeParams_t* params,
                                         int batch_size, hipjpegStatus_t* statuses, void* stream)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    handle->encode_view = handle->encode.get();
    return handle->encode->device_stage(inputs, params, batch_size, statuses, stream);
    });
}

hipjpegStatus_t hipjpegEncodeBatchRelaunch(hipjpegHandle_t handle, void* stream)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    return handle->encode->relaunch(stream);
    });
}

hipjpegStatus_t hipjpegEncodeBatchEntropy(hipjpegHandle_t handle, unsigned flags, hipjpegStatus_t* statuses)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    EncodeBatch& b = *handle->encode;
    std::vector<char> todo(b.size(), 1);
    hipjpegStatus_t st;
    if (flags & HIPJPEG_FLAG_GPU_HUFFMAN) {
        if ((st = b.gpu_entropy_stage(&todo)) != HIPJPEG_STATUS_SUCCESS) return st;
    }
    bool any = false;
    for (int i = 0; i < b.size(); i++) any = any || (todo[i] && b.image(i).status == HIPJPEG_STATUS_SUCCESS);
    if (any) {
        // the host coder needs the coefficients on its side of PCIe
        if ((st = b.fetch_coefficients()) != HIPJPEG_STATUS_SUCCESS) return st;
        handle->pool->parallel_for(b.size(), [&](int i, int) {
            if (todo[i]) b.entropy_stage(i);
        });
    }
    if (statuses)
        for (int i = 0; i < b.size(); i++) statuses[i] = b.image(i).status;
    return HIPJPEG_STATUS_SUCCESS;
    });
}

hipjpegStatus_t hipjpegEncodeBatchHost(hipjpegHandle_t handle, hipjpegStatus_t* statuses) { return hipjpegEncodeBatchEntropy(handle, 0u, statuses); }

hipjpegStatus_t hipjpegEncodeBatch(hipjpegHandle_t handle, const hipjpegEncodeInput_t* inputs, const hipjpegEncodeParams_t* params,
                                   int batch_size, hipjpegStatus_t* statuses, void* stream)
{
    return guarded([&]() -> hipjpegStatus_t {
    hipjpegStatus_t st = hipjpegEncodeBatchDevice(handle, inputs, params, batch_size, statuses, stream);
    if (st != HIPJPEG_STATUS_SUCCESS) return st;
    return hipjpegEncodeBatchHost(handle, statuses);
    });
}

hipjpegStatus_t hipjpegEncodeBatchSubmit(hipjpegHandle_t handle, const hipjpegEncodeInput_t* inputs, const hipjpegEncodeParams_t* params, int batch_size,
                                         unsigned flags, void* stream)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle || batch_size < 0 || (batch_size > 0 && (!inputs || !params))) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    if (handle->encode_in_flight >= hipjpegHandle::kEncodePages) return HIPJPEG_STATUS_INVALID_ARGUMENT;  // every page busy: Wait first
    if (hipSetDevice(handle->device_id) != hipSuccess) return HIPJPEG_STATUS_NO_DEVICE;
    hipjpegHandle::EncodePage& pg = handle->encode_pages[handle->encode_next];
    if (!pg.batch) {
        pg.batch.reset(new EncodeBatch(handle->device_id, &handle->hooks));
        if (hipStreamCreateWithFlags(&pg.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&pg.ready, hipEventDisableTiming) != hipSuccess)
            return HIPJPEG_STATUS_HIP_ERROR;
    }
    pg.inputs.assign(inputs, inputs + batch_size);
    pg.params.assign(params, params + batch_size);
    // the page's stream must see the producer's pixels
    if (hipEventRecord(pg.ready, (hipStream_t)stream) != hipSuccess || hipStreamWaitEvent(pg.stream, pg.ready, 0) != hipSuccess)
        return HIPJPEG_STATUS_HIP_ERROR;
    const int device = handle->device_id;
    hipjpegHandle::EncodePage* page = &pg;
    pg.result = std::async(std::launch::async, [page, device, flags]() -> hipjpegStatus_t {
        if (hipSetDevice(device) != hipSuccess) return HIPJPEG_STATUS_NO_DEVICE;
        EncodeBatch& b = *page->batch;
        hipjpegStatus_t st = b.device_stage(page->inputs.data(), page->params.data(), (int)page->inputs.size(), nullptr, page->stream);
        if (st != HIPJPEG_STATUS_SUCCESS) return st;
        std::vector<char> todo(b.size(), 1);
        if ((flags & HIPJPEG_FLAG_GPU_HUFFMAN) && (st = b.gpu_entropy_stage(&todo)) != HIPJPEG_STATUS_SUCCESS) return st;
        bool any = false;
        for (int i = 0; i < b.size(); i++) any = any || (todo[i] && b.image(i).status == HIPJPEG_STATUS_SUCCESS);
        if (any) {
            if ((st = b.fetch_coefficients()) != HIPJPEG_STATUS_SUCCESS) return st;
            for (int i = 0; i < b.size(); i++)
                if (todo[i]) b.entropy_stage(i);
        }
        return HIPJPEG_STATUS_SUCCESS;
    });
    handle->encode_next = (handle->encode_next + 1) % hipjpegHandle::kEncodePages;
    handle->encode_in_flight++;
    return HIPJPEG_STATUS_SUCCESS;
    });
}

hipjpegStatus_t hipjpegEncodeBatchWait(hipjpegHandle_t handle, hipjpegStatus_t* statuses, int batch_size)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle || handle->encode_in_flight == 0) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    hipjpegHandle::EncodePage& pg = handle->encode_pages[handle->encode_oldest];
    handle->encode_oldest = (handle->encode_oldest + 1) % hipjpegHandle::kEncodePages;
    handle->encode_in_flight--;
    const hipjpegStatus_t st = pg.result.get();
    handle->encode_view = pg.batch.get();
    if (statuses) {
        if (batch_size != pg.batch->size()) return HIPJPEG_STATUS_INVALID_ARGUMENT;
        for (int i = 0; i < batch_size; i++) statuses[i] = pg.batch->image(i).status;
    }
    return st;
    });
}

hipjpegStatus_t hipjpegEncodeGetBitstream(hipjpegHandle_t handle, int index, const uint8_t** data, size_t* length)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle || !data || !length || index < 0 || index >= handle->encode_view->size()) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    PlannedEncode& im = handle->encode_view->image(index);
    if (im.status != HIPJPEG_STATUS_SUCCESS) return im.status;
    *data = im.file();
    *length = im.file_size();
    return HIPJPEG_STATUS_SUCCESS;
    });
}

hipjpegStatus_t hipjpegEncodeGetCoefficients(hipjpegHandle_t handle, int index, int component, const int16_t** coef, int32_t grid[4])
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle || !coef || !grid || index < 0 || index >= handle->encode_view->size()) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    PlannedEncode& im = handle->encode_view->image(index);
    if (im.status != HIPJPEG_STATUS_SUCCESS) return im.status;
    if (component < 0 || component >= im.geom.ncomp) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    hipjpegStatus_t st = handle->encode_view->fetch_coefficients();
    if (st != HIPJPEG_STATUS_SUCCESS) return st;
    *coef = handle->encode_view->host_coef(index, component);
    grid[0] = im.geom.blocks_w[component];
    grid[1] = im.geom.blocks_h[component];
    grid[2] = im.geom.real_w[component];
    grid[3] = im.geom.real_h[component];
    return HIPJPEG_STATUS_SUCCESS;
    });
}

hipjpegStatus_t hipjpegEncodeBatchStats(hipjpegHandle_t handle, int32_t* num_units, uint64_t* pixel_bytes, uint64_t* coef_bytes)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!handle) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    if (num_units) *num_units = handle->encode_view->num_units();
    if (pixel_bytes) *pixel_bytes = handle->encode_view->pixel_bytes();
    if (coef_bytes) *coef_bytes = handle->encode_view->coef_bytes();
    return HIPJPEG_STATUS_SUCCESS;
    });
}

int32_t hipjpegEncodeBatchGpuEntropyImages(hipjpegHandle_t handle)
{
    return handle && handle->encode_view ? (int32_t)handle->encode_view->gpu_entropy_images() : -1;
}

hipjpegStatus_t hipjpegEncodeFromCoefficientsHost(int32_t width, int32_t height, const hipjpegEncodeParams_t* params, const int16_t* const coef[3],
                                                  uint8_t* out, size_t capacity, size_t* length)
{
    return guarded([&]() -> hipjpegStatus_t {
    if (!params || !coef || !length || width < 1 || height < 1 || width > 65535 || height > 65535) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    EncodeGeometry g;
    g.width = width;
    g.height = height;
    hipjpegStatus_t st = subsampling_factors(params->subsampling, &g.ncomp, &g.hs, &g.vs);
    if (st != HIPJPEG_STATUS_SUCCESS) return st;
    compute_geometry(&g);
    for (int c = 0; c < g.ncomp; c++)
        if (!coef[c]) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    uint16_t ql[64], qc[64];
    quality_tables(params->quality, ql, qc);
    EntropyEncodeOptions opt;
    opt.restart_interval = params->restart_interval;
    opt.optimized_huffman = params->optimized_huffman != 0;
    opt.progressive = params->progressive != 0;
    std::vector<uint8_t> bytes;
    encode_jfif(g, ql, qc, coef, opt, &bytes);
    *length = bytes.size();
    if (!out || capacity < bytes.size()) return HIPJPEG_STATUS_BUFFER_TOO_SMALL;
    memcpy(out, bytes.data(), bytes.size());
    return HIPJPEG_STATUS_SUCCESS;
    });
}

}  // extern "C"
