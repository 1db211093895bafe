// encoder_core.cpp -- see encoder_core.h
#include "encoder_core.h"

#include <hip/hip_runtime_api.h>

#include <cstring>

#include "encode_kernels.h"
#include "jpeg_syntax.h"

namespace hipjpeg {

namespace {
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr int kTileBX = 32, kTileBY = 8;
}  // namespace

hipjpegStatus_t subsampling_factors(int subsampling, int* ncomp, int* hs, int* vs)
{
    *ncomp = 3;
    switch (subsampling) {
    case HIPJPEG_CSS_444: *hs = 1; *vs = 1; break;
    case HIPJPEG_CSS_422: *hs = 2; *vs = 1; break;
    case HIPJPEG_CSS_420: *hs = 2; *vs = 2; break;
    case HIPJPEG_CSS_440: *hs = 1; *vs = 2; break;
    case HIPJPEG_CSS_411: *hs = 4; *vs = 1; break;
    case HIPJPEG_CSS_410: *hs = 4; *vs = 2; break;
    case HIPJPEG_CSS_GRAY: *ncomp = 1; *hs = 1; *vs = 1; break;
    default: return HIPJPEG_STATUS_UNSUPPORTED;
    }
    return HIPJPEG_STATUS_SUCCESS;
}

EncodeBatch::EncodeBatch(int device_id, const MemoryHooks* hooks)
    : device_id_(device_id), pinned_desc_(Buffer::kPinned, hooks), device_(Buffer::kDevice, hooks), pinned_coef_(Buffer::kPinned, hooks)
{
}

EncodeBatch::~EncodeBatch()
{
    if (event_) {
        if (launched_) (void)hipEventSynchronize((hipEvent_t)event_);
        (void)hipEventDestroy((hipEvent_t)event_);
    }
}

const int16_t* EncodeBatch::host_coef(int i, int c) const
{
    return reinterpret_cast<const int16_t*>(pinned_coef_.data() + images_[i].coef_offset[c]);
}

hipjpegStatus_t EncodeBatch::device_stage(const hipjpegEncodeInput_t* inputs, const hipjpegEncodeParams_t* params, int n,
                                          hipjpegStatus_t* statuses, void* stream)
{
    if (n < 0 || (n > 0 && (!inputs || !params))) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    if (hipSetDevice(device_id_) != hipSuccess) return HIPJPEG_STATUS_NO_DEVICE;
    if (launched_ && event_) (void)hipEventSynchronize((hipEvent_t)event_);  // previous use of the buffers must have drained
    launched_ = fetched_ = false;
    images_.assign(n, PlannedEncode());
    desc_.assign(n, EncodeImage());
    units_.clear();
    coef_total_ = 0;
    pixel_bytes_ = coef_bytes_ = 0;
    for (int i = 0; i < n; i++) {
        PlannedEncode& im = images_[i];
        im.params = params[i];
        const hipjpegEncodeInput_t& in = inputs[i];
        EncodeGeometry& g = im.geom;
        g.width = in.width;
        g.height = in.height;
        im.status = subsampling_factors(params[i].subsampling, &g.ncomp, &g.hs, &g.vs);
        const int fmt = params[i].input_format;
        if (im.status == HIPJPEG_STATUS_SUCCESS) {
            if (in.width < 1 || in.height < 1 || in.width > 65535 || in.height > 65535) im.status = HIPJPEG_STATUS_INVALID_ARGUMENT;
            if (fmt != HIPJPEG_OUTPUT_RGBI && fmt != HIPJPEG_OUTPUT_BGRI && fmt != HIPJPEG_OUTPUT_RGB_PLANAR && fmt != HIPJPEG_OUTPUT_BGR_PLANAR &&
                fmt != HIPJPEG_OUTPUT_Y)
                im.status = HIPJPEG_STATUS_UNSUPPORTED;
            if (fmt == HIPJPEG_OUTPUT_Y && g.ncomp != 1) im.status = HIPJPEG_STATUS_UNSUPPORTED;  // gray pixels carry no chroma
            const int nplanes = (fmt == HIPJPEG_OUTPUT_RGB_PLANAR || fmt == HIPJPEG_OUTPUT_BGR_PLANAR) ? 3 : 1;
            for (int p = 0; p < nplanes; p++)
                if (!in.plane[p]) im.status = HIPJPEG_STATUS_INVALID_ARGUMENT;
            if (params[i].restart_interval < 0 || params[i].restart_interval > 65535) im.status = HIPJPEG_STATUS_INVALID_ARGUMENT;
        }
        if (im.status != HIPJPEG_STATUS_SUCCESS) continue;
        compute_geometry(&g);
        quality_tables(params[i].quality, im.qlum, im.qchr);
        EncodeImage& d = desc_[i];
        memset(&d, 0, sizeof d);
        d.width = (uint32_t)g.width;
        d.height = (uint32_t)g.height;
        d.ncomp = (uint32_t)g.ncomp;
        d.hs = (uint32_t)g.hs;
        d.vs = (uint32_t)g.vs;
        d.in_format = fmt == HIPJPEG_OUTPUT_Y ? (uint32_t)kInGray : (uint32_t)fmt;  // RGBI/BGRI/planar values coincide with InFormat
        for (int p = 0; p < 3; p++) {
            d.in[p] = static_cast<const uint8_t*>(in.plane[p]);
            d.in_pitch[p] = in.pitch[p];
        }
        for (int c = 0; c < g.ncomp; c++) {
            d.blocks_w[c] = (uint32_t)g.blocks_w[c];
            d.blocks_h[c] = (uint32_t)g.blocks_h[c];
            d.real_w[c] = (uint32_t)g.real_w[c];
            d.real_h[c] = (uint32_t)g.real_h[c];
            im.coef_offset[c] = coef_total_;
            coef_total_ += (size_t)g.blocks_w[c] * g.blocks_h[c] * 128;
        }
        for (int t = 0; t < 2; t++) {
            const uint16_t* q = t ? im.qchr : im.qlum;
            for (int k = 0; k < 64; k++) {
                const uint32_t div = 8u * q[kZigzagNatural[k]];
                d.quant[t].magic[k] = (1u << 28) / div + 1;
                d.quant[t].half[k] = div >> 1;
            }
        }
        // tiles cover the real luma blocks only
        const int tiles_x = (g.real_w[0] + kTileBX - 1) / kTileBX, tiles_y = (g.real_h[0] + kTileBY - 1) / kTileBY;
        for (int ty = 0; ty < tiles_y; ty++)
            for (int tx = 0; tx < tiles_x; tx++) units_.push_back(EncodeUnit{(uint32_t)i, (uint32_t)tx, (uint32_t)ty, 0u});
        pixel_bytes_ += (uint64_t)g.width * g.height * (g.ncomp == 1 ? 1 : 3);
        for (int c = 0; c < g.ncomp; c++) coef_bytes_ += (uint64_t)g.real_w[c] * g.real_h[c] * 128;
    }
    units_offset_ = align_up(sizeof(EncodeImage) * (size_t)n, 256);
    desc_bytes_ = align_up(units_offset_ + sizeof(EncodeUnit) * units_.size(), 256);
    coef_offset_ = desc_bytes_;
    hipjpegStatus_t st;
    if ((st = pinned_desc_.reserve(desc_bytes_ + 256)) != HIPJPEG_STATUS_SUCCESS) return st;
    if ((st = device_.reserve(desc_bytes_ + coef_total_ + 256)) != HIPJPEG_STATUS_SUCCESS) return st;
    if ((st = pinned_coef_.reserve(coef_total_ + 256)) != HIPJPEG_STATUS_SUCCESS) return st;
    for (int i = 0; i < n; i++) {
        if (images_[i].status != HIPJPEG_STATUS_SUCCESS) continue;
        for (int c = 0; c < images_[i].geom.ncomp; c++)
            desc_[i].coef[c] = reinterpret_cast<int16_t*>(device_.data() + coef_offset_ + images_[i].coef_offset[c]);
    }
    if (n) memcpy(pinned_desc_.data(), desc_.data(), sizeof(EncodeImage) * (size_t)n);
    if (!units_.empty()) memcpy(pinned_desc_.data() + units_offset_, units_.data(), sizeof(EncodeUnit) * units_.size());
    if (statuses)
        for (int i = 0; i < n; i++) statuses[i] = images_[i].status;
    stream_ = stream;
    if (desc_bytes_ && hipMemcpyAsync(device_.data(), pinned_desc_.data(), desc_bytes_, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
        return HIPJPEG_STATUS_HIP_ERROR;
    return relaunch(stream);
}

hipjpegStatus_t EncodeBatch::relaunch(void* stream)
{
    int rc = launch_forward(reinterpret_cast<const EncodeImage*>(device_.data()), reinterpret_cast<const EncodeUnit*>(device_.data() + units_offset_),
                            (int)units_.size(), stream);
    if (rc != 0) return HIPJPEG_STATUS_HIP_ERROR;
    if (!event_) {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
        event_ = ev;
    }
    if (hipEventRecord((hipEvent_t)event_, (hipStream_t)stream) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
    launched_ = true;
    fetched_ = false;
    stream_ = stream;
    return HIPJPEG_STATUS_SUCCESS;
}

hipjpegStatus_t EncodeBatch::fetch_coefficients()
{
    if (!launched_) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    if (fetched_) return HIPJPEG_STATUS_SUCCESS;
    if (hipSetDevice(device_id_) != hipSuccess) return HIPJPEG_STATUS_NO_DEVICE;
    if (coef_total_ &&
        hipMemcpyAsync(pinned_coef_.data(), device_.data() + coef_offset_, coef_total_, hipMemcpyDeviceToHost, (hipStream_t)stream_) != hipSuccess)
        return HIPJPEG_STATUS_HIP_ERROR;
    if (hipEventRecord((hipEvent_t)event_, (hipStream_t)stream_) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
    if (hipEventSynchronize((hipEvent_t)event_) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
    fetched_ = true;
    return HIPJPEG_STATUS_SUCCESS;
}

void EncodeBatch::entropy_stage(int i)
{
    PlannedEncode& im = images_[i];
    if (im.status != HIPJPEG_STATUS_SUCCESS) return;
    const int16_t* coef[3] = {nullptr, nullptr, nullptr};
    for (int c = 0; c < im.geom.ncomp; c++) coef[c] = host_coef(i, c);
    EntropyEncodeOptions opt;
    opt.restart_interval = im.params.restart_interval;
    opt.optimized_huffman = im.params.optimized_huffman != 0;
    im.bitstream.clear();
    encode_jfif(im.geom, im.qlum, im.qchr, coef, opt, &im.bitstream);
}

}  // namespace hipjpeg
