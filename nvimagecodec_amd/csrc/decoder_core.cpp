// decoder_core.cpp -- see decoder_core.h
#include "decoder_core.h"
#include "host_copy.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "decode_kernels.h"
#include "diagnostics.h"
#include "entropy_decode.h"
#include "gpu_huffman_host.h"
#include "progressive_gpu_host.h"

namespace hipjpeg {

namespace {
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
}  // namespace

hipjpegStatus_t status_from_parse(ParseStatus s)
{
    switch (s) {
    case kParseOk: return HIPJPEG_STATUS_SUCCESS;
    case kParseUnsupported: return HIPJPEG_STATUS_UNSUPPORTED;
    case kParseTruncated: return HIPJPEG_STATUS_TRUNCATED;
    default: return HIPJPEG_STATUS_BAD_JPEG;
    }
}

// Same classification the framework's parser applies (reference src/parsers/jpeg.cpp:70-114).
// ---------------------------------------------------------------- Buffer
hipjpegStatus_t Buffer::reserve(size_t bytes)
{
    if (bytes <= cap_) return HIPJPEG_STATUS_SUCCESS;
    release();
    size_t want = align_up(bytes + bytes / 8, 1 << 20);  // headroom so a slightly bigger next batch does not reallocate
    void* p = nullptr;
    if (kind_ == kDevice && hooks_ && hooks_->device_malloc) {
        if (hooks_->device_malloc(hooks_->device_ctx, &p, want, nullptr) != 0 || !p) return HIPJPEG_STATUS_ALLOC_FAILED;
        custom_ = true;
    } else if (kind_ == kPinned && hooks_ && hooks_->pinned_malloc) {
        if (hooks_->pinned_malloc(hooks_->pinned_ctx, &p, want, nullptr) != 0 || !p) return HIPJPEG_STATUS_ALLOC_FAILED;
        custom_ = true;
    } else {
        hipError_t e = kind_ == kDevice ? hipMalloc(&p, want) : hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e != hipSuccess) return HIPJPEG_STATUS_ALLOC_FAILED;
        custom_ = false;
    }
    ptr_ = static_cast<uint8_t*>(p);
    cap_ = want;
    return HIPJPEG_STATUS_SUCCESS;
}

void Buffer::release()
{
    if (!ptr_) return;
    if (custom_) {
        if (kind_ == kDevice)
            hooks_->device_free(hooks_->device_ctx, ptr_, cap_, nullptr);
        else
            hooks_->pinned_free(hooks_->pinned_ctx, ptr_, cap_, nullptr);
    } else if (kind_ == kDevice) {
        (void)hipFree(ptr_);
    } else {
        (void)hipHostFree(ptr_);
    }
    ptr_ = nullptr;
    cap_ = 0;
}

// ---------------------------------------------------------------- DecodeBatch
DecodeBatch::DecodeBatch(int device_id, const MemoryHooks* hooks)
    : device_id_(device_id), pinned_(Buffer::kPinned, hooks), device_(Buffer::kDevice, hooks), planes_(Buffer::kDevice, hooks),
      work_(Buffer::kDevice, hooks)
{
}

DecodeBatch::~DecodeBatch()
{
    if (taken_units_dev_) (void)hipFree(taken_units_dev_);
    if (copied_event_) (void)hipEventDestroy((hipEvent_t)copied_event_);
    if (entropy_event_) (void)hipEventDestroy((hipEvent_t)entropy_event_);
    if (done_event_) {
        if (in_flight_) (void)hipEventSynchronize((hipEvent_t)done_event_);
        (void)hipEventDestroy((hipEvent_t)done_event_);
    }
}

namespace {

// Decide which kernels handle this frame.  Returns false when the layout is outside the decoder's scope.
bool choose_variant(const FrameInfo& f, OutFormat fmt, bool fancy, int* variant)
{
    const bool four = f.color == ColorModel::CMYK || f.color == ColorModel::YCCK;
    if (four ? f.ncomp != 4 : (f.ncomp != 1 && f.ncomp != 3)) return false;
    for (int c = 0; c < f.ncomp; c++)
        if (f.hmax % f.comp[c].h || f.vmax % f.comp[c].v) return false;  // fractional upsampling: libjpeg refuses too
    if (four) {
        // CMYK / YCCK: all four components through planes, then cmyk_color_kernel -- the formats the reference's CPU path
        // produces from such files (extensions/libjpeg_turbo/jpeg_mem.cpp:318-337: I_RGB, I_BGR, P_Y) and their planar twins
        if (fmt == kOutPlanarYUV) return false;
        *variant = -3;
        return true;
    }
    if (fmt == kOutPlanarYUV) {
        *variant = -2;
        return true;
    }
    if (fmt == kOutY) {
        // libjpeg JCS_GRAYSCALE from YCbCr/gray = the luma plane; from an RGB-model JPEG it is a weighted sum (not done here)
        if (f.color == ColorModel::RGB) return false;
        if (f.comp[0].h != f.hmax || f.comp[0].v != f.vmax) return false;
        *variant = -2;
        return true;
    }
    if (f.ncomp == 1) {
        *variant = kVarGray;
        return true;
    }
    const int fx1 = f.hmax / f.comp[1].h, fy1 = f.vmax / f.comp[1].v, fx2 = f.hmax / f.comp[2].h, fy2 = f.vmax / f.comp[2].v;
    const bool luma_full = f.comp[0].h == f.hmax && f.comp[0].v == f.vmax;
    if (luma_full && fx1 == fx2 && fy1 == fy2 && fx1 <= 2 && fy1 <= 2) {
        *variant = fx1 == 1 ? (fy1 == 1 ? kVar11 : kVar12) : (fy1 == 1 ? kVar21 : kVar22);
        return true;
    }
    // Generic path = replication only.  Make sure libjpeg would replicate as well (jdsample.c jinit_upsampler).
    for (int c = 0; c < f.ncomp; c++) {
        int fx = f.hmax / f.comp[c].h, fy = f.vmax / f.comp[c].v;
        bool triangle = fancy && ((fx == 2 && fy == 1 && f.comp[c].samp_w > 2) || (fx == 2 && fy == 2 && f.comp[c].samp_w > 2) ||
                                  (fx == 1 && fy == 2));
        if (triangle) return false;
    }
    *variant = -1;
    return true;
}

}  // namespace

// Largest frame the decoder takes: the reference's CPU path refuses width x height x components >= 2^29 before it allocates
// anything (extensions/libjpeg_turbo/jpeg_mem.cpp:183-196), and a forged SOF of a few hundred bytes must not be able to
// reserve tens of GB of pinned memory for the whole batch.  HIPJPEG_MAX_IMAGE_SAMPLES overrides the bound (tests).
static uint64_t max_image_samples()
{
    static const uint64_t v = [] {
        const char* e = getenv("HIPJPEG_MAX_IMAGE_SAMPLES");
        const unsigned long long x = e ? strtoull(e, nullptr, 10) : 0ull;
        return x ? (uint64_t)x : (uint64_t)1 << 29;
    }();
    return v;
}

hipjpegStatus_t DecodeBatch::plan(const uint8_t* const* data, const size_t* lengths, int n, const hipjpegOutput_t* outputs,
                                  hipjpegOutputFormat_t format, unsigned flags, hipjpegStatus_t* statuses,
                                  const hipjpegOutputFormat_t* formats, ForkJoinPool* pool, const hipjpegTransform_t* transforms)
{
    // An allocation that fails costs only the images that made the arenas grow: the biggest remaining image is given up
    // (ALLOC_FAILED) and the layout is computed again without it.
    std::vector<char> give_up((size_t)std::max(n, 0), 0);
    pool_ = pool;
    for (int attempt = 0;; attempt++) {
        const hipjpegStatus_t st = plan_once(data, lengths, n, outputs, format, flags, statuses, formats, pool, transforms, give_up);
        if (st != HIPJPEG_STATUS_ALLOC_FAILED || attempt >= 16) return st;
        int worst = -1;
        size_t worst_bytes = 0;
        for (int i = 0; i < n; i++) {
            if (images_[i].status != HIPJPEG_STATUS_SUCCESS) continue;
            const size_t b = images_[i].frame.total_blocks() * 128 + (size_t)images_[i].frame.width * images_[i].frame.height;
            if (b >= worst_bytes) {
                worst_bytes = b;
                worst = i;
            }
        }
        if (worst < 0) return st;
        give_up[worst] = 1;
    }
}

hipjpegStatus_t DecodeBatch::plan_once(const uint8_t* const* data, const size_t* lengths, int n, const hipjpegOutput_t* outputs,
                                       hipjpegOutputFormat_t format, unsigned flags, hipjpegStatus_t* statuses,
                                       const hipjpegOutputFormat_t* formats, ForkJoinPool* pool, const hipjpegTransform_t* transforms,
                                       const std::vector<char>& give_up)
{
    if (n < 0 || (n > 0 && (!data || !lengths || !outputs))) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    if ((int)format < 0 || (int)format > (int)HIPJPEG_OUTPUT_YUV_PLANAR) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    ScopedRange range("hipjpeg plan (parse headers, lay out staging)");
    fault_point("plan");
    finalized_ = false;
    host_drops_missing_.store(false, std::memory_order_relaxed);
    // (every element is reset by its own prepare() below -- on the pool: a batch of a thousand small pictures spent a third of its
    // plan() clearing these two arrays on the caller's thread)
    images_.resize((size_t)n);
    desc_.resize((size_t)n);
    const bool fancy = (flags & HIPJPEG_FLAG_FANCY_UPSAMPLING) != 0;
    const bool want_gpu_entropy = (flags & HIPJPEG_FLAG_GPU_HUFFMAN) != 0;
    entropy_done_ = false;
    entropy_pending_ = false;  // a caller that re-plans without resolve() gives up the statuses of the previous batch
    huff_images_.clear();
    huff_to_image_.clear();
    prog_images_.clear();
    prog_to_image_.clear();
    prog_scan_total_ = 0;
    prog_slot_words_ = 0;
    max_prog_units_ = 0;

    size_t max_units = 0, coef_total = 0, plane_total = 0, max_xform_units = 0;
    std::vector<size_t> xform_plane_off;  // per TransformImage plane, in order
    xform_desc_.clear();
    size_t huff_stream_total = 0, huff_subseq_ub = 0, huff_pool_total = 0, huff_blocks_total = 0, huff_raw_total = 0, huff_chunks_total = 0;
    size_t huff_blockpos_total = 0, huff_boundary_total = 0, prog_pos_total = 0;
    max_huff_units_ = max_huff_wunits_ = 0;
    max_pool_words_ = 0;
    std::vector<size_t> plane_off((size_t)n * 4, (size_t)-1);
    coef_bytes_ = output_bytes_ = 0;
    // per-image header work (marker walk through the whole file, eligibility, table sizing) is independent: all cores
    auto prepare = [&](int i) {
        PlannedImage& im = images_[i];
        im = PlannedImage();
        desc_[i] = DecodeImage();
        im.data = data[i];
        im.size = lengths[i];
        if (formats && ((int)formats[i] < 0 || (int)formats[i] > (int)HIPJPEG_OUTPUT_YUV_PLANAR)) {
            im.status = HIPJPEG_STATUS_INVALID_ARGUMENT;
            return;
        }
        const OutFormat fmt = (OutFormat)(formats ? formats[i] : format);
        im.status = data[i] ? status_from_parse(parse_jpeg(data[i], lengths[i], &im.frame)) : HIPJPEG_STATUS_INVALID_ARGUMENT;
        if (im.status == HIPJPEG_STATUS_SUCCESS &&
            ((uint64_t)im.frame.width * (uint64_t)im.frame.height * (uint64_t)im.frame.ncomp >= max_image_samples() || give_up[i]))
            im.status = HIPJPEG_STATUS_ALLOC_FAILED;  // this image only; its neighbours decode
        if (im.status == HIPJPEG_STATUS_SUCCESS && !choose_variant(im.frame, fmt, fancy, &im.variant)) im.status = HIPJPEG_STATUS_UNSUPPORTED;
        const FrameInfo& f = im.frame;
        const int nplanes_out = (fmt == kOutInterleavedRGB || fmt == kOutInterleavedBGR || fmt == kOutY) ? 1 : 3;
        if (im.status == HIPJPEG_STATUS_SUCCESS) {
            for (int p = 0; p < (fmt == kOutPlanarYUV ? f.ncomp : nplanes_out); p++)
                if (!outputs[i].plane[p]) im.status = HIPJPEG_STATUS_INVALID_ARGUMENT;
        }
        if (im.status == HIPJPEG_STATUS_SUCCESS && transforms) {
            // region of interest / EXIF orientation: normalise, validate (crop semantics of the reference's CPU path,
            // extensions/libjpeg_turbo/libjpeg_turbo_decoder.cpp:355-370: the region must lie inside the image)
            hipjpegTransform_t t = transforms[i];
            if (t.x0 == 0 && t.y0 == 0 && t.x1 == 0 && t.y1 == 0) {
                t.x1 = f.width;
                t.y1 = f.height;
            }
            if (t.orientation == 0) t.orientation = 1;
            const bool whole = t.x0 == 0 && t.y0 == 0 && t.x1 == f.width && t.y1 == f.height;
            if (t.orientation < 1 || t.orientation > 8 || t.x0 < 0 || t.y0 < 0 || t.x1 > f.width || t.y1 > f.height || t.x1 <= t.x0 || t.y1 <= t.y0)
                im.status = HIPJPEG_STATUS_INVALID_ARGUMENT;
            else if (!(whole && t.orientation == 1)) {
                if (fmt == kOutPlanarYUV)
                    im.status = HIPJPEG_STATUS_UNSUPPORTED;  // geometry on subsampled planes is not defined
                else {
                    im.has_transform = true;
                    im.transform = t;
                }
            }
        }
        if (im.status == HIPJPEG_STATUS_SUCCESS) {
            // rows of the caller's buffer must hold a row of what is written there (nvJPEG checks the pitch for the reference;
            // here a short pitch would make the kernels write overlapping rows or leave the buffer)
            int ow = f.width;
            if (im.has_transform) {
                const int rw = im.transform.x1 - im.transform.x0, rh = im.transform.y1 - im.transform.y0;
                ow = im.transform.orientation >= 5 ? rh : rw;
            }
            if (fmt == kOutPlanarYUV) {
                for (int c = 0; c < f.ncomp; c++)
                    if (outputs[i].pitch[c] < (uint32_t)f.comp[c].samp_w) im.status = HIPJPEG_STATUS_INVALID_ARGUMENT;
            } else {
                const uint32_t bpp = (fmt == kOutInterleavedRGB || fmt == kOutInterleavedBGR) ? 3u : 1u;
                for (int p = 0; p < nplanes_out; p++)
                    if ((uint64_t)outputs[i].pitch[p] < (uint64_t)ow * bpp) im.status = HIPJPEG_STATUS_INVALID_ARGUMENT;
            }
        }
        const bool big_enough = (uint64_t)f.width * (uint64_t)f.height > gpu_entropy_min_pixels_ || gpu_entropy_min_pixels_ == 0;
        if (im.status == HIPJPEG_STATUS_SUCCESS && want_gpu_entropy && big_enough && gpu_entropy_eligible(f)) {
            im.gpu_entropy = true;
            im.pool_words = gpu_pool_words(f.scans[0]);
            // Zero-copy input: page-locked caller memory (hipHostMalloc / hipHostRegister, a pinned torch tensor) can
            // be read by the copy engine directly, so the scan's bytes skip the staging copy -- per rank one pass over host DRAM instead
            // of three (read, non-temporal store into the staging area, the engine's read).  HIPJPEG_NO_ZERO_COPY=1 switches it off.
            static const bool zero_copy = getenv("HIPJPEG_NO_ZERO_COPY") == nullptr;
            if (zero_copy) {
                hipPointerAttribute_t attr;
                if (hipPointerGetAttributes(&attr, im.data) == hipSuccess) {
                    im.input_pinned = attr.type == hipMemoryTypeHost && attr.devicePointer != nullptr;
                    im.input_device_view = static_cast<const uint8_t*>(attr.devicePointer);  // the same bytes as the device addresses them
                } else
                    (void)hipGetLastError();  // ordinary pageable memory: the query fails and leaves an error behind
            }
        } else if (im.status == HIPJPEG_STATUS_SUCCESS && want_gpu_entropy && big_enough && gpu_progressive_eligible(f)) {
            im.gpu_entropy = im.gpu_prog = true;
            im.pool_words = prog_pool_words(f);
        }
        if (im.status != HIPJPEG_STATUS_SUCCESS) return;
        // the part of the device descriptor that does not depend on where things will lie in the arenas
        DecodeImage& d = desc_[i];
        d.width = (uint32_t)f.width;
        d.height = (uint32_t)f.height;
        d.ncomp = (uint32_t)f.ncomp;
        d.hmax = (uint32_t)f.hmax;
        d.vmax = (uint32_t)f.vmax;
        d.color_model = (uint32_t)f.color;
        d.out_format = (uint32_t)fmt;
        d.flags = (fancy ? kFlagFancyUpsampling : 0) | (f.saw_adobe ? kFlagAdobeMarker : 0);
        for (int p = 0; p < 3; p++) {
            d.out[p] = static_cast<uint8_t*>(outputs[i].plane[p]);
            d.out_pitch[p] = outputs[i].pitch[p];
        }
        for (int c = 0; c < f.ncomp; c++) {
            const Component& k = f.comp[c];
            DecodeComponent& dc = d.comp[c];
            dc.blocks_w = (uint16_t)k.blocks_w;
            dc.blocks_h = (uint16_t)k.blocks_h;
            dc.samp_w = (uint16_t)k.samp_w;
            dc.samp_h = (uint16_t)k.samp_h;
            dc.h = (uint16_t)k.h;
            dc.v = (uint16_t)k.v;
            for (int pp = 0; pp < 2; pp++)
                for (int j = 0; j < 4; j++)
                    for (int r = 0; r < 8; r++) {
                        const uint32_t q16 = (uint32_t)f.qtab[c][r * 8 + 4 * pp + j] & 0xFFFFu;
                        uint32_t& pk = dc.qpk[pp][j * 4 + (r >> 1)];
                        pk = (r & 1) ? (pk | (q16 << 16)) : q16;
                    }
        }
    };
    if (pool && n > 1)
        pool->parallel_for(n, [&](int i, int) { prepare(i); });
    else
        for (int i = 0; i < n; i++) prepare(i);

    // The progressive walk / replay launches are sized for the batch's maxima (largest table x most scans, enqueue_progressive): keep
    // their dynamic LDS inside the device's workgroup limit by handing the most demanding images to the host entropy stage -- a refused
    // launch would fail the whole batch.  No golden and no libjpeg-written file comes near the limit.
    {
        static const size_t lds_limit = [] {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, 0) != hipSuccess || v <= 0) v = 64 * 1024;
            if (const char* e = getenv("HIPJPEG_PROG_LDS_LIMIT")) v = atoi(e);  // test aid
            return (size_t)v;
        }();
        for (;;) {
            ProgLdsShape all;
            int worst = -1;
            size_t worst_need = 0;
            for (int i = 0; i < n; i++) {
                if (!images_[i].gpu_prog) continue;
                const ProgLdsShape sh = prog_lds_shape(images_[i].frame);
                all.merge(sh);
                const size_t need = std::max(prog_walk_lds_bytes(sh), prog_replay_lds_bytes(sh));
                if (need > worst_need) {
                    worst_need = need;
                    worst = i;
                }
            }
            if (worst < 0 || std::max(prog_walk_lds_bytes(all), prog_replay_lds_bytes(all)) <= lds_limit) break;
            images_[worst].gpu_entropy = images_[worst].gpu_prog = false;
            images_[worst].pool_words = 0;
        }
    }

    for (int i = 0; i < n; i++) {
        PlannedImage& im = images_[i];
        const OutFormat fmt = (OutFormat)(formats && im.status != HIPJPEG_STATUS_INVALID_ARGUMENT ? formats[i] : format);
        const FrameInfo& f = im.frame;
        if (im.status != HIPJPEG_STATUS_SUCCESS) continue;

        DecodeImage& d = desc_[i];  // (prepare() filled in what does not depend on the layout)
        if (im.has_transform) {
            // the pixel kernels write a full-frame intermediate picture (same format, 16-byte aligned rows); the geometry
            // pass copies the region, turned upright, into the caller's buffer
            const int np = (fmt == kOutInterleavedRGB || fmt == kOutInterleavedBGR || fmt == kOutY) ? 1 : 3;
            const int bpp = (fmt == kOutInterleavedRGB || fmt == kOutInterleavedBGR) ? 3 : 1;
            TransformImage t;
            memset(&t, 0, sizeof t);
            const uint32_t ipitch = (uint32_t)align_up((size_t)f.width * bpp, 16);
            for (int p = 0; p < np; p++) {
                t.dst[p] = static_cast<uint8_t*>(outputs[i].plane[p]);
                t.dst_pitch[p] = outputs[i].pitch[p];
                t.src_pitch[p] = ipitch;
                d.out_pitch[p] = ipitch;
                xform_plane_off.push_back(plane_total);
                plane_total += align_up((size_t)ipitch * f.height + 16, 256);
            }
            t.x0 = im.transform.x0;
            t.y0 = im.transform.y0;
            t.rw = im.transform.x1 - im.transform.x0;
            t.rh = im.transform.y1 - im.transform.y0;
            const bool swap = im.transform.orientation >= 5;
            t.out_w = swap ? t.rh : t.rw;
            t.out_h = swap ? t.rw : t.rh;
            t.orientation = im.transform.orientation;
            t.nplanes = np;
            t.bpp = bpp;
            im.xform_index = (int)xform_desc_.size();
            xform_desc_.push_back(t);
            max_xform_units += (size_t)(t.out_h + kTransformRowsPerUnit - 1) / kTransformRowsPerUnit;
        }
        for (int c = 0; c < f.ncomp; c++) {
            const Component& k = f.comp[c];
            DecodeComponent& dc = d.comp[c];
            const size_t nblk = (size_t)k.blocks_w * k.blocks_h;
            const size_t units = (nblk + kBlocksPerUnit - 1) / kBlocksPerUnit;
            // which components go through an intermediate plane
            bool needs_plane = (im.variant == -1) || (im.variant == -3) || (im.variant >= kVar11 && c > 0);
            bool to_output = (im.variant == -2) && (fmt == kOutPlanarYUV || c == 0);
            if (needs_plane) {
                dc.plane_pitch = (uint32_t)align_up((size_t)k.blocks_w * 8 + 16, 16);
                plane_off[(size_t)i * 4 + c] = plane_total;
                plane_total += align_up((size_t)dc.plane_pitch * k.blocks_h * 8 + 16, 256);
            }
            if (needs_plane || to_output) max_units += units;
            if (c == 0 && im.variant >= 0) max_units += (size_t)((k.blocks_w + kLumaTileW - 1) / kLumaTileW) * (size_t)((k.blocks_h + kLumaTileH - 1) / kLumaTileH);
        }
        if (im.variant == -1 || im.variant == -3) max_units += (size_t)f.height;
        if (im.gpu_prog) {
            // progressive: every scan is staged and destuffed like a baseline scan; the walk and replay kernels take it from there
            im.prog_index = (int)prog_to_image_.size();
            prog_to_image_.push_back(i);
            im.prog_huff_first = (uint32_t)prog_scan_total_;
            prog_scan_total_ += f.scans.size();
            for (size_t sidx = 0; sidx < f.scans.size(); sidx++) {
                const ScanHeader& sc = f.scans[sidx];
                const size_t raw_len = sc.data_end - sc.data_begin;
                im.prog_raw_offset[sidx] = huff_raw_total;  // relative; rebased below
                huff_raw_total += align_up(raw_len, 16) + 16;
                im.prog_stream_offset[sidx] = huff_stream_total;
                huff_stream_total += align_up(destuffed_capacity(sc), 64);
                im.prog_first_chunk[sidx] = (uint32_t)huff_chunks_total;
                huff_chunks_total += (raw_len + kDestuffChunk - 1) / kDestuffChunk;
                if (sc.ss != 0) {
                    const int cc = sc.comp_index[0];
                    const size_t nb = (size_t)((f.comp[cc].samp_w + 7) / 8) * (size_t)((f.comp[cc].samp_h + 7) / 8);
                    im.prog_pos_offset[sidx] = prog_pos_total * 4;
                    prog_pos_total += (nb + 63) & ~(size_t)63;
                    prog_slot_words_ = std::max<unsigned>(prog_slot_words_, (unsigned)prog_table_words(sc.ac[sc.ta[0]]));
                } else if (sc.ah == 0) {
                    for (int k = 0; k < sc.ncomp; k++)
                        prog_slot_words_ = std::max<unsigned>(prog_slot_words_, (unsigned)prog_table_words(sc.dc[sc.td[k]]));
                }
            }
            im.tables_offset = huff_pool_total;  // relative; rebased below
            huff_pool_total += align_up(im.pool_words * 2, 64);
            for (int c = 0; c < f.ncomp; c++) {  // compact DC planes, in the same scratch as the baseline path's
                im.dc_plane_offset[c] = huff_blocks_total * 2;
                const size_t nblk = (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h;
                huff_blocks_total += (nblk + 31) & ~(size_t)31;
                max_prog_units_ += (nblk + 255) / 256;
            }
        } else if (im.gpu_entropy) {
            im.huff_index = (int)huff_to_image_.size();
            huff_to_image_.push_back(i);
            const size_t raw_len = f.scans[0].data_end - f.scans[0].data_begin;
            const size_t cap = align_up(destuffed_capacity(f.scans[0]), 64);
            im.raw_offset = huff_raw_total;  // relative; rebased below
            huff_raw_total += align_up(raw_len, 16) + 16;
            im.stream_offset = huff_stream_total;  // offset into the device-only scratch
            huff_stream_total += cap;
            im.first_chunk = (uint32_t)huff_chunks_total;
            huff_chunks_total += (raw_len + kDestuffChunk - 1) / kDestuffChunk;
            const size_t nsub = (cap * 8 + kSubseqBits - 1) / kSubseqBits;
            huff_subseq_ub += nsub;
            max_huff_units_ += (nsub + kHuffOwn - 1) / kHuffOwn;
            const size_t pool_words = im.pool_words;
            im.tables_offset = huff_pool_total;  // relative; rebased below
            huff_pool_total += align_up(pool_words * 2, 64);
            max_pool_words_ = std::max(max_pool_words_, pool_words);
            im.dc_diff_offset = huff_blocks_total * 2;
            huff_blocks_total += (f.total_blocks() + 31) & ~(size_t)31;
            for (int c = 0; c < f.ncomp; c++) {  // DC planes live in the same scratch, behind the differences
                im.dc_plane_offset[c] = huff_blocks_total * 2;
                huff_blocks_total += ((size_t)f.comp[c].blocks_w * f.comp[c].blocks_h + 31) & ~(size_t)31;
            }
            max_huff_wunits_ += ((size_t)f.mcus_x * f.mcus_y + kHuffMcusPerWg - 1) / kHuffMcusPerWg + 1;
            im.boundary_offset = huff_boundary_total;  // relative; rebased below (restart boundaries + per-subsequence index)
            im.num_boundaries = (uint32_t)f.scans[0].rst_after.size();
            if (f.scans[0].restart_interval) huff_boundary_total += align_up(((size_t)im.num_boundaries + nsub + 1) * 4, 64);
            im.block_pos_offset = huff_blockpos_total * 4;
            huff_blockpos_total += (f.total_blocks() + 63) & ~(size_t)63;
        }
        coef_bytes_ += f.total_blocks() * 128;
        if (fmt == kOutPlanarYUV) {
            for (int c = 0; c < f.ncomp; c++) output_bytes_ += (uint64_t)f.comp[c].samp_w * f.comp[c].samp_h;
        } else {
            output_bytes_ += (uint64_t)f.width * f.height * (fmt == kOutY ? 1 : 3);
        }
    }

    // Staging layout (pinned mirror <-> device):  descriptors | work units | entropy descriptors, units, tables, destuffed
    // streams | coefficients of host-decoded images  ||  (device only from here) coefficients of GPU-decoded images
    const size_t ng = huff_to_image_.size();
    desc_offset_ = 0;
    units_offset_ = align_up(desc_offset_ + sizeof(DecodeImage) * (size_t)n, 256);
    huff_desc_offset_ = align_up(units_offset_ + sizeof(WorkUnit) * max_units, 256);
    huff_units_offset_ = align_up(huff_desc_offset_ + sizeof(HuffImage) * (ng + prog_scan_total_), 256);
    huff_wunits_offset_ = align_up(huff_units_offset_ + sizeof(HuffUnit) * max_huff_units_, 256);
    huff_dc_units_offset_ = align_up(huff_wunits_offset_ + sizeof(HuffUnit) * max_huff_wunits_, 256);
    huff_list_offset_ = align_up(huff_dc_units_offset_ + sizeof(HuffUnit) * ng * 4, 256);
    // (the HuffImage array holds the baseline images first, then one entry per scan of the progressive images)
    huff_chunk_units_offset_ = align_up(huff_list_offset_ + sizeof(uint32_t) * ng, 256);
    huff_drops_offset_ = align_up(huff_chunk_units_offset_ + sizeof(HuffUnit) * huff_chunks_total, 256);
    xform_desc_offset_ = align_up(huff_drops_offset_ + sizeof(uint32_t) * huff_chunks_total, 256);
    xform_units_offset_ = align_up(xform_desc_offset_ + sizeof(TransformImage) * xform_desc_.size(), 256);
    prog_desc_offset_ = align_up(xform_units_offset_ + sizeof(WorkUnit) * max_xform_units, 256);
    prog_units_offset_ = align_up(prog_desc_offset_ + sizeof(ProgImage) * prog_to_image_.size(), 256);
    const size_t tables_base = align_up(prog_units_offset_ + sizeof(HuffUnit) * max_prog_units_, 256);
    const size_t boundaries_base = align_up(tables_base + huff_pool_total, 256);
    const size_t streams_base = align_up(boundaries_base + huff_boundary_total, 256);
    coef_offset_ = align_up(streams_base + huff_raw_total, 256);
    raw_region_begin_ = streams_base;
    raw_region_end_ = coef_offset_;
    // HIPJPEG_DENSE_STAGING=1: dense int16 blocks for every host-decoded picture as in rounds 1-2 (A/B and cross-check aid)
    static const bool sparse_enabled = getenv("HIPJPEG_DENSE_STAGING") == nullptr;
    sparse_mode_ = sparse_enabled;
    host_coef_used_.store(0);
    for (int i = 0; i < n; i++) {
        images_[i].sparse = false;
        images_[i].host_coef_bytes = 0;
    }
    for (int pass = 0; pass < 2; pass++) {  // host-decoded images first, GPU-decoded ones behind the H2D boundary
        if (pass == 1) {
            h2d_bytes_ = coef_offset_ + coef_total;
            gpu_coef_begin_ = align_up(h2d_bytes_, 256);
            coef_total = gpu_coef_begin_ - coef_offset_;
        }
        for (int i = 0; i < n; i++) {
            PlannedImage& im = images_[i];
            if (im.status != HIPJPEG_STATUS_SUCCESS || (int)im.gpu_entropy != pass) continue;
            if (pass == 0 && sparse_mode_) coef_total += 256;  // (host-decoded pictures are placed while they are decoded: room for alignment)
            for (int c = 0; c < im.frame.ncomp; c++) {
                im.coef_offset[c] = coef_total;
                coef_total += (size_t)im.frame.comp[c].blocks_w * im.frame.comp[c].blocks_h * 128;
            }
            if (im.gpu_entropy) {
                im.raw_offset += streams_base;
                im.boundary_offset += boundaries_base;
                im.tables_offset += tables_base;
                if (im.gpu_prog)
                    for (size_t sidx = 0; sidx < im.frame.scans.size(); sidx++) im.prog_raw_offset[sidx] += streams_base;
            }
        }
    }
    staging_bytes_ = coef_offset_ + coef_total;
    gpu_coef_bytes_ = staging_bytes_ - gpu_coef_begin_;
    plane_bytes_ = plane_total;
    total_subseq_ = huff_subseq_ub;

    if (hipSetDevice(device_id_) != hipSuccess) return HIPJPEG_STATUS_NO_DEVICE;
    // the previous use of these buffers (H2D copy + kernels) must have drained before they are rewritten
    if (in_flight_) {
        if (hipEventSynchronize((hipEvent_t)done_event_) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
        in_flight_ = false;
    }
    hipjpegStatus_t st;
    if ((st = pinned_.reserve(h2d_bytes_ + 256)) != HIPJPEG_STATUS_SUCCESS) return st;
    if ((st = device_.reserve(staging_bytes_ + 256)) != HIPJPEG_STATUS_SUCCESS) return st;
    if ((st = planes_.reserve(plane_bytes_ + 256)) != HIPJPEG_STATUS_SUCCESS) return st;
    // device-only scratch of the entropy kernels: subsequence states | first block indices | change counters | DC differences
    work_first_block_ = align_up(total_subseq_ * 8, 256);
    work_changed_ = work_first_block_ + align_up(total_subseq_ * 4, 256);
    work_incoming_ = work_changed_ + 256;
    work_tail_ = align_up(work_incoming_ + max_huff_units_ * 8, 256);  // tail tasks (256 B per unit) + counts
    work_dc_diff_ = align_up(work_tail_ + max_huff_units_ * (kTailTaskBytes + 4), 256);
    work_block_pos_ = align_up(work_dc_diff_ + huff_blocks_total * 2, 256);
    work_drops_ = align_up(work_block_pos_ + huff_blockpos_total * 4, 256);
    work_prog_pos_ = align_up(work_drops_ + huff_chunks_total * 4, 256);
    work_group_sums_ = align_up(work_prog_pos_ + prog_pos_total * 4, 256);  // per block-pass workgroup: DC difference sums of its MCUs
    work_streams_ = align_up(work_group_sums_ + max_huff_wunits_ * 16, 256);
    if ((ng || prog_scan_total_) && (st = work_.reserve(work_streams_ + huff_stream_total + 256)) != HIPJPEG_STATUS_SUCCESS) return st;
    huff_images_.assign(ng + prog_scan_total_, HuffImage());
    prog_images_.assign(prog_to_image_.size(), ProgImage());

    for (int i = 0; i < n; i++) {
        if (images_[i].status != HIPJPEG_STATUS_SUCCESS) continue;
        DecodeImage& d = desc_[i];
        for (int c = 0; c < images_[i].frame.ncomp; c++) {
            images_[i].coef_offset[c] += coef_offset_;
            d.comp[c].coef = reinterpret_cast<const int16_t*>(device_.data() + images_[i].coef_offset[c]);
            if (images_[i].gpu_entropy) {
                d.comp[c].dc = reinterpret_cast<const int16_t*>(work_.data() + work_dc_diff_ + images_[i].dc_plane_offset[c]);
                d.comp[c].dc_stride = 1;
            } else {
                d.comp[c].dc = d.comp[c].coef;
                d.comp[c].dc_stride = 64;
            }
            if (plane_off[(size_t)i * 4 + c] != (size_t)-1) d.comp[c].plane = planes_.data() + plane_off[(size_t)i * 4 + c];
        }
    }
    {
        size_t k = 0;
        for (int i = 0; i < n; i++) {
            if (images_[i].status != HIPJPEG_STATUS_SUCCESS || !images_[i].has_transform) continue;
            TransformImage& t = xform_desc_[images_[i].xform_index];
            for (int p = 0; p < t.nplanes; p++, k++) {
                uint8_t* base = planes_.data() + xform_plane_off[k];
                t.src[p] = base;
                desc_[i].out[p] = base;
            }
        }
    }
    h2d_used_ = h2d_bytes_;
    if (statuses)
        for (int i = 0; i < n; i++) statuses[i] = images_[i].status;
    return HIPJPEG_STATUS_SUCCESS;
}

// The parser counted, per kDestuffChunk-byte chunk of the scan, the bytes that byte-stuffing removal drops; they ride to the device with
// the bitstream (the compact kernel reads them where round 1's count kernel left its results).
void DecodeBatch::stage_chunk_drops(const ScanHeader& sc, uint32_t first_chunk)
{
    static_assert(kScanChunkBytes == (size_t)kDestuffChunk, "the parser counts in the destuff kernels' chunks");
    uint32_t* dst = reinterpret_cast<uint32_t*>(pinned_.data() + huff_drops_offset_) + first_chunk;
    const size_t chunks = (sc.data_end - sc.data_begin + kScanChunkBytes - 1) / kScanChunkBytes;
    if (sc.chunk_drops.size() != chunks) {  // a frame that was parsed without the walk's counts: the device counts for this batch
        host_drops_missing_.store(true, std::memory_order_relaxed);
        return;
    }
    memcpy(dst, sc.chunk_drops.data(), chunks * sizeof(uint32_t));
}

void DecodeBatch::entropy_stage(int i)
{
    PlannedImage& im = images_[i];
    if (im.status != HIPJPEG_STATUS_SUCCESS) return;
    ScopedRange range(im.gpu_entropy ? "hipjpeg host stage (stage bitstream, is_gpu_huffman=1)" : "hipjpeg host stage (Huffman decode, is_gpu_huffman=0)");
    fault_point("entropy_stage");
    if (im.gpu_prog) {
        // progressive: stage every scan's bytes as they are, expand the scans' Huffman tables, describe scans and chains
        const FrameInfo& f = im.frame;
        for (size_t sidx = 0; sidx < f.scans.size(); sidx++) {
            const ScanHeader& sc = f.scans[sidx];
            const size_t len = sc.data_end - sc.data_begin;
            uint8_t* raw = pinned_.data() + im.prog_raw_offset[sidx];
            copy_to_staging(raw, im.data + sc.data_begin, len);
            memset(raw + len, 0x01, align_up(len, 16) + 16 - len);  // neither FF nor 00
            HuffImage& h = huff_images_[huff_to_image_.size() + im.prog_huff_first + sidx];
            memset(&h, 0, sizeof h);
            h.raw_bytes = (uint32_t)len;
            h.first_chunk = im.prog_first_chunk[sidx];
            im.stream_bytes += (uint32_t)len;
            stage_chunk_drops(sc, h.first_chunk);
        }
        fill_prog_image(f, &prog_images_[im.prog_index], reinterpret_cast<uint16_t*>(pinned_.data() + im.tables_offset));
        for (int c = 0; c < f.ncomp; c++) im.coef_or[c] = 32767u;  // successive approximation: any int16 may come out
        return;
    }
    if (im.gpu_entropy) {
        // host part of the GPU entropy path: stage the scan's bytes as they are (the device removes the byte stuffing),
        // expand the Huffman tables, describe the scan
        const ScanHeader& sc = im.frame.scans[0];
        im.stream_bytes = (uint32_t)(sc.data_end - sc.data_begin);
        uint8_t* raw = pinned_.data() + im.raw_offset;
        if (!im.input_pinned) {  // (else: transfer() sends the bytes from where they are)
            copy_to_staging(raw, im.data + sc.data_begin, im.stream_bytes);
            memset(raw + im.stream_bytes, 0x01, align_up((size_t)im.stream_bytes, 16) + 16 - im.stream_bytes);  // neither FF nor 00
        }
        stage_chunk_drops(sc, im.first_chunk);
        HuffImage& h = huff_images_[im.huff_index];
        fill_huff_image(im.frame, im.stream_bytes, &h);
        build_gpu_pool(sc, &h, reinterpret_cast<uint16_t*>(pinned_.data() + im.tables_offset));
        if (sc.restart_interval) {
            // restart boundaries in bits, and for every subsequence the first boundary at or behind its first bit
            uint32_t* bnd = reinterpret_cast<uint32_t*>(pinned_.data() + im.boundary_offset);
            uint32_t* sub = bnd + im.num_boundaries;
            for (uint32_t b = 0; b < im.num_boundaries; b++) bnd[b] = sc.rst_after[b] * 8u;
            uint32_t idx = 0;
            for (uint32_t j = 0; j <= h.num_subseq; j++) {
                while (idx < im.num_boundaries && bnd[idx] < j * (uint32_t)kSubseqBits) idx++;
                sub[j] = idx;
            }
            h.restart_interval = (uint32_t)sc.restart_interval;
            h.num_boundaries = im.num_boundaries;
        }
        // magnitude bound for the 24-bit multiplier decision without seeing the coefficients: DC values live in int16,
        // AC magnitudes are below 2^(largest size category any AC table of the scan can code)
        for (int c = 0; c < im.frame.ncomp; c++) {
            int maxcat = 0;
            const HuffSpec& t = sc.ac[sc.ta[c]];
            int nv = 0;
            for (int l = 1; l <= 16; l++) nv += t.bits[l];
            for (int v = 0; v < nv; v++) maxcat = std::max(maxcat, t.vals[v] & 15);
            im.coef_or[c] = 32767u | ((1u << maxcat) - 1);
            im.ac_bound[c] = (1u << maxcat) - 1;
        }
        return;
    }
    EntropyStatus es;
    size_t dense_bytes = 0;
    for (int c = 0; c < im.frame.ncomp; c++) dense_bytes += (size_t)im.frame.comp[c].blocks_w * im.frame.comp[c].blocks_h * 128;
    if (sparse_mode_) {
        // Zero-run-compressed staging: the picture is decoded into this thread's scratch as a sparse stream (a record of the non-zero
        // coefficients per block), then given the next free bytes of the host-decoded region -- what crosses PCIe is the stream, a third
        // of the dense blocks for a q90 photograph.  Frames the format does not cover (progressive, several scans), and the odd picture
        // whose stream would be larger than its dense blocks, are placed the same way as dense blocks.
        static thread_local std::vector<uint8_t> scratch;
        size_t bytes = 0;
        bool sparse = sparse_staging_applies(im.frame);
        es = kEntropyOk;
        if (sparse) {
            scratch.resize(sparse_stream_capacity(im.frame));
            es = decode_coefficients_sparse(im.data, im.size, im.frame, scratch.data(), &bytes);
            if (es == kEntropyOk && bytes > dense_bytes) sparse = false;
        }
        if (es == kEntropyOk) {
            const size_t need = align_up(sparse ? bytes : dense_bytes, 256);
            const size_t at = coef_offset_ + host_coef_used_.fetch_add(need);
            im.host_coef_offset = at;
            im.host_coef_bytes = need;
            im.sparse = sparse;
            if (sparse) {
                copy_to_staging(pinned_.data() + at, scratch.data(), bytes);
            } else {
                int16_t* coef[4] = {nullptr, nullptr, nullptr, nullptr};
                size_t off = at;
                for (int c = 0; c < im.frame.ncomp; c++) {
                    im.coef_offset[c] = off;
                    coef[c] = reinterpret_cast<int16_t*>(pinned_.data() + off);
                    off += (size_t)im.frame.comp[c].blocks_w * im.frame.comp[c].blocks_h * 128;
                }
                es = decode_coefficients(im.data, im.size, im.frame, coef, im.coef_or);
            }
        }
    } else {
        int16_t* coef[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int c = 0; c < im.frame.ncomp; c++) coef[c] = reinterpret_cast<int16_t*>(pinned_.data() + im.coef_offset[c]);
        es = decode_coefficients(im.data, im.size, im.frame, coef, im.coef_or);
    }
    for (int c = 0; c < 4; c++) im.ac_bound[c] = im.coef_or[c];  // the OR covers the DC values too: an upper bound all the same
    switch (es) {
    case kEntropyOk: break;
    case kEntropyTruncated: im.status = HIPJPEG_STATUS_TRUNCATED; break;
    case kEntropyMissingTable: im.status = HIPJPEG_STATUS_BAD_JPEG; break;
    default: im.status = HIPJPEG_STATUS_CORRUPT; break;
    }
}

void DecodeBatch::finalize(hipjpegStatus_t* statuses)
{
    fault_point("finalize");
    generic_units_.clear();
    cmyk_units_.clear();
    plane_units_.clear();
    fused_plane_units_.clear();
    host_taken_.clear();
    for (int e = 0; e < kNumLumaLayouts; e++) {
        for (auto& v : luma_units_[e]) v.clear();
        for (auto& v : fused_luma_units_[e]) v.clear();
    }
    // HIPJPEG_FUSED_DECODE=1: the pixel kernels Huffman-decode the blocks themselves (decode_kernels.hip FUSED builds) instead of reading
    // coefficient blocks the block pass wrote to HBM.  Measured: HBM traffic of the step falls by more than half,
    // its TIME rises (2.72 -> 3.20 ms per 256 x 1080p): both halves are bound by instruction issue, and inside the pixel kernels only 32
    // of a wave's 64 lanes have a block to decode.  Off by default; kept as a switch and covered by the parity tests.
    static const bool fused_enabled = getenv("HIPJPEG_FUSED_DECODE") != nullptr && atoi(getenv("HIPJPEG_FUSED_DECODE")) != 0;
    fused_ = fused_enabled;
    const int n = (int)images_.size();
    // all or nothing per batch: with fused_ set the block pass is not launched at all, so every GPU-decoded baseline picture must be one the
    // FUSED builds cover completely -- a region of interest launches only the tiles that touch it, the other blocks would go undecoded and
    // unchecked
    for (int i = 0; i < n && fused_; i++) {
        const PlannedImage& im = images_[i];
        if (im.status == HIPJPEG_STATUS_SUCCESS && im.gpu_entropy && !im.gpu_prog && im.has_transform) fused_ = false;
    }
    for (int i = 0; i < n; i++) {
        PlannedImage& im = images_[i];
        if (statuses) statuses[i] = im.status;
        if (im.status != HIPJPEG_STATUS_SUCCESS) continue;
        const FrameInfo& f = im.frame;
        DecodeImage& d = desc_[i];
        if (sparse_mode_ && !im.gpu_entropy) {
            // placed while it was decoded (entropy_stage): point the descriptor at where the picture landed
            size_t blocks_before = 0;
            for (int c = 0; c < f.ncomp; c++) {
                DecodeComponent& dc = d.comp[c];
                if (im.sparse) {
                    dc.coef = reinterpret_cast<const int16_t*>(device_.data() + im.host_coef_offset);
                    dc.block_off = reinterpret_cast<const uint32_t*>(device_.data() + im.host_coef_offset) + blocks_before;
                    blocks_before += (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h;
                } else {
                    dc.coef = reinterpret_cast<const int16_t*>(device_.data() + im.coef_offset[c]);
                    dc.block_off = nullptr;
                }
                dc.dc = dc.coef;
                dc.dc_stride = 64;
            }
        }
        const OutFormat fmt = (OutFormat)d.out_format;
        const bool fused = fused_ && im.gpu_entropy && !im.gpu_prog;
        d.huff_index = fused ? (uint32_t)im.huff_index : 0u;
        std::vector<WorkUnit>& plane_list = fused ? fused_plane_units_ : plane_units_;
        auto& luma_lists = fused ? fused_luma_units_ : luma_units_;
        for (int c = 0; c < f.ncomp; c++) {
            const uint32_t nblk = (uint32_t)f.comp[c].blocks_w * f.comp[c].blocks_h;
            bool needs_plane = (im.variant == -1) || (im.variant == -3) || (im.variant >= kVar11 && c > 0);
            bool to_output = (im.variant == -2) && (fmt == kOutPlanarYUV || c == 0);
            if (needs_plane || to_output) {
                uint32_t mode = to_output ? (uint32_t)(kToOutput | (c << 8)) : (uint32_t)kToPlane;
                for (uint32_t b = 0; b < nblk; b += kBlocksPerUnit) plane_list.push_back(WorkUnit{(uint32_t)i, b, (uint32_t)c, mode});
            }
        }
        if (im.variant >= 0) {
            // luma tiles: 32 blocks wide x 4 block rows (one block row per wave); rows that are pure MCU padding are skipped
            uint32_t real_rows = (uint32_t)(f.height + 7) / 8, first_row = 0, first_col = 0, end_col = (uint32_t)f.comp[0].blocks_w;
            if (im.has_transform) {  // tiles that do not touch the region of interest are not decoded
                first_row = (uint32_t)im.transform.y0 / 8 / kLumaTileH * kLumaTileH;
                real_rows = std::min(real_rows, (uint32_t)(im.transform.y1 + 7) / 8);
                first_col = (uint32_t)im.transform.x0 / 8 / kLumaTileW * kLumaTileW;
                end_col = std::min(end_col, (uint32_t)(im.transform.x1 + 7) / 8);
            }
            // the everyday configuration has a kernel of its own (decode_kernels.hip, COMMON)
            const bool everyday = im.variant != kVarGray && d.color_model == 1 && (d.flags & kFlagFancyUpsampling) &&
                                  (im.variant == kVar11 || im.variant == kVar12 || d.comp[1].samp_w > 2);
            const int layout = !everyday ? 0 : (fmt == kOutInterleavedRGB || fmt == kOutInterleavedBGR) ? 1 : (fmt == kOutPlanarRGB || fmt == kOutPlanarBGR) ? 2 : 0;
            const int flavour = layout;
            // a ragged right edge of at most half a tile (1920 pixels = 7.5 tiles) is covered by narrow tiles, 16 x 8 blocks,
            // so that no wave runs half empty
            const uint32_t span = end_col - first_col, ragged = span % kLumaTileW;
            const uint32_t wide_end = (ragged != 0 && ragged <= kLumaTileW / 2) ? end_col - ragged : end_col;
            for (uint32_t by = first_row; by < real_rows; by += kLumaTileH)
                for (uint32_t bx = first_col; bx < wide_end; bx += kLumaTileW)
                    luma_lists[flavour][im.variant].push_back(WorkUnit{(uint32_t)i, bx, by, 0u});
            if (wide_end < end_col)
                for (uint32_t by = first_row; by < real_rows; by += 2 * kLumaTileH)
                    luma_lists[flavour][im.variant].push_back(WorkUnit{(uint32_t)i, wide_end, by, 1u});
        } else if (im.variant == -1) {
            for (int y = 0; y < f.height; y++) generic_units_.push_back(WorkUnit{(uint32_t)i, (uint32_t)y, 0u, 0u});
        } else if (im.variant == -3) {
            for (int y = 0; y < f.height; y++) cmyk_units_.push_back(WorkUnit{(uint32_t)i, (uint32_t)y, 0u, 0u});
        }
    }
    // what transfer() has to copy: in sparse mode the host-decoded region ends where the last picture was placed
    h2d_used_ = sparse_mode_ ? std::min(h2d_bytes_, align_up(coef_offset_ + host_coef_used_.load(), 256)) : h2d_bytes_;
    // write descriptors + unit tables into the staging area
    uint8_t* base = pinned_.data();
    if (n) memcpy(base + desc_offset_, desc_.data(), sizeof(DecodeImage) * (size_t)n);
    size_t off = units_offset_;
    auto put = [&](const std::vector<WorkUnit>& v, size_t* where) {
        *where = off;
        if (!v.empty()) memcpy(base + off, v.data(), v.size() * sizeof(WorkUnit));
        off += v.size() * sizeof(WorkUnit);
    };
    put(plane_units_, &unit_off_plane_);
    // The luma tiles of a tile row share chroma lines at their seams (fancy upsampling reads a sample to the left and to the right of a
    // tile: one more 128-byte line on either side), and the hardware deals workgroups to the eight XCDs -- eight L2 caches -- in turn.  So the
    // rows are dealt to eight queues and the list takes one tile of every queue in turn: position p goes to XCD p % 8, a row's tiles follow
    // each other on ONE XCD and find their neighbour's lines in its L2.  (HIPJPEG_ROW_MAJOR_TILES=1: the plain order, for A/B runs.)
    static const bool row_major = getenv("HIPJPEG_ROW_MAJOR_TILES") != nullptr;
    auto deal_rows_to_xcds = [&](std::vector<WorkUnit>& v) {
        if (row_major || v.size() < 64) return;
        constexpr size_t kXcds = 8;
        std::vector<WorkUnit> queue[kXcds], rest, out;
        out.reserve(v.size());
        auto flush = [&]() {  // one picture's tiles: a tile of every queue in turn, then its narrow edge tiles (the list stays sorted by picture)
            size_t taken[kXcds] = {0};
            for (bool any = true; any;) {
                any = false;
                for (size_t q = 0; q < kXcds; q++)
                    if (taken[q] < queue[q].size()) {
                        out.push_back(queue[q][taken[q]++]);
                        any = true;
                    }
            }
            out.insert(out.end(), rest.begin(), rest.end());
            for (auto& q : queue) q.clear();
            rest.clear();
        };
        size_t row = 0;
        for (size_t a = 0; a < v.size();) {
            size_t b = a + 1;
            while (b < v.size() && v[b].image == v[a].image && v[b].comp == v[a].comp && v[b].mode == v[a].mode) b++;
            if (a > 0 && v[a].image != v[a - 1].image) {
                flush();
                row = 0;
            }
            if (v[a].mode != 0) {
                rest.insert(rest.end(), v.begin() + (long)a, v.begin() + (long)b);
            } else {
                std::vector<WorkUnit>& q = queue[row++ % kXcds];
                q.insert(q.end(), v.begin() + (long)a, v.begin() + (long)b);
            }
            a = b;
        }
        flush();
        v.swap(out);
    };
    for (int e = 0; e < kNumLumaLayouts; e++)
        for (int k = 0; k < kNumLumaVariants; k++) deal_rows_to_xcds(luma_units_[e][k]);
    for (int e = 0; e < kNumLumaLayouts; e++)
        for (int k = 0; k < kNumLumaVariants; k++) put(luma_units_[e][k], &unit_off_luma_[e][k]);
    put(fused_plane_units_, &unit_off_fused_plane_);
    for (int e = 0; e < kNumLumaLayouts; e++)
        for (int k = 0; k < kNumLumaVariants; k++) put(fused_luma_units_[e][k], &unit_off_fused_luma_[e][k]);
    put(generic_units_, &unit_off_generic_);
    put(cmyk_units_, &unit_off_cmyk_);
    // geometry pass
    xform_units_.clear();
    for (int i = 0; i < n; i++) {
        const PlannedImage& im = images_[i];
        if (im.status != HIPJPEG_STATUS_SUCCESS || !im.has_transform) continue;
        const TransformImage& t = xform_desc_[im.xform_index];
        for (int oy = 0; oy < t.out_h; oy += kTransformRowsPerUnit) xform_units_.push_back(WorkUnit{(uint32_t)im.xform_index, (uint32_t)oy, 0u, 0u});
    }
    if (!xform_desc_.empty()) memcpy(base + xform_desc_offset_, xform_desc_.data(), sizeof(TransformImage) * xform_desc_.size());
    if (!xform_units_.empty()) memcpy(base + xform_units_offset_, xform_units_.data(), sizeof(WorkUnit) * xform_units_.size());

    // GPU entropy descriptors: batch-wide subsequence numbering, one workgroup per 256 subsequences of an image
    huff_units_.clear();
    huff_wunits_.clear();
    huff_chunk_units_.clear();
    huff_dc_units_.clear();
    huff_list_.clear();
    uint32_t first_subseq = 0;
    stream_bytes_total_ = 0;
    for (size_t g = 0; g < huff_to_image_.size(); g++) {
        PlannedImage& im = images_[huff_to_image_[g]];
        HuffImage& h = huff_images_[g];
        if (im.status != HIPJPEG_STATUS_SUCCESS) {
            h.num_subseq = 0;
            continue;
        }
        h.stream = work_.data() + work_streams_ + im.stream_offset;
        h.raw = device_.data() + im.raw_offset;
        h.raw_src = im.input_pinned ? im.input_device_view + im.frame.scans[0].data_begin : nullptr;
        if (h.restart_interval) {
            h.boundaries = reinterpret_cast<const uint32_t*>(device_.data() + im.boundary_offset);
            h.sub_boundary = h.boundaries + h.num_boundaries;
        }
        h.raw_bytes = im.stream_bytes;
        h.first_chunk = im.first_chunk;
        for (uint32_t c = 0; c * (uint32_t)kDestuffChunk < im.stream_bytes; c++) huff_chunk_units_.push_back(HuffUnit{(uint32_t)g, c});
        h.pool = reinterpret_cast<const uint16_t*>(device_.data() + im.tables_offset);
        h.dc_diff = reinterpret_cast<int16_t*>(work_.data() + work_dc_diff_ + im.dc_diff_offset);
        for (int c = 0; c < im.frame.ncomp; c++) h.dc_plane[c] = reinterpret_cast<int16_t*>(work_.data() + work_dc_diff_ + im.dc_plane_offset[c]);
        h.block_pos = reinterpret_cast<uint32_t*>(work_.data() + work_block_pos_ + im.block_pos_offset);
        for (uint32_t m = 0; m < h.mcus_x * h.mcus_y; m += kHuffMcusPerWg) huff_wunits_.push_back(HuffUnit{(uint32_t)g, m});
        for (int c = 0; c < im.frame.ncomp; c++) h.coef[c] = reinterpret_cast<int16_t*>(device_.data() + im.coef_offset[c]);
        h.first_subseq = first_subseq;
        first_subseq += h.num_subseq;
        for (uint32_t j = 0; j < h.num_subseq; j += kHuffOwn) huff_units_.push_back(HuffUnit{(uint32_t)g, j});
        if (h.restart_interval)  // the predictor starts over inside the image: the per-component scan (everything else: per MCU group)
            for (int c = 0; c < im.frame.ncomp; c++) huff_dc_units_.push_back(HuffUnit{(uint32_t)g, (uint32_t)c});
        huff_list_.push_back((uint32_t)g);
        stream_bytes_total_ += im.stream_bytes;
    }
    total_subseq_ = first_subseq;
    // progressive images: pointers of the scan descriptors, chunk units for the destuff kernels, units of the replay kernel
    prog_units_.clear();
    for (size_t q = 0; q < prog_images_.size(); q++) {
        PlannedImage& im = images_[prog_to_image_[q]];
        ProgImage& pi = prog_images_[q];
        if (im.status != HIPJPEG_STATUS_SUCCESS) {
            pi.num_scans = 0;
            pi.ncomp = 0;
            pi.dc_len = 0;
            memset(pi.chain_len, 0, sizeof pi.chain_len);
            continue;
        }
        const FrameInfo& f = im.frame;
        for (size_t sidx = 0; sidx < f.scans.size(); sidx++) {
            const size_t hidx = huff_to_image_.size() + im.prog_huff_first + sidx;
            HuffImage& h = huff_images_[hidx];
            h.stream = work_.data() + work_streams_ + im.prog_stream_offset[sidx];
            h.raw = device_.data() + im.prog_raw_offset[sidx];
            ProgScan& ps = pi.scan[sidx];
            ps.stream = h.stream;
            ps.huff_image = (uint32_t)hidx;
            ps.block_pos = f.scans[sidx].ss != 0 ? reinterpret_cast<uint32_t*>(work_.data() + work_prog_pos_ + im.prog_pos_offset[sidx]) : nullptr;
            for (uint32_t c = 0; c * (uint32_t)kDestuffChunk < h.raw_bytes; c++) huff_chunk_units_.push_back(HuffUnit{(uint32_t)hidx, c});
        }
        pi.pool = reinterpret_cast<const uint16_t*>(device_.data() + im.tables_offset);
        for (int c = 0; c < f.ncomp; c++) {
            pi.coef[c] = reinterpret_cast<int16_t*>(device_.data() + im.coef_offset[c]);
            pi.dc_plane[c] = reinterpret_cast<int16_t*>(work_.data() + work_dc_diff_ + im.dc_plane_offset[c]);
            const uint32_t nblk = (uint32_t)f.comp[c].blocks_w * f.comp[c].blocks_h;
            for (uint32_t b = 0; b < nblk; b += 256) prog_units_.push_back(HuffUnit{(uint32_t)q, ((uint32_t)c << 28) | b});
        }
        stream_bytes_total_ += im.stream_bytes;
    }
    if (!prog_images_.empty()) {
        memcpy(base + prog_desc_offset_, prog_images_.data(), sizeof(ProgImage) * prog_images_.size());
        if (!prog_units_.empty()) memcpy(base + prog_units_offset_, prog_units_.data(), sizeof(HuffUnit) * prog_units_.size());
    }
    if (!huff_images_.empty()) {
        memcpy(base + huff_desc_offset_, huff_images_.data(), sizeof(HuffImage) * huff_images_.size());
        if (!huff_units_.empty()) memcpy(base + huff_units_offset_, huff_units_.data(), sizeof(HuffUnit) * huff_units_.size());
        if (!huff_wunits_.empty()) memcpy(base + huff_wunits_offset_, huff_wunits_.data(), sizeof(HuffUnit) * huff_wunits_.size());
        if (!huff_dc_units_.empty()) memcpy(base + huff_dc_units_offset_, huff_dc_units_.data(), sizeof(HuffUnit) * huff_dc_units_.size());
        if (!huff_list_.empty()) memcpy(base + huff_list_offset_, huff_list_.data(), sizeof(uint32_t) * huff_list_.size());
        if (!huff_chunk_units_.empty())
            memcpy(base + huff_chunk_units_offset_, huff_chunk_units_.data(), sizeof(HuffUnit) * huff_chunk_units_.size());
    }
    finalized_ = true;
}

hipjpegStatus_t DecodeBatch::transfer(void* stream, bool kernels_on_other_stream)
{
    if (!finalized_) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    ScopedRange range("hipjpeg transfer to device");
    fault_point("transfer");
    entropy_done_ = false;
    pixels_launched_ = false;
    if (h2d_used_ == 0) return HIPJPEG_STATUS_SUCCESS;
    // only descriptors, bitstreams of GPU-decoded images and coefficients of host-decoded images cross PCIe
    zero_copy_images_ = 0;
    for (const PlannedImage& im : images_) zero_copy_images_ += (im.status == HIPJPEG_STATUS_SUCCESS && im.gpu_entropy && !im.gpu_prog && im.input_pinned) ? 1 : 0;
    hipError_t e = hipSuccess;
    if (zero_copy_images_ == 0) {
        e = hipMemcpyAsync(device_.data(), pinned_.data(), h2d_used_, hipMemcpyHostToDevice, (hipStream_t)stream);
    } else {
        // everything but the staged bitstreams in two pieces (in front of and behind their region), the region's padding bytes in one
        // fill (neither FF nor 00, see entropy_stage), then one copy per scan: from the caller's pinned memory, or from the staging area
        // for the images of the batch that are not zero-copy; the zero-copy ones are pulled over by ONE kernel (gather_raw_kernel)
        hipStream_t s = (hipStream_t)stream;
        const size_t rb = std::min(raw_region_begin_, h2d_used_), re = std::min(raw_region_end_, h2d_used_);
        if (rb > 0) e = hipMemcpyAsync(device_.data(), pinned_.data(), rb, hipMemcpyHostToDevice, s);
        if (e == hipSuccess && re > rb) e = hipMemsetAsync(device_.data() + rb, 0x01, re - rb, s);
        if (e == hipSuccess && h2d_used_ > re) e = hipMemcpyAsync(device_.data() + re, pinned_.data() + re, h2d_used_ - re, hipMemcpyHostToDevice, s);
        for (const PlannedImage& im : images_) {
            if (e != hipSuccess) break;
            if (im.status != HIPJPEG_STATUS_SUCCESS || !im.gpu_entropy) continue;
            if (im.gpu_prog) {
                for (size_t sidx = 0; sidx < im.frame.scans.size() && e == hipSuccess; sidx++) {
                    const ScanHeader& sc = im.frame.scans[sidx];
                    const size_t len = sc.data_end - sc.data_begin;
                    if (len) e = hipMemcpyAsync(device_.data() + im.prog_raw_offset[sidx], pinned_.data() + im.prog_raw_offset[sidx], align_up(len, 16) + 16, hipMemcpyHostToDevice, s);
                }
                continue;
            }
            if (!im.input_pinned)  // (the pinned ones: one kernel for all of them, below)
                e = hipMemcpyAsync(device_.data() + im.raw_offset, pinned_.data() + im.raw_offset, align_up((size_t)im.stream_bytes, 16) + 16, hipMemcpyHostToDevice, s);
        }
        if (e == hipSuccess &&
            launch_gather_raw(reinterpret_cast<const HuffImage*>(device_.data() + huff_desc_offset_),
                              reinterpret_cast<const HuffUnit*>(device_.data() + huff_chunk_units_offset_), (int)huff_chunk_units_.size(), stream) != 0)
            e = hipErrorLaunchFailure;
    }
    if (e == hipSuccess && kernels_on_other_stream) {
        if (!copied_event_) {
            hipEvent_t ev;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
            copied_event_ = ev;
        }
        e = hipEventRecord((hipEvent_t)copied_event_, (hipStream_t)stream);
        copy_pending_ = true;
    }
    return e == hipSuccess ? HIPJPEG_STATUS_SUCCESS : HIPJPEG_STATUS_HIP_ERROR;
}

// GPU entropy stage: remove the byte stuffing, synchronise the subsequence decoders, scan block counts, write the
// coefficient blocks, integrate DC.  The common case is enqueued without any host round trip: launch 1 leaves every workgroup in
// a local fixpoint, launch 2 repairs the workgroup boundaries and counts the workgroups whose outgoing state moved; when
// that count is zero the states are the global fixpoint (no workgroup consumed a state that changed afterwards).  Only if
// it is not zero -- a correction crossed a whole workgroup -- more launches follow and the write passes are repeated.
// Blocks at the end: the host needs the per-image status to fall back to its own entropy decoder for streams the kernels
// flagged (corrupt or truncated data).
// Everything of the entropy stage that needs no answer from the device: enqueued, not waited for.
hipjpegStatus_t DecodeBatch::enqueue_gpu_entropy(void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    entropy_done_ = true;
    if (huff_units_.empty() && prog_units_.empty()) return HIPJPEG_STATUS_SUCCESS;
    EntropyLaunch L = entropy_launch_args();
    if (huff_chunk_units_.empty() && hipMemsetAsync(L.changed, 0, 256, s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;  // (else: the compact kernel clears them)
    // the per-chunk counts of bytes to drop came up with the bitstreams (the parser's marker walk sees every FF anyway);
    // HIPJPEG_DEVICE_DESTUFF_COUNT=1 counts them on the device as round 1 did (A/B and cross-check aid)
    static const bool force_device_count = getenv("HIPJPEG_DEVICE_DESTUFF_COUNT") != nullptr;
    const bool device_count = force_device_count || host_drops_missing_.load(std::memory_order_relaxed);
    if (launch_destuff(L.dimg, reinterpret_cast<const HuffUnit*>(device_.data() + huff_chunk_units_offset_), (int)huff_chunk_units_.size(),
                       device_count ? reinterpret_cast<uint32_t*>(work_.data() + work_drops_) : reinterpret_cast<uint32_t*>(device_.data() + huff_drops_offset_),
                       device_count, L.changed, stream) != 0)
        return HIPJPEG_STATUS_HIP_ERROR;
    if (!prog_units_.empty()) {
        const hipjpegStatus_t ps = enqueue_progressive(stream);
        if (ps != HIPJPEG_STATUS_SUCCESS) return ps;
    }
    if (huff_units_.empty()) {
        entropy_pending_ = true;
        return HIPJPEG_STATUS_SUCCESS;
    }
    static const int tail_after = getenv("HIPJPEG_TAIL_AFTER") ? atoi(getenv("HIPJPEG_TAIL_AFTER")) : 2;  // tuning aid; 0 = no tail kernel
    uint16_t* tail_tasks = reinterpret_cast<uint16_t*>(work_.data() + work_tail_);
    uint32_t* tail_count = reinterpret_cast<uint32_t*>(work_.data() + work_tail_ + (size_t)max_huff_units_ * kTailTaskBytes);
    if (launch_huff_sync(L.dimg, L.dunits, L.nunits, L.states, L.incoming, L.changed, 1, tail_after > 0 ? tail_after : 1 << 20,
                         tail_after > 0 ? tail_tasks : nullptr, tail_after > 0 ? tail_count : nullptr, L.pool_bytes, stream) != 0)
        return HIPJPEG_STATUS_HIP_ERROR;
    if (launch_huff_sync(L.dimg, L.dunits, L.nunits, L.states, L.incoming, L.changed, 0, 1 << 20, nullptr, nullptr, L.pool_bytes, stream, 1) != 0)
        return HIPJPEG_STATUS_HIP_ERROR;
    if (!entropy_write_passes(L, stream)) return HIPJPEG_STATUS_HIP_ERROR;
    if (hipMemcpyAsync(L.host_changed, L.changed, 8 * sizeof(unsigned int), hipMemcpyDeviceToHost, s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
    if (hipMemcpyAsync(L.himg, L.dimg, sizeof(HuffImage) * huff_images_.size(), hipMemcpyDeviceToHost, s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
    entropy_pending_ = true;
    return HIPJPEG_STATUS_SUCCESS;
}

// Progressive images: walk (block start positions of every scan, DC planes), replay (coefficient blocks), verdicts back.
hipjpegStatus_t DecodeBatch::enqueue_progressive(void* stream)
{
    ProgImage* dprog = reinterpret_cast<ProgImage*>(device_.data() + prog_desc_offset_);
    const HuffImage* dimg = reinterpret_cast<const HuffImage*>(device_.data() + huff_desc_offset_);
    const unsigned slot = (unsigned)align_up(std::max<unsigned>(prog_slot_words_, 256u), 64);
    unsigned dc_slots = 1, ac_waves = 0, rings = 0;  // per workgroup (= image): DC table slots, AC scans (one wave each), hand-over rings
    for (const ProgImage& pi : prog_images_) {
        unsigned scans = 0, handovers = 0;
        for (int c = 0; c < 4; c++) {
            scans += pi.chain_len[c];
            handovers += pi.chain_len[c] > 0 ? pi.chain_len[c] - 1 : 0;
        }
        ac_waves = std::max(ac_waves, scans);
        rings = std::max(rings, handovers);
        for (uint32_t k = 0; k < pi.num_scans; k++)
            if (pi.scan[k].ss == 0 && pi.scan[k].ah == 0) dc_slots = std::max<unsigned>(dc_slots, pi.scan[k].ncomp);
    }
    static const bool debug_stats = getenv("HIPJPEG_DEBUG_TIMING") != nullptr;
    if (debug_stats)
        fprintf(stderr, "[hipjpeg] progressive walk: %zu images, %u + 1 waves per workgroup, %u table slots of %u entries, %u rings\n", prog_images_.size(),
                ac_waves, dc_slots + ac_waves, slot, rings);
    if (launch_prog_walk(dprog, dimg, (int)prog_images_.size(), slot, dc_slots, ac_waves, rings, stream) != 0) return HIPJPEG_STATUS_HIP_ERROR;
    if (launch_prog_replay(dprog, dimg, reinterpret_cast<const HuffUnit*>(device_.data() + prog_units_offset_), (int)prog_units_.size(), slot, stream) != 0)
        return HIPJPEG_STATUS_HIP_ERROR;
    if (hipMemcpyAsync(pinned_.data() + prog_desc_offset_, dprog, sizeof(ProgImage) * prog_images_.size(), hipMemcpyDeviceToHost,
                       (hipStream_t)stream) != hipSuccess)
        return HIPJPEG_STATUS_HIP_ERROR;
    return HIPJPEG_STATUS_SUCCESS;
}

DecodeBatch::EntropyLaunch DecodeBatch::entropy_launch_args()
{
    EntropyLaunch L;
    L.dimg = reinterpret_cast<HuffImage*>(device_.data() + huff_desc_offset_);
    L.dunits = reinterpret_cast<const HuffUnit*>(device_.data() + huff_units_offset_);
    L.dwunits = reinterpret_cast<const HuffUnit*>(device_.data() + huff_wunits_offset_);
    L.ddc = reinterpret_cast<const HuffUnit*>(device_.data() + huff_dc_units_offset_);
    L.dlist = reinterpret_cast<const uint32_t*>(device_.data() + huff_list_offset_);
    L.states = reinterpret_cast<unsigned long long*>(work_.data());
    L.first_block = reinterpret_cast<uint32_t*>(work_.data() + work_first_block_);
    L.group_sums = reinterpret_cast<int32_t*>(work_.data() + work_group_sums_);
    L.changed = reinterpret_cast<unsigned int*>(work_.data() + work_changed_);
    L.incoming = reinterpret_cast<unsigned long long*>(work_.data() + work_incoming_);
    L.pool_bytes = (unsigned)align_up(max_pool_words_ * 2, 256);
    L.nunits = (int)huff_units_.size();
    L.host_changed = reinterpret_cast<unsigned int*>(pinned_.data() + h2d_bytes_);  // 256 spare bytes behind the staged data
    L.himg = reinterpret_cast<HuffImage*>(pinned_.data() + huff_desc_offset_);
    return L;
}

bool DecodeBatch::entropy_write_passes(const EntropyLaunch& L, void* stream)
{
    return launch_huff_scan(L.dimg, L.dlist, (int)huff_list_.size(), L.states, L.first_block, stream) == 0 &&
           launch_huff_write(L.dimg, L.dunits, L.nunits, L.dwunits, (int)huff_wunits_.size(), L.states, L.first_block, L.group_sums, L.pool_bytes,
                             stream, fused_) == 0 &&
           launch_huff_dc(L.dimg, L.ddc, (int)huff_dc_units_.size(), L.dwunits, (int)huff_wunits_.size(), L.group_sums, stream) == 0;
}

hipjpegStatus_t DecodeBatch::wait_done()
{
    if (!in_flight_) return HIPJPEG_STATUS_SUCCESS;
    return hipEventSynchronize((hipEvent_t)done_event_) == hipSuccess ? HIPJPEG_STATUS_SUCCESS : HIPJPEG_STATUS_HIP_ERROR;
}

// Waits for the work enqueued on `stream` and settles what the entropy kernels reported: extra synchronisation launches if
// a correction crossed a whole workgroup, host entropy decoding for streams the kernels flagged (corrupt or truncated
// data) -- after either, the pixel kernels run again.  Per-image statuses are final afterwards.
hipjpegStatus_t DecodeBatch::resolve(void* stream)
{
    if (!entropy_pending_) return HIPJPEG_STATUS_SUCCESS;
    ScopedRange range("hipjpeg resolve (GPU entropy verdicts)");
    fault_point("resolve");
    hipStream_t s = (hipStream_t)stream;
    // wait for THIS batch's work only (a later batch may already be queued on the same stream)
    if (in_flight_ ? hipEventSynchronize((hipEvent_t)done_event_) != hipSuccess : hipStreamSynchronize(s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
    entropy_pending_ = false;
    EntropyLaunch L = entropy_launch_args();
    HuffImage* dimg = L.dimg;
    const HuffUnit* dunits = L.dunits;
    unsigned long long* states = L.states;
    unsigned long long* incoming = L.incoming;
    unsigned int* changed = L.changed;
    unsigned int* host_changed = L.host_changed;
    HuffImage* himg = L.himg;
    const unsigned pool_bytes = L.pool_bytes;
    const int nunits = L.nunits;
    bool redo_pixels = false;
    auto write_passes = [&]() -> bool { return entropy_write_passes(L, stream); };
    last_sync_launches_ = 2;
    sync_rounds_total_ = host_changed[2];
    sync_rounds_max_ = host_changed[3];
    static const bool debug_stats = getenv("HIPJPEG_DEBUG_TIMING") != nullptr;
    if (debug_stats)
        fprintf(stderr, "[hipjpeg] entropy: %d workgroups, launch 1 rounds avg %.2f max %u, tail rounds avg %.2f max %u; launch 2: %u rounds in total, max %u, %u boundary changes\n",
                nunits, (double)sync_rounds_total_ / nunits, sync_rounds_max_, (double)host_changed[6] / nunits, host_changed[7], host_changed[4],
                host_changed[5], host_changed[0]);
    const bool has_baseline = !huff_units_.empty();
    bool converged = has_baseline ? *host_changed == 0 : true;
    // The ripple launch queued with the batch (pass 1) left groups whose last end state still moved: corrections that cross more
    // than one group border.  A few more launches settle a photograph; a periodic stream (stripes, a test pattern) would need one
    // launch per group, each a sequential walk through 255 subsequences -- those images go to the host decoder instead
    // (HuffImage::gave_up / moved_pass say which), 25 times faster for them.
    constexpr int kExtraRippleLaunches = 4;
    uint32_t last_pass = 1;
    std::vector<char> unsettled(huff_images_.size(), 0);
    if (!converged) {
        for (int pass = 0; pass < kExtraRippleLaunches && !converged; pass++) {
            if (hipMemsetAsync(changed, 0, sizeof(unsigned int), s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
            if (launch_huff_sync(dimg, dunits, nunits, states, incoming, changed, 0, 1 << 20, nullptr, nullptr, pool_bytes, stream, ++last_pass) != 0)
                return HIPJPEG_STATUS_HIP_ERROR;
            if (hipMemcpyAsync(host_changed, changed, sizeof(unsigned int), hipMemcpyDeviceToHost, s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
            if (hipStreamSynchronize(s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
            last_sync_launches_++;
            converged = *host_changed == 0;
        }
        // the write passes queued with the batch ran on unsettled states: fetch the flags, clear the verdicts, repeat them
        if (hipMemcpyAsync(himg, dimg, sizeof(HuffImage) * huff_images_.size(), hipMemcpyDeviceToHost, s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
        if (hipStreamSynchronize(s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
        for (size_t g = 0; g < huff_to_image_.size(); g++) {
            unsettled[g] = himg[g].gave_up != 0 || (!converged && himg[g].moved_pass == last_pass);
            himg[g].status = 0;
        }
        if (hipMemcpyAsync(dimg, himg, sizeof(HuffImage) * huff_images_.size(), hipMemcpyHostToDevice, s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
        if (!write_passes()) return HIPJPEG_STATUS_HIP_ERROR;
        // FUSED builds: the pixel kernels decode the blocks, so their verdicts belong to this read-back
        if (fused_ && pixels_launched_ && launch_pixel_kernels(stream, -1) != 0) return HIPJPEG_STATUS_HIP_ERROR;
        if (hipMemcpyAsync(himg, dimg, sizeof(HuffImage) * huff_images_.size(), hipMemcpyDeviceToHost, s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
        if (hipStreamSynchronize(s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
        redo_pixels = true;
    } else {
        for (size_t g = 0; g < huff_to_image_.size(); g++) unsettled[g] = himg[g].gave_up != 0;  // a group ran out of rounds in the tail kernel
    }
    // The kernels could not vouch for a stream (corrupt / truncated data, a periodic stream): the host entropy decoder produces
    // either the coefficients or the precise error.  Decoding runs on the batch's thread pool when there is one -- a batch of
    // test patterns hands over every image -- a helping of images at a time; the uploads follow on this thread.
    std::vector<int> takeover;
    for (size_t g = 0; g < huff_to_image_.size(); g++) {
        const PlannedImage& im = images_[huff_to_image_[g]];
        if (im.status != HIPJPEG_STATUS_SUCCESS) continue;
        if (himg[g].status == 0 && !unsettled[g]) continue;
        takeover.push_back(huff_to_image_[g]);
    }
    host_fallback_images_ = (int)takeover.size();
    const ProgImage* hprog = reinterpret_cast<const ProgImage*>(pinned_.data() + prog_desc_offset_);
    if (debug_stats && !prog_to_image_.empty()) {
        const ProgImage& q0 = hprog[0];
        for (uint32_t k = 0; k < q0.num_scans; k++) {
            // the same scan over all images of the batch that have it (min / mean / max), and where image 0's wave ran
            uint32_t lo = ~0u, hi = 0, cnt = 0;
            double sum = 0;
            for (size_t q = 0; q < prog_to_image_.size(); q++) {
                if (hprog[q].num_scans <= k) continue;
                const uint32_t t = hprog[q].scan[k].walk_ticks;
                lo = std::min(lo, t);
                hi = std::max(hi, t);
                sum += t;
                cnt++;
            }
            const uint32_t hw = q0.scan[k].pad_ticks[0];  // HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
            fprintf(stderr,
                    "[hipjpeg] progressive image 0 scan %u (ss %u se %u ah %u al %u): walk %.3f ms, of which waiting for neighbours %.3f ms; se %u cu %u simd %u; "
                    "all %u images: %.3f / %.3f / %.3f ms\n",
                    k, q0.scan[k].ss, q0.scan[k].se, q0.scan[k].ah, q0.scan[k].al, q0.scan[k].walk_ticks * 1e-5, q0.scan[k].wait_ticks * 1e-5, (hw >> 13) & 7u,
                    (hw >> 8) & 15u, (hw >> 4) & 3u, cnt, lo * 1e-5, cnt ? sum / cnt * 1e-5 : 0.0, hi * 1e-5);
            if (getenv("HIPJPEG_WALK_LAPS"))  // a library built with -DHJ_WALK_PROFILE (tools/walk_laps.sh): the packed lap timers
                fprintf(stderr, "[hipjpeg]   laps (cycles / count): fast %u / %u, event %u / %u, window %u / %u, block %u / %u\n",
                        (q0.scan[k].wait_ticks & 0xFFFFu) << 12, (q0.scan[k].pad_ticks[1] & 0xFFFFu) << 4, (q0.scan[k].wait_ticks >> 16) << 12,
                        (q0.scan[k].pad_ticks[1] >> 16) << 4, (q0.scan[k].pad_ticks[0] & 0xFFFFu) << 12, (q0.scan[k].pad_ticks[2] & 0xFFFFu) << 4,
                        (q0.scan[k].pad_ticks[0] >> 16) << 12, (q0.scan[k].pad_ticks[2] >> 16) << 4);
        }
    }
    for (size_t q = 0; q < prog_to_image_.size(); q++) {
        const PlannedImage& im = images_[prog_to_image_[q]];
        if (im.status != HIPJPEG_STATUS_SUCCESS || hprog[q].status == 0) continue;
        takeover.push_back(prog_to_image_[q]);
    }
    constexpr size_t kHelping = 32;
    for (size_t first = 0; first < takeover.size(); first += kHelping) {
        const size_t count = std::min(kHelping, takeover.size() - first);
        std::vector<std::vector<int16_t>> coefs(count);
        std::vector<EntropyStatus> verdict(count, kEntropyOk);
        auto decode_one = [&](int j) {
            const PlannedImage& im = images_[takeover[first + j]];
            const FrameInfo& f = im.frame;
            coefs[j].assign(f.total_blocks() * 64, 0);
            int16_t* coef[4] = {nullptr, nullptr, nullptr, nullptr};
            size_t off = 0;
            for (int c = 0; c < f.ncomp; c++) {
                coef[c] = coefs[j].data() + off;
                off += (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h * 64;
            }
            verdict[j] = decode_coefficients(im.data, im.size, f, coef, nullptr);
        };
        if (pool_ && count > 1)
            pool_->parallel_for((int)count, [&](int j, int) { decode_one(j); });
        else
            for (size_t j = 0; j < count; j++) decode_one((int)j);
        for (size_t j = 0; j < count; j++) {
            PlannedImage& im = images_[takeover[first + j]];
            if (verdict[j] != kEntropyOk) {
                im.status = verdict[j] == kEntropyTruncated ? HIPJPEG_STATUS_TRUNCATED : HIPJPEG_STATUS_CORRUPT;
                continue;
            }
            const FrameInfo& f = im.frame;
            size_t off = 0;
            for (int c = 0; c < f.ncomp; c++) {
                const size_t nblk = (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h;
                int16_t* coef = coefs[j].data() + off;
                off += nblk * 64;
                std::vector<int16_t> dc(nblk);  // the kernels take this image's DC values from its compact DC plane
                for (size_t b = 0; b < nblk; b++) {
                    dc[b] = coef[b * 64];
                    coef[b * 64] = 0;
                }
                if (hipMemcpy(device_.data() + im.coef_offset[c], coef, nblk * 128, hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(work_.data() + work_dc_diff_ + im.dc_plane_offset[c], dc.data(), nblk * 2, hipMemcpyHostToDevice) != hipSuccess)
                    return HIPJPEG_STATUS_HIP_ERROR;
            }
            if (fused_ && im.gpu_entropy && !im.gpu_prog) host_taken_.push_back(takeover[first + j]);
            redo_pixels = true;
        }
    }
    if (redo_pixels && pixels_launched_) {
        // the pixel kernels already ran on coefficients that have just been replaced
        if (launch_pixel_kernels(stream, -1) != 0 || hipStreamSynchronize(s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
    }
    return HIPJPEG_STATUS_SUCCESS;
}

// The plain K1 / K2 builds for the images in host_taken_: their units are those of the FUSED lists (same geometry), picked out by
// image, uploaded to a scratch buffer of their own.  Rare path (damaged or periodic streams): blocking copies are fine.
int DecodeBatch::launch_taken_pixels(void* stream, int which)
{
    const DecodeImage* dimg = reinterpret_cast<const DecodeImage*>(device_.data() + desc_offset_);
    static const int hs[kNumLumaVariants] = {0, 1, 2, 2, 1}, vs[kNumLumaVariants] = {0, 1, 1, 2, 2};
    std::vector<char> taken(images_.size(), 0);
    for (int i : host_taken_) taken[(size_t)i] = 1;
    std::vector<WorkUnit> units;
    struct Run {
        int layout, variant;  // layout < 0: K1
        size_t first, count;
    };
    std::vector<Run> runs;
    auto pick = [&](const std::vector<WorkUnit>& from, int layout, int variant) {
        const size_t first = units.size();
        for (const WorkUnit& u : from)
            if (taken[u.image]) units.push_back(u);
        if (units.size() > first) runs.push_back(Run{layout, variant, first, units.size() - first});
    };
    if (which < 0 || which == 0) pick(fused_plane_units_, -1, 0);
    if (which < 0 || which == 1)
        for (int e = 0; e < kNumLumaLayouts; e++)
            for (int k = 0; k < kNumLumaVariants; k++) pick(fused_luma_units_[e][k], e, k);
    if (units.empty()) return 0;
    const size_t bytes = units.size() * sizeof(WorkUnit);
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;  // an earlier launch may still read the scratch buffer
    if (bytes > taken_units_cap_) {
        if (taken_units_dev_) (void)hipFree(taken_units_dev_);
        taken_units_dev_ = nullptr;
        taken_units_cap_ = 0;
        if (hipMalloc(&taken_units_dev_, bytes * 2) != hipSuccess) return -1;
        taken_units_cap_ = bytes * 2;
    }
    if (hipMemcpy(taken_units_dev_, units.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return -1;
    const WorkUnit* du = static_cast<const WorkUnit*>(taken_units_dev_);
    int rc = 0;
    for (const Run& r : runs) {
        if (rc != 0) break;
        rc = r.layout < 0 ? launch_idct_plane(dimg, du + r.first, (int)r.count, stream)
                          : launch_luma_color(r.layout, hs[r.variant], vs[r.variant], dimg, du + r.first, (int)r.count, stream);
    }
    return rc;
}

int DecodeBatch::launch_pixel_kernels(void* stream, int which)
{
    const DecodeImage* dimg = reinterpret_cast<const DecodeImage*>(device_.data() + desc_offset_);
    auto units_at = [&](size_t off) { return reinterpret_cast<const WorkUnit*>(device_.data() + off); };
    static const int hs[kNumLumaVariants] = {0, 1, 2, 2, 1}, vs[kNumLumaVariants] = {0, 1, 1, 2, 2};
    static const bool debug_sync = getenv("HIPJPEG_DEBUG_SYNC") != nullptr;
    auto check = [&](const char* what, int n) {
        if (!debug_sync) return;
        hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        fprintf(stderr, "[hipjpeg] %s units=%d -> %s\n", what, n, hipGetErrorString(e));
    };
    int rc = 0;
    if (which == 7) {
        // measurement aid: K1 and K2 alternate over slices of HIPJPEG_PIXEL_CHUNK images, so that a slice's chroma
        // planes are still in the 256 MB Infinity Cache when its K2 reads them -- DESIGN.md 3.1 has what it measured
        static const int chunk = getenv("HIPJPEG_PIXEL_CHUNK") ? std::max(1, atoi(getenv("HIPJPEG_PIXEL_CHUNK"))) : 16;
        auto slice = [](const std::vector<WorkUnit>& v, uint32_t a, uint32_t b, size_t* first) {
            auto lo = std::lower_bound(v.begin(), v.end(), a, [](const WorkUnit& u, uint32_t x) { return u.image < x; });
            auto hi = std::lower_bound(lo, v.end(), b, [](const WorkUnit& u, uint32_t x) { return u.image < x; });
            *first = (size_t)(lo - v.begin());
            return (int)(hi - lo);
        };
        const uint32_t n = (uint32_t)images_.size();
        for (uint32_t a = 0; a < n && rc == 0; a += (uint32_t)chunk) {
            size_t first = 0;
            {
                const int cnt = slice(plane_units_, a, a + chunk, &first);
                rc = launch_idct_plane(dimg, units_at(unit_off_plane_) + first, cnt, stream);
            }
            for (int e = 0; e < kNumLumaLayouts; e++)
                for (int k = 0; k < kNumLumaVariants && rc == 0; k++) {
                    const int cnt = slice(luma_units_[e][k], a, a + chunk, &first);
                    rc = launch_luma_color(e, hs[k], vs[k], dimg, units_at(unit_off_luma_[e][k]) + first, cnt, stream);
                }
        }
        return rc;
    }
    HuffImage* himg = reinterpret_cast<HuffImage*>(device_.data() + huff_desc_offset_);
    const unsigned pool_bytes = (unsigned)align_up(max_pool_words_ * 2, 256);
    if (rc == 0 && (which < 0 || which == 0)) {
        rc = launch_idct_plane(dimg, units_at(unit_off_plane_), (int)plane_units_.size(), stream);
        check("idct_plane", (int)plane_units_.size());
        if (rc == 0) rc = launch_idct_plane_fused(dimg, units_at(unit_off_fused_plane_), (int)fused_plane_units_.size(), himg, pool_bytes, stream);
        check("idct_plane_fused", (int)fused_plane_units_.size());
    }
    for (int e = 0; e < kNumLumaLayouts; e++)
        for (int k = 0; k < kNumLumaVariants && rc == 0 && (which < 0 || which == 1); k++) {
            rc = launch_luma_color(e, hs[k], vs[k], dimg, units_at(unit_off_luma_[e][k]), (int)luma_units_[e][k].size(), stream);
            check("luma_color", (int)luma_units_[e][k].size());
            if (rc == 0)
                rc = launch_luma_color_fused(e, hs[k], vs[k], dimg, units_at(unit_off_fused_luma_[e][k]), (int)fused_luma_units_[e][k].size(), himg, pool_bytes,
                                             stream);
            check("luma_color_fused", (int)fused_luma_units_[e][k].size());
        }
    // images the host entropy decoder took over: the plain builds, on the coefficients it uploaded
    if (rc == 0 && !host_taken_.empty() && (which < 0 || which == 0 || which == 1)) rc = launch_taken_pixels(stream, which);
    if (rc == 0 && (which < 0 || which == 2)) {
        rc = launch_generic_color(dimg, units_at(unit_off_generic_), (int)generic_units_.size(), stream);
        check("generic_color", (int)generic_units_.size());
    }
    if (rc == 0 && (which < 0 || which == 2)) {
        rc = launch_cmyk_color(dimg, units_at(unit_off_cmyk_), (int)cmyk_units_.size(), stream);
        check("cmyk_color", (int)cmyk_units_.size());
    }
    if (rc == 0 && (which < 0 || which == 4)) {
        rc = launch_transform(reinterpret_cast<const TransformImage*>(device_.data() + xform_desc_offset_),
                              reinterpret_cast<const WorkUnit*>(device_.data() + xform_units_offset_), (int)xform_units_.size(), stream);
        check("transform", (int)xform_units_.size());
    }
    return rc;
}

hipjpegStatus_t DecodeBatch::launch(void* stream, int which, void* entropy_stream)
{
    if (!finalized_) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    ScopedRange range("hipjpeg device stage (launch)");
    fault_point("launch");
    // HIPJPEG_DEBUG_SYNC=1: synchronise after every launch and report which kernel failed (debug aid only)
    static const bool debug_sync = getenv("HIPJPEG_DEBUG_SYNC") != nullptr;
    auto check = [&](const char* what, int n) {
        if (!debug_sync) return;
        hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        fprintf(stderr, "[hipjpeg] %s units=%d -> %s\n", what, n, hipGetErrorString(e));
    };
    if (debug_sync) {
        for (size_t i = 0; i < images_.size(); i++) {
            const DecodeImage& d = desc_[i];
            fprintf(stderr, "[hipjpeg] img %zu st=%d var=%d %dx%d ncomp=%d hmax=%d vmax=%d coef=%p,%p,%p plane=%p,%p,%p pitch=%u,%u,%u out=%p opitch=%u\n",
                    i, (int)images_[i].status, images_[i].variant, (int)d.width, (int)d.height, (int)d.ncomp, (int)d.hmax, (int)d.vmax,
                    (const void*)d.comp[0].coef, (const void*)d.comp[1].coef, (const void*)d.comp[2].coef, (void*)d.comp[0].plane,
                    (void*)d.comp[1].plane, (void*)d.comp[2].plane, d.comp[0].plane_pitch, d.comp[1].plane_pitch, d.comp[2].plane_pitch,
                    (void*)d.out[0], d.out_pitch[0]);
        }
        fprintf(stderr, "[hipjpeg] device=%p+%zu planes=%p+%zu unit offs plane=%zu generic=%zu coef_off=%zu staging=%zu\n", (void*)device_.data(),
                device_.capacity(), (void*)planes_.data(), planes_.capacity(), unit_off_plane_, unit_off_generic_, coef_offset_, staging_bytes_);
        check("transfer", 0);
    }
    last_stream_ = stream;
    // With an entropy stream the (latency-bound) entropy kernels of this batch run beside the (VALU-bound) pixel kernels of
    // the batch before it; the pixel kernels on `stream` wait for them on the device.
    void* es_stream = (entropy_stream && which < 0 && !entropy_done_ && !(huff_units_.empty() && prog_units_.empty())) ? entropy_stream : stream;
    if (copied_event_ && copy_pending_) {
        // the H2D copy went out on another stream: the kernels wait for it on the device, the host does not
        if (hipStreamWaitEvent((hipStream_t)es_stream, (hipEvent_t)copied_event_, 0) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
        if (es_stream != stream && hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)copied_event_, 0) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
        copy_pending_ = false;
    }
    if ((which < 0 && !entropy_done_) || which == 3 || which == 6) {
        hipjpegStatus_t es = enqueue_gpu_entropy(es_stream);
        if (es != HIPJPEG_STATUS_SUCCESS) return es;
        if (which == 3) return resolve(stream);
        if (es_stream != stream) {
            if (!entropy_event_) {
                hipEvent_t ev;
                if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
                entropy_event_ = ev;
            }
            if (hipEventRecord((hipEvent_t)entropy_event_, (hipStream_t)es_stream) != hipSuccess ||
                hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)entropy_event_, 0) != hipSuccess)
                return HIPJPEG_STATUS_HIP_ERROR;
        }
    }
    if (which != 6 && launch_pixel_kernels(stream, which) != 0) return HIPJPEG_STATUS_HIP_ERROR;
    if (which < 0) pixels_launched_ = true;
    if (fused_ && entropy_pending_ && !huff_units_.empty() && (which < 0 || which == 1)) {
        // the FUSED pixel kernels are part of the entropy decode: a block they cannot decode flags its image -- the verdicts are
        // fetched behind them (the copy queued with the entropy stage saw the position pass's verdicts only)
        EntropyLaunch L = entropy_launch_args();
        if (hipMemcpyAsync(L.himg, L.dimg, sizeof(HuffImage) * huff_images_.size(), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess)
            return HIPJPEG_STATUS_HIP_ERROR;
    }
    if (!done_event_) {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
        done_event_ = ev;
    }
    if (hipEventRecord((hipEvent_t)done_event_, (hipStream_t)stream) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
    in_flight_ = true;
    return HIPJPEG_STATUS_SUCCESS;
}

void DecodeBatch::output_size(int i, int* w, int* h) const
{
    const PlannedImage& im = images_[i];
    int ow = im.frame.width, oh = im.frame.height;
    if (im.has_transform) {
        const int rw = im.transform.x1 - im.transform.x0, rh = im.transform.y1 - im.transform.y0;
        ow = im.transform.orientation >= 5 ? rh : rw;
        oh = im.transform.orientation >= 5 ? rw : rh;
    }
    *w = ow;
    *h = oh;
}

void DecodeBatch::stats(int32_t num_units[3], uint64_t* coef_bytes, uint64_t* output_bytes) const
{
    if (num_units) {
        num_units[0] = (int32_t)(plane_units_.size() + fused_plane_units_.size());
        num_units[1] = 0;
        for (int e = 0; e < kNumLumaLayouts; e++) {
            for (const auto& v : luma_units_[e]) num_units[1] += (int32_t)v.size();
            for (const auto& v : fused_luma_units_[e]) num_units[1] += (int32_t)v.size();
        }
        num_units[2] = (int32_t)generic_units_.size();
    }
    if (coef_bytes) *coef_bytes = coef_bytes_;
    if (output_bytes) *output_bytes = output_bytes_;
}

}  // namespace hipjpeg
