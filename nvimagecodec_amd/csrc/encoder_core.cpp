// encoder_core.cpp -- see encoder_core.h
#include "encoder_core.h"

#include <hip/hip_runtime_api.h>

#include <cstdlib>
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
    : device_id_(device_id), pinned_desc_(Buffer::kPinned, hooks), device_(Buffer::kDevice, hooks), pinned_coef_(Buffer::kPinned, hooks),
      henc_dev_(Buffer::kDevice, hooks), henc_dev2_(Buffer::kDevice, hooks), henc_pinned_(Buffer::kPinned, hooks), henc_out_(Buffer::kPinned, hooks)
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
    for (auto& v : unit_lists_) v.clear();
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
                fmt != HIPJPEG_OUTPUT_Y && fmt != HIPJPEG_OUTPUT_YUV_PLANAR)
                im.status = HIPJPEG_STATUS_UNSUPPORTED;
            if (fmt == HIPJPEG_OUTPUT_Y && g.ncomp != 1) im.status = HIPJPEG_STATUS_UNSUPPORTED;  // gray pixels carry no chroma
            if (fmt == HIPJPEG_OUTPUT_YUV_PLANAR && g.ncomp != 3) im.status = HIPJPEG_STATUS_UNSUPPORTED;
            const int nplanes = (fmt == HIPJPEG_OUTPUT_RGB_PLANAR || fmt == HIPJPEG_OUTPUT_BGR_PLANAR || fmt == HIPJPEG_OUTPUT_YUV_PLANAR) ? 3 : 1;
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
        d.in_format = fmt == HIPJPEG_OUTPUT_Y ? (uint32_t)kInGray : (uint32_t)fmt;  // RGBI/BGRI/planar/YUV values coincide with InFormat
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
                const int nat = kZigzagNatural[k], tr = (nat & 7) * 8 + (nat >> 3);
                d.qnat[t].magic[tr] = d.quant[t].magic[k];
                d.qnat[t].half16[tr] = (div >> 1) << 4;
            }
        }
        // tiles cover the real luma blocks only
        const int tiles_x = (g.real_w[0] + kTileBX - 1) / kTileBX, tiles_y = (g.real_h[0] + kTileBY - 1) / kTileBY;
        // interleaved RGB/BGR (any base address and pitch) into 4:2:0 / 4:2:2 / 4:4:4 has a kernel of its own (encode_kernels.hip
        // forward_pair_kernel).  HIPJPEG_ENCODE_ONE_LANE_KERNEL (read per batch) sends everything to the one-lane-per-block kernel:
        // the cross-check campaigns compare the two on the same pixels.
        int flavour = 0;
        const bool interleaved = fmt == HIPJPEG_OUTPUT_RGBI || fmt == HIPJPEG_OUTPUT_BGRI;
        const bool planar_rgb = fmt == HIPJPEG_OUTPUT_RGB_PLANAR || fmt == HIPJPEG_OUTPUT_BGR_PLANAR;
        if (g.ncomp == 3 && (interleaved || planar_rgb) && getenv("HIPJPEG_ENCODE_ONE_LANE_KERNEL") == nullptr) {
            flavour = (g.hs == 2 && g.vs == 2) ? 1 : (g.hs == 2 && g.vs == 1) ? 2 : (g.hs == 1 && g.vs == 1) ? 3 : 0;
            if (flavour != 0 && planar_rgb) flavour += 4;
        }
        if (fmt == HIPJPEG_OUTPUT_YUV_PLANAR) {
            // planes that are components already: one lane per real block of each component
            for (int c = 0; c < 3; c++)
                for (int b = 0; b < g.real_w[c] * g.real_h[c]; b += 256) unit_lists_[4].push_back(EncodeUnit{(uint32_t)i, (uint32_t)b, 0u, (uint32_t)c});
        } else {
            for (int ty = 0; ty < tiles_y; ty++)
                for (int tx = 0; tx < tiles_x; tx++) unit_lists_[flavour].push_back(EncodeUnit{(uint32_t)i, (uint32_t)tx, (uint32_t)ty, 0u});
        }
        pixel_bytes_ += (uint64_t)g.width * g.height * (g.ncomp == 1 ? 1 : 3);
        for (int c = 0; c < g.ncomp; c++) coef_bytes_ += (uint64_t)g.real_w[c] * g.real_h[c] * 128;
    }
    for (int f = 0; f < kUnitLists; f++) {  // one table, the flavours back to back
        unit_first_[f] = units_.size();
        units_.insert(units_.end(), unit_lists_[f].begin(), unit_lists_[f].end());
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
    const EncodeImage* dimg = reinterpret_cast<const EncodeImage*>(device_.data());
    const EncodeUnit* dunits = reinterpret_cast<const EncodeUnit*>(device_.data() + units_offset_);
    static const int pair_hs[4] = {0, 2, 2, 1}, pair_vs[4] = {0, 2, 1, 1};
    int rc = launch_forward(dimg, dunits + unit_first_[0], (int)unit_lists_[0].size(), stream);
    for (int f = 1; f < 4 && rc == 0; f++) rc = launch_forward_pair(pair_hs[f], pair_vs[f], false, dimg, dunits + unit_first_[f], (int)unit_lists_[f].size(), stream);
    for (int f = 1; f < 4 && rc == 0; f++)
        rc = launch_forward_pair(pair_hs[f], pair_vs[f], true, dimg, dunits + unit_first_[4 + f], (int)unit_lists_[4 + f].size(), stream);
    if (rc == 0) rc = launch_forward_planes(dimg, dunits + unit_first_[4], (int)unit_lists_[4].size(), stream);
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

// GPU entropy coder.  Two short host round trips: after the length scan (the bit-buffer sizes) and after the layout (the
// file sizes); everything else is queued on the stream the forward kernel ran on.
hipjpegStatus_t EncodeBatch::gpu_entropy_stage(std::vector<char>* todo)
{
    const int n = (int)images_.size();
    todo->assign(n, 0);
    gpu_entropy_images_ = 0;
    if (!launched_) return HIPJPEG_STATUS_INVALID_ARGUMENT;
    if (hipSetDevice(device_id_) != hipSuccess) return HIPJPEG_STATUS_NO_DEVICE;
    hipStream_t s = (hipStream_t)stream_;
    std::vector<int> idx;  // images taken here
    for (int i = 0; i < n; i++) {
        PlannedEncode& im = images_[i];
        im.gpu_bitstream = nullptr;
        im.gpu_bitstream_len = 0;
        if (im.status != HIPJPEG_STATUS_SUCCESS) continue;
        // HIPJPEG_NO_GPU_OPTIMIZED_HUFFMAN=1: per-image tables on the host coder as in round 2 (A/B and cross-check aid)
        static const bool gpu_optimized = getenv("HIPJPEG_NO_GPU_OPTIMIZED_HUFFMAN") == nullptr;
        if (im.params.restart_interval == 0 && !im.params.progressive && (gpu_optimized || !im.params.optimized_huffman))
            idx.push_back(i);
        else
            (*todo)[i] = 1;  // restart markers / progressive scans: the host coder
    }
    const int ng = (int)idx.size();
    if (ng == 0) return HIPJPEG_STATUS_SUCCESS;

    // ---- phase 1: descriptors, work units, code tables up; block lengths and their prefix sums
    std::vector<HencImage> desc(ng);
    std::vector<HencUnit> units;
    size_t total_blocks = 0;
    for (int g = 0; g < ng; g++) {
        const PlannedEncode& im = images_[idx[g]];
        const EncodeGeometry& eg = im.geom;
        HencImage& h = desc[g];
        memset(&h, 0, sizeof h);
        for (int c = 0; c < eg.ncomp; c++) {
            h.coef[c] = desc_[idx[g]].coef[c];
            h.blocks_w[c] = (uint32_t)eg.blocks_w[c];
            h.real_w[c] = (uint32_t)eg.real_w[c];
            h.real_h[c] = (uint32_t)eg.real_h[c];
        }
        h.mcus_x = (uint32_t)eg.mcus_x;
        h.mcus_y = (uint32_t)eg.mcus_y;
        h.ncomp = (uint32_t)eg.ncomp;
        h.hs = (uint32_t)eg.hs;
        h.vs = (uint32_t)eg.vs;
        h.bpm = eg.ncomp == 3 ? (uint32_t)(eg.hs * eg.vs + 2) : 1u;
        h.total_blocks = h.mcus_x * h.mcus_y * h.bpm;
        h.first_block = (uint32_t)total_blocks;
        for (uint32_t b = 0; b < h.total_blocks; b += 256) units.push_back(HencUnit{(uint32_t)g, b});
        total_blocks += (h.total_blocks + 63) & ~(size_t)63;
    }
    // images that want tables of their own (optimized_huffman): slot k of the per-image histogram / code-table arrays
    std::vector<int> opt_slot(ng, -1);
    int nopt = 0;
    for (int g = 0; g < ng; g++)
        if (images_[idx[g]].params.optimized_huffman) opt_slot[g] = nopt++;
    constexpr size_t kHistBytes = 2 * 2 * 256 * sizeof(uint32_t);
    const size_t o_units = align_up(sizeof(HencImage) * (size_t)ng, 256);
    const size_t o_tables = align_up(o_units + sizeof(HencUnit) * units.size(), 256);
    const size_t o_opt_tables = align_up(o_tables + sizeof(StandardCodeTables), 256);
    const size_t up1 = align_up(o_opt_tables + sizeof(StandardCodeTables) * (size_t)nopt, 256);  // uploaded part
    const size_t o_bits = up1;
    const size_t o_off = align_up(o_bits + total_blocks * 2, 256);
    const size_t o_total = align_up(o_off + total_blocks * 4, 256);
    const size_t o_hist = align_up(o_total + (size_t)ng * 4, 256);
    const size_t dev1 = o_hist + align_up(kHistBytes * (size_t)nopt, 256);
    hipjpegStatus_t st;
    if ((st = henc_dev_.reserve(dev1 + 256)) != HIPJPEG_STATUS_SUCCESS) return st;
    // pinned staging: phase-1 upload | totals | phase-2 upload (descriptors again, chunk units, headers) | lengths, offsets
    std::vector<std::vector<uint8_t>> headers(ng);
    for (int g = 0; g < ng; g++) {
        const PlannedEncode& im = images_[idx[g]];
        write_standard_headers(im.geom, im.qlum, im.qchr, &headers[g]);
    }
    const size_t p_totals = up1;
    const size_t p_hist = align_up(p_totals + (size_t)ng * 4, 256);
    const size_t p_up2 = align_up(p_hist + kHistBytes * (size_t)nopt, 256);
    // worst case for the chunk count: sized after the totals are known -> reserve generously from the block count
    // (a block codes to at most 64 * (16 + 15) bits; in practice ~10 bytes) -- the exact size is re-checked below
    if ((st = henc_pinned_.reserve(p_up2 + 256)) != HIPJPEG_STATUS_SUCCESS) return st;
    uint8_t* pin = henc_pinned_.data();
    memcpy(pin, desc.data(), sizeof(HencImage) * (size_t)ng);
    memcpy(pin + o_units, units.data(), sizeof(HencUnit) * units.size());
    standard_code_tables(reinterpret_cast<StandardCodeTables*>(pin + o_tables));
    uint8_t* dev = henc_dev_.data();
    const HencImage* dimg = reinterpret_cast<const HencImage*>(dev);
    const HencUnit* dunits = reinterpret_cast<const HencUnit*>(dev + o_units);
    const StandardCodeTables* dtables = reinterpret_cast<const StandardCodeTables*>(dev + o_tables);
    uint16_t* block_bits = reinterpret_cast<uint16_t*>(dev + o_bits);
    uint32_t* block_off = reinterpret_cast<uint32_t*>(dev + o_off);
    uint32_t* total_bits = reinterpret_cast<uint32_t*>(dev + o_total);
    if (nopt > 0) {
        // ---- phase 0: symbol statistics on the device, optimal tables on the host (a few dozen microseconds per image), tables back up.
        // The coefficients never leave HBM; what crosses PCIe is 4 KB of counts and 1.6 KB of tables per image.
        for (int g = 0; g < ng; g++)
            if (opt_slot[g] >= 0) desc[g].hist = reinterpret_cast<uint32_t*>(dev + o_hist + kHistBytes * (size_t)opt_slot[g]);
        memcpy(pin, desc.data(), sizeof(HencImage) * (size_t)ng);
        if (hipMemcpyAsync(dev, pin, o_opt_tables, hipMemcpyHostToDevice, s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
        if (hipMemsetAsync(dev + o_hist, 0, kHistBytes * (size_t)nopt, s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
        if (launch_henc_hist(dimg, dunits, (int)units.size(), stream_) != 0) return HIPJPEG_STATUS_HIP_ERROR;
        if (hipMemcpyAsync(pin + p_hist, dev + o_hist, kHistBytes * (size_t)nopt, hipMemcpyDeviceToHost, s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
        if (hipStreamSynchronize(s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
        for (int g = 0; g < ng; g++) {
            if (opt_slot[g] < 0) continue;
            const PlannedEncode& im = images_[idx[g]];
            const auto* counts = reinterpret_cast<const uint32_t(*)[2][256]>(pin + p_hist + kHistBytes * (size_t)opt_slot[g]);
            headers[g].clear();
            optimal_code_tables(counts, im.geom, im.qlum, im.qchr,
                                reinterpret_cast<StandardCodeTables*>(pin + o_opt_tables + sizeof(StandardCodeTables) * (size_t)opt_slot[g]), &headers[g]);
            desc[g].hist = nullptr;
            desc[g].tables = reinterpret_cast<const StandardCodeTables*>(dev + o_opt_tables + sizeof(StandardCodeTables) * (size_t)opt_slot[g]);
        }
        memcpy(pin, desc.data(), sizeof(HencImage) * (size_t)ng);
    }
    if (hipMemcpyAsync(dev, pin, up1, hipMemcpyHostToDevice, s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
    if (launch_henc_length(dimg, dunits, (int)units.size(), dtables, block_bits, stream_) != 0) return HIPJPEG_STATUS_HIP_ERROR;
    if (launch_henc_scan(dimg, ng, block_bits, block_off, total_bits, stream_) != 0) return HIPJPEG_STATUS_HIP_ERROR;
    uint32_t* h_totals = reinterpret_cast<uint32_t*>(pin + p_totals);
    if (hipMemcpyAsync(h_totals, total_bits, (size_t)ng * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
    if (hipStreamSynchronize(s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;

    // ---- phase 2: bit buffers, stuffing, file assembly
    std::vector<HencUnit> chunk_units;
    size_t raw_total = 0, arena_cap = 0, nchunks = 0;
    std::vector<size_t> raw_off(ng), hdr_off(ng);
    size_t hdr_total = 0;
    for (int g = 0; g < ng; g++) {
        HencImage& h = desc[g];
        const uint32_t pad = (8 - (h_totals[g] & 7)) & 7;
        h.raw_bytes = (h_totals[g] + pad) / 8;
        h.first_chunk = (uint32_t)nchunks;
        h.num_chunks = (h.raw_bytes + kHencChunk - 1) / kHencChunk;
        for (uint32_t c = 0; c < h.num_chunks; c++) chunk_units.push_back(HencUnit{(uint32_t)g, c});
        nchunks += h.num_chunks;
        raw_off[g] = raw_total;
        raw_total += align_up((size_t)h.raw_bytes + 16, 256);
        h.header_bytes = (uint32_t)headers[g].size();
        hdr_off[g] = hdr_total;
        hdr_total += align_up(headers[g].size(), 16);
        arena_cap += align_up((size_t)h.header_bytes + 2 * (size_t)h.raw_bytes + 2, 16);  // every byte could be 0xFF
    }
    const size_t q_units = align_up(sizeof(HencImage) * (size_t)ng, 256);
    const size_t q_headers = align_up(q_units + sizeof(HencUnit) * chunk_units.size(), 256);
    const size_t up2 = align_up(q_headers + hdr_total, 256);
    const size_t q_ff = up2;
    const size_t q_out = align_up(q_ff + nchunks * 4, 256);
    const size_t q_len = align_up(q_out + nchunks * 4, 256);
    const size_t q_foff = align_up(q_len + (size_t)ng * 4, 256);
    const size_t q_raw = align_up(q_foff + (size_t)ng * 8, 256);
    const size_t q_arena = align_up(q_raw + raw_total, 256);
    if ((st = henc_dev2_.reserve(q_arena + arena_cap + 256)) != HIPJPEG_STATUS_SUCCESS) return st;  // (arena unused when the files go to the host directly)
    // the pinned buffer is about to grow: keep what is still needed
    const size_t p_len = align_up(p_up2 + up2, 256);
    const size_t p_foff = align_up(p_len + (size_t)ng * 4, 256);
    if ((st = henc_pinned_.reserve(p_foff + (size_t)ng * 8 + 256)) != HIPJPEG_STATUS_SUCCESS) return st;
    pin = henc_pinned_.data();
    uint8_t* dev2 = henc_dev2_.data();
    for (int g = 0; g < ng; g++) {
        desc[g].raw = dev2 + q_raw + raw_off[g];
        desc[g].header = dev2 + q_headers + hdr_off[g];
        memcpy(pin + p_up2 + q_headers + hdr_off[g], headers[g].data(), headers[g].size());
    }
    memcpy(pin + p_up2, desc.data(), sizeof(HencImage) * (size_t)ng);
    memcpy(pin + p_up2 + q_units, chunk_units.data(), sizeof(HencUnit) * chunk_units.size());
    const HencImage* dimg2 = reinterpret_cast<const HencImage*>(dev2);
    const HencUnit* dchunks = reinterpret_cast<const HencUnit*>(dev2 + q_units);
    uint32_t* chunk_ff = reinterpret_cast<uint32_t*>(dev2 + q_ff);
    uint32_t* chunk_out = reinterpret_cast<uint32_t*>(dev2 + q_out);
    uint32_t* final_len = reinterpret_cast<uint32_t*>(dev2 + q_len);
    unsigned long long* final_off = reinterpret_cast<unsigned long long*>(dev2 + q_foff);
    // The finished files go straight into pinned host memory when it is ours (hipHostMalloc: mapped into the device's address
    // space): the expand kernel's stores cross PCIe themselves and no copy follows.  With a caller-supplied pinned allocator
    // the mapping is unknown, so the files are assembled in HBM and copied.
    static const bool no_direct = getenv("HIPJPEG_ENCODE_STAGED_OUTPUT") != nullptr;
    bool direct = !no_direct && henc_out_.reserve(arena_cap + 256) == HIPJPEG_STATUS_SUCCESS && !henc_out_.custom();
    uint8_t* arena = direct ? henc_out_.data() : dev2 + q_arena;
    if (hipMemcpyAsync(dev2, pin + p_up2, up2, hipMemcpyHostToDevice, s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
    if (launch_henc_zero(dev2 + q_raw, raw_total, stream_) != 0) return HIPJPEG_STATUS_HIP_ERROR;
    if (launch_henc_write(dimg2, dunits, (int)units.size(), dtables, block_off, block_bits, stream_) != 0) return HIPJPEG_STATUS_HIP_ERROR;
    if (launch_henc_count(dimg2, dchunks, (int)nchunks, chunk_ff, stream_) != 0) return HIPJPEG_STATUS_HIP_ERROR;
    if (launch_henc_layout(dimg2, ng, chunk_ff, chunk_out, final_len, final_off, stream_) != 0) return HIPJPEG_STATUS_HIP_ERROR;
    if (launch_henc_expand(dimg2, dchunks, (int)nchunks, chunk_out, final_len, final_off, arena, stream_) != 0) return HIPJPEG_STATUS_HIP_ERROR;
    uint32_t* h_len = reinterpret_cast<uint32_t*>(pin + p_len);
    unsigned long long* h_foff = reinterpret_cast<unsigned long long*>(pin + p_foff);
    if (hipMemcpyAsync(h_len, final_len, (size_t)ng * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(h_foff, final_off, (size_t)ng * 8, hipMemcpyDeviceToHost, s) != hipSuccess)
        return HIPJPEG_STATUS_HIP_ERROR;
    if (hipStreamSynchronize(s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
    const size_t used = (size_t)h_foff[ng - 1] + align_up((size_t)h_len[ng - 1], 16);
    if (used > arena_cap) return HIPJPEG_STATUS_HIP_ERROR;  // cannot happen: the capacity assumes every byte is stuffed
    if (!direct) {
        if ((st = henc_out_.reserve(used + 256)) != HIPJPEG_STATUS_SUCCESS) return st;
        if (hipMemcpyAsync(henc_out_.data(), arena, used, hipMemcpyDeviceToHost, s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
        if (hipStreamSynchronize(s) != hipSuccess) return HIPJPEG_STATUS_HIP_ERROR;
    }
    for (int g = 0; g < ng; g++) {
        PlannedEncode& im = images_[idx[g]];
        im.gpu_bitstream = henc_out_.data() + h_foff[g];
        im.gpu_bitstream_len = h_len[g];
    }
    gpu_entropy_images_ = (uint64_t)ng;
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
    opt.progressive = im.params.progressive != 0;
    im.bitstream.clear();
    im.gpu_bitstream = nullptr;
    im.gpu_bitstream_len = 0;
    encode_jfif(im.geom, im.qlum, im.qchr, coef, opt, &im.bitstream);
}

}  // namespace hipjpeg
