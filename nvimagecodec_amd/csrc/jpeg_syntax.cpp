// jpeg_syntax.cpp -- see jpeg_syntax.h
#include "jpeg_syntax.h"

#include <algorithm>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <cstring>

namespace hipjpeg {

const uint8_t kZigzagNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                    15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

namespace {

inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// Finds the end of an entropy-coded segment: first 0xFF followed by something other than 0x00, 0xFF or RSTn.
// *plain = false when the segment holds fill bytes (FF FF), a lone FF at the end of the input, or restart markers that do not
// count RST0, RST1, ... RST7, RST0 ...; rst_after (optional) receives, for every RSTn marker, the offset of the byte behind it in
// the DESTUFFED segment (stuffed zeros and the markers themselves removed).
// The walk through the entropy-coded data is the bulk of the parser's time: half a megabyte per 1080p image with an FF every
// hundred bytes (entropy-coded data leans towards one-bits).  Where the CPU has AVX2 (checked once) the data is taken 32 bytes at
// a time and a chunk that holds nothing but stuffed FFs (FF 00) costs two compares and a population count, without a
// data-dependent branch -- a memchr call per FF mispredicts once per FF.
struct ScanWalk {
    size_t begin, stuffed = 0, markers = 0;
    bool* plain;
    std::vector<uint32_t>* rst_after;
    std::vector<uint32_t>* drops = nullptr;  // ScanHeader::chunk_drops
    inline void drop(size_t index, uint32_t count = 1)
    {
        const size_t c = (index - begin) / kScanChunkBytes;
        if (c >= drops->size()) drops->resize(c + 1, 0u);
        (*drops)[c] += count;
    }
    // looks at the FF at offset i (i + 1 < size): false = the scan ends here
    inline bool visit(const uint8_t* data, size_t i)
    {
        const uint8_t m = data[i + 1];
        if (m == 0x00) {
            stuffed++;
            if (drops) drop(i + 1);
        } else if (m >= 0xD0 && m <= 0xD7) {
            if ((unsigned)(m - 0xD0) != (markers & 7)) *plain = false;
            markers++;
            if (drops) {
                drop(i);
                drop(i + 1);
            }
            if (rst_after) rst_after->push_back((uint32_t)(i + 2 - begin - stuffed - 2 * markers));
        } else if (m == 0xFF) {
            *plain = false;  // fill byte; the next FF is looked at in its turn
        } else {
            return false;
        }
        return true;
    }
};

size_t find_scan_end_plain(const uint8_t* data, size_t pos, size_t size, ScanWalk& w)
{
    while (pos < size) {
        const uint8_t* ff = static_cast<const uint8_t*>(memchr(data + pos, 0xFF, size - pos));
        if (!ff) return size;
        const size_t i = ff - data;
        if (i + 1 >= size) {
            *w.plain = false;
            return size;
        }
        if (!w.visit(data, i)) return i;
        pos = i + 1;  // behind FF 00 / FF RSTn the second byte is no FF: looking at it again costs nothing
    }
    return size;
}

#if defined(__x86_64__)
__attribute__((target("avx2,popcnt"))) size_t find_scan_end_avx2(const uint8_t* data, size_t pos, size_t size, ScanWalk& w)
{
    const __m256i ff = _mm256_set1_epi8((char)0xFF), zero = _mm256_setzero_si256();
    while (pos + 33 <= size) {
        const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(data + pos));
        const __m256i y = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(data + pos + 1));  // the byte behind each byte
        uint32_t mask = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, ff));
        const uint32_t stuffing = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(y, zero));
        if ((mask & ~stuffing) == 0) {  // nothing but FF 00 in this chunk: no data-dependent branch per FF
            w.stuffed += (size_t)__builtin_popcount(mask);
            if (w.drops && mask) {
                // the dropped bytes are the ones behind the FFs: offsets pos + 1 .. pos + 32, in one or two counting chunks
                const size_t first = (pos + 1 - w.begin) / kScanChunkBytes, last = (pos + 32 - w.begin) / kScanChunkBytes;
                if (first == last) {
                    w.drop(pos + 1, (uint32_t)__builtin_popcount(mask));
                } else {
                    const size_t in_first = last * kScanChunkBytes + w.begin - (pos + 1);  // FFs at pos .. pos + in_first - 1 drop into `first`
                    const uint32_t low = mask & (uint32_t)((1ull << in_first) - 1);
                    if (low) w.drop(pos + 1, (uint32_t)__builtin_popcount(low));
                    if (mask & ~low) w.drop(pos + 32, (uint32_t)__builtin_popcount(mask & ~low));
                }
            }
            pos += 32;
            continue;
        }
        while (mask) {  // a marker, a fill byte or the end of the scan: the chunk's FFs one by one
            const size_t i = pos + (size_t)__builtin_ctz(mask);
            mask &= mask - 1;
            if (!w.visit(data, i)) return i;
        }
        pos += 32;
    }
    return find_scan_end_plain(data, pos, size, w);
}
#endif

size_t find_scan_end(const uint8_t* data, size_t pos, size_t size, bool* plain, std::vector<uint32_t>* rst_after, std::vector<uint32_t>* chunk_drops)
{
    *plain = true;
    ScanWalk w;
    w.begin = pos;
    w.plain = plain;
    w.rst_after = rst_after;
    w.drops = chunk_drops;
    size_t end;
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2)
        end = find_scan_end_avx2(data, pos, size, w);
    else
#endif
        end = find_scan_end_plain(data, pos, size, w);
    if (chunk_drops) chunk_drops->resize((end - pos + kScanChunkBytes - 1) / kScanChunkBytes, 0u);  // chunks without an FF count zero
    return end;
}

ParseStatus finish_frame(FrameInfo* f)
{
    f->hmax = f->vmax = 1;
    for (int c = 0; c < f->ncomp; c++) {
        Component& k = f->comp[c];
        if (k.h < 1 || k.h > 4 || k.v < 1 || k.v > 4) return kParseBadStream;
        if (k.h > f->hmax) f->hmax = k.h;
        if (k.v > f->vmax) f->vmax = k.v;
    }
    f->mcus_x = (f->width + 8 * f->hmax - 1) / (8 * f->hmax);
    f->mcus_y = (f->height + 8 * f->vmax - 1) / (8 * f->vmax);
    for (int c = 0; c < f->ncomp; c++) {
        Component& k = f->comp[c];
        k.blocks_w = f->mcus_x * k.h;
        k.blocks_h = f->mcus_y * k.v;
        k.samp_w = (f->width * k.h + f->hmax - 1) / f->hmax;
        k.samp_h = (f->height * k.v + f->vmax - 1) / f->vmax;
    }
    // Same colour-model inference libjpeg applies (the CPU plugin inherits it through jpeg_read_header,
    // extensions/libjpeg_turbo/jpeg_mem.cpp:147): JFIF => YCbCr, Adobe transform flag, else component ids.
    if (f->ncomp == 1) {
        f->color = ColorModel::Gray;
    } else if (f->ncomp == 3) {
        if (f->saw_jfif)
            f->color = ColorModel::YCbCr;
        else if (f->saw_adobe)
            f->color = f->adobe_transform == 0 ? ColorModel::RGB : ColorModel::YCbCr;
        else if (f->comp[0].id == 'R' && f->comp[1].id == 'G' && f->comp[2].id == 'B')
            f->color = ColorModel::RGB;
        else
            f->color = ColorModel::YCbCr;
    } else if (f->ncomp == 4) {
        f->color = (f->saw_adobe && f->adobe_transform == 2) ? ColorModel::YCCK : ColorModel::CMYK;
    } else {
        return kParseUnsupported;
    }
    return kParseOk;
}

}  // namespace

ParseStatus parse_jpeg(const uint8_t* data, size_t size, FrameInfo* f, bool headers_only)
{
    *f = FrameInfo();
    if (!data || size < 4 || data[0] != 0xFF || data[1] != 0xD8) return kParseBadStream;

    HuffSpec dc[4], ac[4];
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false}, qt16[4] = {false, false, false, false};
    bool comp_q_latched[4] = {false, false, false, false};
    int restart_interval = 0;
    bool got_sof = false;
    memset(qt, 0, sizeof qt);

    size_t pos = 2;
    for (;;) {
        // locate next marker
        while (pos < size && data[pos] != 0xFF) pos++;
        while (pos < size && data[pos] == 0xFF) pos++;
        if (pos >= size) break;
        int m = data[pos++];
        if (m == 0xD9) break;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (pos + 2 > size) return kParseTruncated;
        int L = be16(data + pos);
        if (L < 2 || pos + (size_t)L > size) return kParseTruncated;
        const uint8_t* seg = data + pos + 2;
        const uint8_t* seg_end = data + pos + L;
        switch (m) {
        case 0xE0:
            if (L >= 7 && !memcmp(seg, "JFIF", 5)) f->saw_jfif = true;
            break;
        case 0xEE:
            if (L >= 14 && !memcmp(seg, "Adobe", 5)) {
                f->saw_adobe = true;
                f->adobe_transform = seg[11];
            }
            break;
        case 0xDB: {
            const uint8_t* q = seg;
            while (q < seg_end) {
                int pq = *q >> 4, tq = *q & 15;
                q++;
                if (tq > 3 || pq > 1 || q + 64 * (pq + 1) > seg_end) return kParseBadStream;
                for (int i = 0; i < 64; i++) qt[tq][kZigzagNatural[i]] = (uint16_t)(pq ? be16(q + 2 * i) : q[i]);
                qt_present[tq] = true;
                qt16[tq] = pq != 0;
                q += 64 * (pq + 1);
            }
            break;
        }
        case 0xC4: {
            const uint8_t* q = seg;
            while (q < seg_end) {
                int tc = *q >> 4, th = *q & 15;
                q++;
                if (tc > 1 || th > 3 || q + 16 > seg_end) return kParseBadStream;
                HuffSpec& t = tc ? ac[th] : dc[th];
                t = HuffSpec();
                int n = 0;
                for (int i = 1; i <= 16; i++) {
                    t.bits[i] = q[i - 1];
                    n += q[i - 1];
                }
                q += 16;
                if (n > 256 || q + n > seg_end) return kParseBadStream;
                memcpy(t.vals, q, n);
                q += n;
                // jdhuff.c jpeg_make_d_derived_tbl: after the codes of length l "code" must still fit in l bits -- the code space may
                // not be over-subscribed, and not filled either: no code word is all ones (a complete code would let a decoder parse
                // the one-bits that pad a stream's end as symbols, for ever)
                int code = 0;
                for (int l = 1; l <= 16; l++) {
                    code += t.bits[l];
                    if (code >= (1 << l) && n > 0) return kParseBadStream;
                    code <<= 1;
                }
                t.present = true;
            }
            break;
        }
        case 0xDD:
            if (L != 4) return kParseBadStream;
            restart_interval = be16(seg);
            break;
        case 0xC0:
        case 0xC1:
        case 0xC2: {
            if (got_sof || L < 8) return kParseBadStream;
            f->sof = m;
            f->precision = seg[0];
            f->height = be16(seg + 1);
            f->width = be16(seg + 3);
            f->ncomp = seg[5];
            if (f->precision != 8) return kParseUnsupported;
            if (f->width == 0 || f->height == 0) return kParseUnsupported;  // DNL-defined height: not handled
            if (f->ncomp < 1 || f->ncomp > 4 || L != 8 + 3 * f->ncomp) return kParseBadStream;
            for (int c = 0; c < f->ncomp; c++) {
                f->comp[c].id = seg[6 + 3 * c];
                f->comp[c].h = seg[7 + 3 * c] >> 4;
                f->comp[c].v = seg[7 + 3 * c] & 15;
                f->comp[c].tq = seg[8 + 3 * c];
                if (f->comp[c].tq > 3) return kParseBadStream;
            }
            ParseStatus st = finish_frame(f);
            if (st != kParseOk) return st;
            got_sof = true;
            break;
        }
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            // lossless / hierarchical / arithmetic: a different decoder's job
            f->sof = m;
            return kParseUnsupported;
        case 0xDA: {
            if (!got_sof) return kParseBadStream;
            if (headers_only) return kParseOk;
            if (L < 8) return kParseBadStream;  // shortest SOS: one component; also keeps seg[0] inside the segment
            ScanHeader sc;
            sc.ncomp = seg[0];
            if (sc.ncomp < 1 || sc.ncomp > 4 || L != 6 + 2 * sc.ncomp) return kParseBadStream;
            for (int i = 0; i < sc.ncomp; i++) {
                int cid = seg[1 + 2 * i], found = -1;
                for (int c = 0; c < f->ncomp; c++)
                    if (f->comp[c].id == cid) found = c;
                if (found < 0) return kParseBadStream;
                for (int j = 0; j < i; j++)
                    if (sc.comp_index[j] == found) return kParseBadStream;
                sc.comp_index[i] = found;
                sc.td[i] = seg[2 + 2 * i] >> 4;
                sc.ta[i] = seg[2 + 2 * i] & 15;
                if (sc.td[i] > 3 || sc.ta[i] > 3) return kParseBadStream;
                if (!comp_q_latched[found]) {
                    // jdinput.c latch_quant_tables: the table in force at the component's first scan sticks
                    int tq = f->comp[found].tq;
                    if (!qt_present[tq]) return kParseBadStream;
                    memcpy(f->qtab[found], qt[tq], sizeof qt[tq]);
                    f->qtab_16bit[found] = qt16[tq];
                    comp_q_latched[found] = true;
                }
            }
            sc.ss = seg[1 + 2 * sc.ncomp];
            sc.se = seg[2 + 2 * sc.ncomp];
            sc.ah = seg[3 + 2 * sc.ncomp] >> 4;
            sc.al = seg[3 + 2 * sc.ncomp] & 15;
            if (!f->progressive()) {
                sc.ss = 0;
                sc.se = 63;
                sc.ah = sc.al = 0;
            } else {
                if (sc.ss > sc.se || sc.se > 63 || sc.al > 13 || sc.ah > 13) return kParseBadStream;
                if (sc.ss == 0 && sc.se != 0) return kParseBadStream;       // DC scans carry DC only
                if (sc.ss != 0 && sc.ncomp != 1) return kParseBadStream;    // AC scans are single-component
                if (sc.ah != 0 && sc.ah != sc.al + 1) return kParseBadStream;
            }
            // MCU-interleaved scans may not exceed 10 blocks per MCU (T.81 B.2.3)
            if (sc.ncomp > 1) {
                int nb = 0;
                for (int i = 0; i < sc.ncomp; i++) nb += f->comp[sc.comp_index[i]].h * f->comp[sc.comp_index[i]].v;
                if (nb > 10) return kParseBadStream;
            }
            for (int i = 0; i < 4; i++) {
                sc.dc[i] = dc[i];
                sc.ac[i] = ac[i];
            }
            sc.restart_interval = restart_interval;
            sc.data_begin = pos + L;
            sc.data_end = find_scan_end(data, sc.data_begin, size, &sc.plain_stuffing, headers_only ? nullptr : &sc.rst_after,
                                        headers_only ? nullptr : &sc.chunk_drops);
            f->scans.push_back(sc);
            pos = sc.data_end;
            continue;
        }
        default:
            break;
        }
        pos += L;
    }
    if (!got_sof) return kParseBadStream;
    if (!headers_only && f->scans.empty()) return kParseBadStream;
    return kParseOk;
}

// The sampling layout as the API names it (the reference's parser does the same from the SOF factors, src/parsers/jpeg.cpp:262-330).
hipjpegChromaSubsampling_t classify_subsampling(const FrameInfo& f)
{
    if (f.ncomp == 1) return HIPJPEG_CSS_GRAY;
    if (f.ncomp != 3) return HIPJPEG_CSS_UNKNOWN;
    int yh = f.comp[0].h, yv = f.comp[0].v, uh = f.comp[1].h, uv = f.comp[1].v, vh = f.comp[2].h, vv = f.comp[2].v;
    int minh = std::min(yh, std::min(uh, vh)), minv = std::min(yv, std::min(uv, vv));
    if (minh == 0 || minv == 0) return HIPJPEG_CSS_UNKNOWN;
    if (yh % minh || uh % minh || vh % minh || yv % minv || uv % minv || vv % minv) return HIPJPEG_CSS_UNKNOWN;
    yh /= minh; uh /= minh; vh /= minh;
    yv /= minv; uv /= minv; vv /= minv;
    if (uh != vh || uv != vv || uh != 1 || uv != 1) return HIPJPEG_CSS_UNKNOWN;
    if (yh == 1 && yv == 1) return HIPJPEG_CSS_444;
    if (yh == 2 && yv == 1) return HIPJPEG_CSS_422;
    if (yh == 2 && yv == 2) return HIPJPEG_CSS_420;
    if (yh == 1 && yv == 2) return HIPJPEG_CSS_440;
    if (yh == 4 && yv == 1) return HIPJPEG_CSS_411;
    if (yh == 4 && yv == 2) return HIPJPEG_CSS_410;
    if (yh == 2 && yv == 4) return HIPJPEG_CSS_410V;
    return HIPJPEG_CSS_UNKNOWN;
}


}  // namespace hipjpeg
