// entropy_encode.cpp -- see entropy_encode.h
#include "entropy_encode.h"

#include <algorithm>
#include <cstring>

#include "jpeg_syntax.h"

namespace hipjpeg {

namespace {

// ITU T.81 Annex K.1 (natural order)
const uint8_t kStdLumQ[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
                              69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55,  64,
                              81, 104, 113, 92, 49, 64, 78,  87,  103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kStdChrQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                              99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

// ITU T.81 Annex K.3
const uint8_t kDcLumBits[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChrBits[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumBits[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChrBits[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct HuffTable {
    uint8_t bits[17];
    uint8_t vals[256];
    int nvals;
    uint32_t code[256];
    uint8_t size[256];
    void set(const uint8_t* b, const uint8_t* v)
    {
        memcpy(bits, b, 17);
        nvals = 0;
        for (int i = 1; i <= 16; i++) nvals += bits[i];
        memcpy(vals, v, nvals);
        derive();
    }
    void derive()
    {
        memset(code, 0, sizeof code);
        memset(size, 0, sizeof size);
        uint32_t c = 0;
        int k = 0;
        for (int l = 1; l <= 16; l++) {
            for (int i = 0; i < bits[l]; i++, k++) {
                code[vals[k]] = c++;
                size[vals[k]] = (uint8_t)l;
            }
            c <<= 1;
        }
    }
};

// jchuff.c jpeg_gen_optimal_table: Huffman code lengths from symbol frequencies, limited to 16 bits, with the
// reserved all-ones code point (the extra symbol 256).
void gen_optimal_table(long freq_in[257], HuffTable* t)
{
    long freq[257];
    memcpy(freq, freq_in, sizeof freq);
    uint8_t bits[33];
    int codesize[257], others[257];
    memset(bits, 0, sizeof bits);
    memset(codesize, 0, sizeof codesize);
    for (int i = 0; i < 257; i++) others[i] = -1;
    freq[256] = 1;
    for (;;) {
        int c1 = -1, c2 = -1;
        long v = 1000000000L;
        for (int i = 0; i <= 256; i++)
            if (freq[i] && freq[i] <= v) {
                v = freq[i];
                c1 = i;
            }
        v = 1000000000L;
        for (int i = 0; i <= 256; i++)
            if (freq[i] && freq[i] <= v && i != c1) {
                v = freq[i];
                c2 = i;
            }
        if (c2 < 0) break;
        freq[c1] += freq[c2];
        freq[c2] = 0;
        codesize[c1]++;
        while (others[c1] >= 0) {
            c1 = others[c1];
            codesize[c1]++;
        }
        others[c1] = c2;
        codesize[c2]++;
        while (others[c2] >= 0) {
            c2 = others[c2];
            codesize[c2]++;
        }
    }
    for (int i = 0; i <= 256; i++)
        if (codesize[i]) bits[std::min(codesize[i], 32)]++;
    for (int i = 32; i > 16; i--) {
        while (bits[i] > 0) {
            int j = i - 2;
            while (bits[j] == 0) j--;
            bits[i] -= 2;
            bits[i - 1]++;
            bits[j + 1] += 2;
            bits[j]--;
        }
    }
    int i = 16;
    while (bits[i] == 0) i--;
    bits[i]--;  // remove the reserved symbol
    memcpy(t->bits, bits, 17);
    t->bits[0] = 0;
    int p = 0;
    for (int l = 1; l <= 32; l++)
        for (int s = 0; s < 256; s++)
            if (codesize[s] == l) t->vals[p++] = (uint8_t)s;
    t->nvals = p;
    t->derive();
}

class BitWriter {
public:
    explicit BitWriter(std::vector<uint8_t>* out) : out_(out) {}
    inline void put(uint32_t code, int size)
    {
        acc_ = (acc_ << size) | (code & ((1u << size) - 1));
        n_ += size;
        if (n_ >= 32) flush_words();
    }
    void flush_words()
    {
        while (n_ >= 8) {
            uint8_t b = (uint8_t)(acc_ >> (n_ - 8));
            out_->push_back(b);
            if (b == 0xFF) out_->push_back(0);
            n_ -= 8;
        }
    }
    // pad to a byte boundary with 1 bits (jchuff.c flush_bits)
    void align()
    {
        flush_words();
        if (n_ > 0) {
            put(0x7F, 8 - n_);
            flush_words();
        }
        acc_ = 0;
        n_ = 0;
    }
    void raw(uint8_t b) { out_->push_back(b); }

private:
    std::vector<uint8_t>* out_;
    uint64_t acc_ = 0;
    int n_ = 0;
};

inline int bit_length(int v)
{
    return v ? 32 - __builtin_clz((unsigned)v) : 0;
}

// One block: either emits codes (bw != null) or counts symbol frequencies (dc_freq/ac_freq != null).
inline void code_block(const int16_t* zz, int dc, int* pred, const HuffTable& dct, const HuffTable& act, BitWriter* bw, long* dc_freq,
                       long* ac_freq)
{
    int diff = dc - *pred;
    *pred = dc;
    int t = diff < 0 ? -diff : diff;
    int nb = bit_length(t);
    if (bw) {
        bw->put(dct.code[nb], dct.size[nb]);
        if (nb) bw->put((uint32_t)(diff < 0 ? diff - 1 : diff), nb);
    } else {
        dc_freq[nb]++;
    }
    if (!zz) {  // dummy block: AC all zero
        if (bw)
            bw->put(act.code[0], act.size[0]);
        else
            ac_freq[0]++;
        return;
    }
    int run = 0;
    for (int k = 1; k < 64; k++) {
        int v = zz[k];
        if (v == 0) {
            run++;
            continue;
        }
        while (run > 15) {
            if (bw)
                bw->put(act.code[0xF0], act.size[0xF0]);
            else
                ac_freq[0xF0]++;
            run -= 16;
        }
        int a = v < 0 ? -v : v;
        nb = bit_length(a);
        int sym = (run << 4) + nb;
        if (bw) {
            bw->put(act.code[sym], act.size[sym]);
            bw->put((uint32_t)(v < 0 ? v - 1 : v), nb);
        } else {
            ac_freq[sym]++;
        }
        run = 0;
    }
    if (run > 0) {
        if (bw)
            bw->put(act.code[0], act.size[0]);
        else
            ac_freq[0]++;
    }
}

struct BlockSource {
    const EncodeGeometry& g;
    const int16_t* const* coef;
    // DC value libjpeg gives a block: real blocks their own; dummy blocks the DC of the preceding block in MCU order
    // (jccoefct.c compress_data): right of the last real column -> left neighbour; below the last real row -> last block
    // of the previous block row of the same MCU.
    int dc_of(int c, int bx, int by) const
    {
        const int mh = (c == 0 && g.ncomp == 3) ? g.hs : 1;
        while (by >= g.real_h[c]) {
            bx = (bx / mh) * mh + mh - 1;
            by--;
        }
        if (bx >= g.real_w[c]) bx = g.real_w[c] - 1;
        return coef[c][((size_t)by * g.blocks_w[c] + bx) * 64];
    }
    const int16_t* block(int c, int bx, int by) const
    {
        if (bx < g.real_w[c] && by < g.real_h[c]) return coef[c] + ((size_t)by * g.blocks_w[c] + bx) * 64;
        return nullptr;
    }
};

void put16(std::vector<uint8_t>* o, int v)
{
    o->push_back((uint8_t)(v >> 8));
    o->push_back((uint8_t)v);
}

void write_dht(std::vector<uint8_t>* o, int tc_th, const HuffTable& t)
{
    put16(o, 0xFFC4);
    put16(o, 2 + 1 + 16 + t.nvals);
    o->push_back((uint8_t)tc_th);
    for (int i = 1; i <= 16; i++) o->push_back(t.bits[i]);
    o->insert(o->end(), t.vals, t.vals + t.nvals);
}

// Walks the scan in MCU order calling code_block for every block (real or dummy).
void walk_scan(const EncodeGeometry& g, const BlockSource& src, int restart_interval, const HuffTable* dct[3], const HuffTable* act[3],
               BitWriter* bw, long dc_freq[2][257], long ac_freq[2][257])
{
    int pred[3] = {0, 0, 0}, left = restart_interval, rst = 0;
    for (int my = 0; my < g.mcus_y; my++)
        for (int mx = 0; mx < g.mcus_x; mx++) {
            if (restart_interval && left == 0) {
                if (bw) {
                    bw->align();
                    bw->raw(0xFF);
                    bw->raw((uint8_t)(0xD0 + rst));
                }
                rst = (rst + 1) & 7;
                pred[0] = pred[1] = pred[2] = 0;
                left = restart_interval;
            }
            for (int c = 0; c < g.ncomp; c++) {
                const int mh = (c == 0 && g.ncomp == 3) ? g.hs : 1, mv = (c == 0 && g.ncomp == 3) ? g.vs : 1;
                const int ti = c == 0 ? 0 : 1;
                for (int v = 0; v < mv; v++)
                    for (int h = 0; h < mh; h++) {
                        const int bx = mx * mh + h, by = my * mv + v;
                        code_block(src.block(c, bx, by), src.dc_of(c, bx, by), &pred[c], *dct[c], *act[c], bw, bw ? nullptr : dc_freq[ti],
                                   bw ? nullptr : ac_freq[ti]);
                    }
            }
            left--;
        }
}

}  // namespace

void quality_tables(int quality, uint16_t lum[64], uint16_t chr[64])
{
    if (quality <= 0) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int i = 0; i < 64; i++) {
        long a = ((long)kStdLumQ[i] * scale + 50L) / 100L, b = ((long)kStdChrQ[i] * scale + 50L) / 100L;
        lum[i] = (uint16_t)std::min(255L, std::max(1L, a));
        chr[i] = (uint16_t)std::min(255L, std::max(1L, b));
    }
}

void compute_geometry(EncodeGeometry* g)
{
    if (g->ncomp == 1) g->hs = g->vs = 1;
    g->mcus_x = (g->width + 8 * g->hs - 1) / (8 * g->hs);
    g->mcus_y = (g->height + 8 * g->vs - 1) / (8 * g->vs);
    for (int c = 0; c < g->ncomp; c++) {
        const int h = c == 0 ? g->hs : 1, v = c == 0 ? g->vs : 1;
        g->blocks_w[c] = g->mcus_x * h;
        g->blocks_h[c] = g->mcus_y * v;
        const int sw = c == 0 ? g->width : (g->width + g->hs - 1) / g->hs, sh = c == 0 ? g->height : (g->height + g->vs - 1) / g->vs;
        g->real_w[c] = (sw + 7) / 8;
        g->real_h[c] = (sh + 7) / 8;
    }
}

// SOI, APP0, DQT.., SOFn (jcmarker.c write_file_header + write_frame_header)
static void write_frame_header(const EncodeGeometry& g, const uint16_t qlum[64], const uint16_t qchr[64], int sof_marker, std::vector<uint8_t>* o)
{
    put16(o, 0xFFD8);
    put16(o, 0xFFE0);
    put16(o, 16);
    const uint8_t jfif[14] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    o->insert(o->end(), jfif, jfif + 14);
    for (int t = 0; t < (g.ncomp == 3 ? 2 : 1); t++) {
        put16(o, 0xFFDB);
        put16(o, 67);
        o->push_back((uint8_t)t);
        for (int i = 0; i < 64; i++) o->push_back((uint8_t)(t ? qchr : qlum)[kZigzagNatural[i]]);
    }
    put16(o, sof_marker);
    put16(o, 8 + 3 * g.ncomp);
    o->push_back(8);
    put16(o, g.height);
    put16(o, g.width);
    o->push_back((uint8_t)g.ncomp);
    for (int c = 0; c < g.ncomp; c++) {
        o->push_back((uint8_t)(c + 1));
        o->push_back((uint8_t)(c == 0 ? ((g.hs << 4) | g.vs) : 0x11));
        o->push_back((uint8_t)(c == 0 ? 0 : 1));
    }
}

// Everything in front of the entropy-coded data: SOI .. SOS, in jcmarker.c's order.
static void write_headers(const EncodeGeometry& g, const uint16_t qlum[64], const uint16_t qchr[64], const HuffTable& dcl, const HuffTable& acl,
                          const HuffTable& dcc, const HuffTable& acc, int restart_interval, std::vector<uint8_t>* o)
{
    // marker order of jcmarker.c: SOI, APP0, DQT.., SOF0, DHT.., [DRI], SOS
    write_frame_header(g, qlum, qchr, 0xFFC0, o);
    write_dht(o, 0x00, dcl);
    write_dht(o, 0x10, acl);
    if (g.ncomp == 3) {
        write_dht(o, 0x01, dcc);
        write_dht(o, 0x11, acc);
    }
    if (restart_interval) {
        put16(o, 0xFFDD);
        put16(o, 4);
        put16(o, restart_interval);
    }
    put16(o, 0xFFDA);
    put16(o, 6 + 2 * g.ncomp);
    o->push_back((uint8_t)g.ncomp);
    for (int c = 0; c < g.ncomp; c++) {
        o->push_back((uint8_t)(c + 1));
        o->push_back((uint8_t)(c == 0 ? 0x00 : 0x11));
    }
    o->push_back(0);
    o->push_back(63);
    o->push_back(0);
}

namespace {

// ---- progressive (SOF2) ------------------------------------------------------------------------------------------------
// libjpeg's progressive Huffman coder (jcphuff.c) over the scan script of jcparam.c jpeg_simple_progression, with the
// per-scan optimal tables libjpeg forces in progressive mode (jcmaster.c: optimize_coding = TRUE).

struct ScanSpec {
    int ncomp, comp[3];
    int ss, se, ah, al;
};

// jcparam.c jpeg_simple_progression: the 10-scan script for YCbCr, the all-purpose 6-scan script for one component
std::vector<ScanSpec> simple_progression(int ncomp)
{
    std::vector<ScanSpec> v;
    auto dc = [&](int ah, int al) {
        ScanSpec s{ncomp, {0, 1, 2}, 0, 0, ah, al};
        v.push_back(s);
    };
    auto ac = [&](int c, int ss, int se, int ah, int al) {
        ScanSpec s{1, {c, 0, 0}, ss, se, ah, al};
        v.push_back(s);
    };
    if (ncomp == 3) {
        dc(0, 1);
        ac(0, 1, 5, 0, 2);
        ac(2, 1, 63, 0, 1);
        ac(1, 1, 63, 0, 1);
        ac(0, 6, 63, 0, 2);
        ac(0, 1, 63, 2, 1);
        dc(1, 0);
        ac(2, 1, 63, 1, 0);
        ac(1, 1, 63, 1, 0);
        ac(0, 1, 63, 1, 0);
    } else {
        dc(0, 1);
        ac(0, 1, 5, 0, 2);
        ac(0, 6, 63, 0, 2);
        ac(0, 1, 63, 2, 1);
        dc(1, 0);
        ac(0, 1, 63, 1, 0);
    }
    return v;
}

// One scan, either counting symbols (bw == null) or emitting them.  State and routine names follow jcphuff.c.
class ProgressiveScanCoder {
public:
    ProgressiveScanCoder(const EncodeGeometry& g, const BlockSource& src, const ScanSpec& sc, int restart_interval, BitWriter* bw,
                         const HuffTable* tables, long (*counts)[257])
        : g_(g), src_(src), sc_(sc), bw_(bw), tables_(tables), counts_(counts), interval_(restart_interval), left_(restart_interval)
    {
    }

    void run()
    {
        if (sc_.ss == 0) {
            if (sc_.ncomp > 1) {
                for (int my = 0; my < g_.mcus_y; my++)
                    for (int mx = 0; mx < g_.mcus_x; mx++) {
                        next_mcu();
                        for (int k = 0; k < sc_.ncomp; k++) {
                            const int c = sc_.comp[k];
                            const int mh = c == 0 ? g_.hs : 1, mv = c == 0 ? g_.vs : 1;
                            for (int v = 0; v < mv; v++)
                                for (int h = 0; h < mh; h++) dc_block(c, mx * mh + h, my * mv + v);
                        }
                    }
            } else {
                const int c = sc_.comp[0];
                for (int by = 0; by < g_.real_h[c]; by++)
                    for (int bx = 0; bx < g_.real_w[c]; bx++) {
                        next_mcu();
                        dc_block(c, bx, by);
                    }
            }
        } else {
            // AC scans hold one component: its real blocks in raster order (jcmaster.c per_scan_setup, non-interleaved)
            const int c = sc_.comp[0];
            table_ = c == 0 ? 0 : 1;
            for (int by = 0; by < g_.real_h[c]; by++)
                for (int bx = 0; bx < g_.real_w[c]; bx++) {
                    next_mcu();
                    const int16_t* zz = src_.block(c, bx, by);
                    if (sc_.ah == 0)
                        ac_first(zz);
                    else
                        ac_refine(zz);
                }
            emit_eobrun();  // finish_pass_phuff
        }
        if (bw_) bw_->align();
    }

private:
    // jcphuff.c emit_restart, in front of the MCU that starts a new interval
    void next_mcu()
    {
        if (!interval_) return;
        if (left_ == 0) {
            emit_eobrun();
            if (bw_) {
                bw_->align();
                bw_->raw(0xFF);
                bw_->raw((uint8_t)(0xD0 + rst_));
            }
            rst_ = (rst_ + 1) & 7;
            pred_[0] = pred_[1] = pred_[2] = 0;
            eobrun_ = 0;
            be_ = 0;
            left_ = interval_;
        }
        left_--;
    }
    void emit_symbol(int table, int sym)
    {
        if (bw_)
            bw_->put(tables_[table].code[sym], tables_[table].size[sym]);
        else
            counts_[table][sym]++;
    }
    void emit_bits(uint32_t v, int n)
    {
        if (bw_ && n) bw_->put(v, n);
    }
    void emit_buffered_bits(const uint8_t* p, unsigned n)
    {
        if (!bw_) return;
        for (unsigned i = 0; i < n; i++) bw_->put(p[i], 1);
    }
    void emit_eobrun()
    {
        if (eobrun_ > 0) {
            int nbits = 0;
            for (unsigned t = eobrun_; (t >>= 1) != 0;) nbits++;
            emit_symbol(table_, nbits << 4);
            if (nbits) emit_bits(eobrun_, nbits);
            eobrun_ = 0;
            emit_buffered_bits(corr_, be_);
            be_ = 0;
        }
    }
    void dc_block(int c, int bx, int by)
    {
        const int v = src_.dc_of(c, bx, by);
        if (sc_.ah != 0) {  // encode_mcu_DC_refine: the next lower bit, no table
            emit_bits((uint32_t)(v >> sc_.al) & 1u, 1);
            return;
        }
        const int t2 = v >> sc_.al;  // arithmetic shift (IRIGHT_SHIFT)
        int diff = t2 - pred_[c];
        pred_[c] = t2;
        const int a = diff < 0 ? -diff : diff;
        const int nb = bit_length(a);
        emit_symbol(c == 0 ? 0 : 1, nb);
        if (nb) emit_bits((uint32_t)(diff < 0 ? diff - 1 : diff), nb);
    }
    void ac_first(const int16_t* zz)
    {
        int r = 0;
        for (int k = sc_.ss; k <= sc_.se; k++) {
            int t = zz[k], t2;
            if (t == 0) {
                r++;
                continue;
            }
            if (t < 0) {
                t = (-t) >> sc_.al;
                t2 = ~t;
            } else {
                t >>= sc_.al;
                t2 = t;
            }
            if (t == 0) {
                r++;
                continue;
            }
            if (eobrun_ > 0) emit_eobrun();
            while (r > 15) {
                emit_symbol(table_, 0xF0);
                r -= 16;
            }
            const int nb = bit_length(t);
            emit_symbol(table_, (r << 4) + nb);
            emit_bits((uint32_t)t2, nb);
            r = 0;
        }
        if (r > 0) {
            eobrun_++;
            if (eobrun_ == 0x7FFF) emit_eobrun();
        }
    }
    void ac_refine(const int16_t* zz)
    {
        int absv[64], eob = 0;
        for (int k = sc_.ss; k <= sc_.se; k++) {
            int t = zz[k];
            if (t < 0) t = -t;
            t >>= sc_.al;
            absv[k] = t;
            if (t == 1) eob = k;
        }
        int r = 0;
        unsigned br = 0;
        uint8_t* br_buffer = corr_ + be_;
        for (int k = sc_.ss; k <= sc_.se; k++) {
            const int t = absv[k];
            if (t == 0) {
                r++;
                continue;
            }
            while (r > 15 && k <= eob) {
                emit_eobrun();
                emit_symbol(table_, 0xF0);
                r -= 16;
                emit_buffered_bits(br_buffer, br);
                br_buffer = corr_;
                br = 0;
            }
            if (t > 1) {  // already nonzero: its next bit goes behind the coming symbol
                br_buffer[br++] = (uint8_t)(t & 1);
                continue;
            }
            emit_eobrun();
            emit_symbol(table_, (r << 4) + 1);
            emit_bits(zz[k] < 0 ? 0u : 1u, 1);
            emit_buffered_bits(br_buffer, br);
            br_buffer = corr_;
            br = 0;
            r = 0;
        }
        if (r > 0 || br > 0) {
            eobrun_++;
            be_ += br;
            if (eobrun_ == 0x7FFF || be_ > (kMaxCorrBits - 64 + 1)) emit_eobrun();
        }
    }

    static constexpr unsigned kMaxCorrBits = 1000;  // jcphuff.c MAX_CORR_BITS
    const EncodeGeometry& g_;
    const BlockSource& src_;
    const ScanSpec& sc_;
    BitWriter* bw_;
    const HuffTable* tables_;
    long (*counts_)[257];
    int interval_, left_, rst_ = 0;
    int pred_[3] = {0, 0, 0};
    int table_ = 0;
    unsigned eobrun_ = 0, be_ = 0;
    uint8_t corr_[kMaxCorrBits];
};

void encode_progressive(const EncodeGeometry& g, const uint16_t qlum[64], const uint16_t qchr[64], const int16_t* const coef[3],
                        int restart_interval, std::vector<uint8_t>* o)
{
    bool dri_sent = false;
    BlockSource src{g, coef};
    write_frame_header(g, qlum, qchr, 0xFFC2, o);
    for (const ScanSpec& sc : simple_progression(g.ncomp)) {
        const bool dc_scan = sc.ss == 0;
        const bool needs_table = !(dc_scan && sc.ah != 0);
        HuffTable tables[2];
        bool used[2] = {false, false};
        if (needs_table) {
            long counts[2][257];
            memset(counts, 0, sizeof counts);
            ProgressiveScanCoder(g, src, sc, restart_interval, nullptr, nullptr, counts).run();
            for (int k = 0; k < sc.ncomp; k++) used[sc.comp[k] == 0 ? 0 : 1] = true;
            for (int t = 0; t < 2; t++)
                if (used[t]) gen_optimal_table(counts[t], &tables[t]);
            // jcmarker.c write_scan_header: the tables this scan uses, in component order, each once
            bool sent[2] = {false, false};
            for (int k = 0; k < sc.ncomp; k++) {
                const int t = sc.comp[k] == 0 ? 0 : 1;
                if (sent[t]) continue;
                sent[t] = true;
                write_dht(o, (dc_scan ? 0x00 : 0x10) | t, tables[t]);
            }
        }
        if (restart_interval && !dri_sent) {  // write_scan_header: DRI whenever the interval changes, i.e. once
            put16(o, 0xFFDD);
            put16(o, 4);
            put16(o, restart_interval);
            dri_sent = true;
        }
        put16(o, 0xFFDA);
        put16(o, 6 + 2 * sc.ncomp);
        o->push_back((uint8_t)sc.ncomp);
        for (int k = 0; k < sc.ncomp; k++) {
            const int c = sc.comp[k], t = c == 0 ? 0 : 1;
            o->push_back((uint8_t)(c + 1));
            // only the table a scan uses is named: DC first scans Td, AC scans Ta, DC refinement neither
            o->push_back((uint8_t)(dc_scan ? (sc.ah == 0 ? t << 4 : 0) : t));
        }
        o->push_back((uint8_t)sc.ss);
        o->push_back((uint8_t)sc.se);
        o->push_back((uint8_t)((sc.ah << 4) | sc.al));
        BitWriter bw(o);
        ProgressiveScanCoder(g, src, sc, restart_interval, &bw, tables, nullptr).run();
    }
    put16(o, 0xFFD9);
}

}  // namespace

void write_standard_headers(const EncodeGeometry& g, const uint16_t qlum[64], const uint16_t qchr[64], std::vector<uint8_t>* out)
{
    HuffTable dcl, dcc, acl, acc;
    dcl.set(kDcLumBits, kDcVals);
    dcc.set(kDcChrBits, kDcVals);
    acl.set(kAcLumBits, kAcLumVals);
    acc.set(kAcChrBits, kAcChrVals);
    write_headers(g, qlum, qchr, dcl, acl, dcc, acc, 0, out);
}

void standard_code_tables(StandardCodeTables* t)
{
    HuffTable dc[2], ac[2];
    dc[0].set(kDcLumBits, kDcVals);
    dc[1].set(kDcChrBits, kDcVals);
    ac[0].set(kAcLumBits, kAcLumVals);
    ac[1].set(kAcChrBits, kAcChrVals);
    for (int k = 0; k < 2; k++) {
        for (int i = 0; i < 16; i++) {
            t->dc_code[k][i] = (uint16_t)dc[k].code[i];
            t->dc_size[k][i] = dc[k].size[i];
        }
        for (int i = 0; i < 256; i++) {
            t->ac_code[k][i] = (uint16_t)ac[k].code[i];
            t->ac_size[k][i] = ac[k].size[i];
        }
    }
}

void optimal_code_tables(const uint32_t counts[2][2][256], const EncodeGeometry& g, const uint16_t qlum[64], const uint16_t qchr[64],
                         StandardCodeTables* t, std::vector<uint8_t>* headers)
{
    HuffTable dc[2], ac[2];
    dc[0].set(kDcLumBits, kDcVals);
    dc[1].set(kDcChrBits, kDcVals);
    ac[0].set(kAcLumBits, kAcLumVals);
    ac[1].set(kAcChrBits, kAcChrVals);
    const int ntab = g.ncomp == 3 ? 2 : 1;
    for (int k = 0; k < ntab; k++) {
        long freq[257];
        memset(freq, 0, sizeof freq);
        for (int i = 0; i < 256; i++) freq[i] = (long)counts[k][0][i];
        gen_optimal_table(freq, &dc[k]);
        memset(freq, 0, sizeof freq);
        for (int i = 0; i < 256; i++) freq[i] = (long)counts[k][1][i];
        gen_optimal_table(freq, &ac[k]);
    }
    for (int k = 0; k < 2; k++) {
        for (int i = 0; i < 16; i++) {
            t->dc_code[k][i] = (uint16_t)dc[k].code[i];
            t->dc_size[k][i] = dc[k].size[i];
        }
        for (int i = 0; i < 256; i++) {
            t->ac_code[k][i] = (uint16_t)ac[k].code[i];
            t->ac_size[k][i] = ac[k].size[i];
        }
    }
    write_headers(g, qlum, qchr, dc[0], ac[0], dc[1], ac[1], 0, headers);
}

void encode_jfif(const EncodeGeometry& g, const uint16_t qlum[64], const uint16_t qchr[64], const int16_t* const coef[3],
                 const EntropyEncodeOptions& opt, std::vector<uint8_t>* o)
{
    if (opt.progressive) {
        o->reserve(o->size() + (size_t)g.width * g.height / 2 + 1024);
        encode_progressive(g, qlum, qchr, coef, opt.restart_interval, o);
        return;
    }
    HuffTable dcl, dcc, acl, acc;
    dcl.set(kDcLumBits, kDcVals);
    dcc.set(kDcChrBits, kDcVals);
    acl.set(kAcLumBits, kAcLumVals);
    acc.set(kAcChrBits, kAcChrVals);
    const HuffTable* dct[3] = {&dcl, &dcc, &dcc};
    const HuffTable* act[3] = {&acl, &acc, &acc};
    BlockSource src{g, coef};
    if (opt.optimized_huffman) {
        long dc_freq[2][257], ac_freq[2][257];
        memset(dc_freq, 0, sizeof dc_freq);
        memset(ac_freq, 0, sizeof ac_freq);
        walk_scan(g, src, opt.restart_interval, dct, act, nullptr, dc_freq, ac_freq);
        gen_optimal_table(dc_freq[0], &dcl);
        gen_optimal_table(ac_freq[0], &acl);
        if (g.ncomp == 3) {
            gen_optimal_table(dc_freq[1], &dcc);
            gen_optimal_table(ac_freq[1], &acc);
        }
    }
    o->reserve(o->size() + (size_t)g.width * g.height / 2 + 1024);
    write_headers(g, qlum, qchr, dcl, acl, dcc, acc, opt.restart_interval, o);
    BitWriter bw(o);
    walk_scan(g, src, opt.restart_interval, dct, act, &bw, nullptr, nullptr);
    bw.align();
    put16(o, 0xFFD9);
}

}  // namespace hipjpeg
