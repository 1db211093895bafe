// gpu_huffman_host.cpp -- see gpu_huffman_host.h
#include "gpu_huffman_host.h"

#include <cstring>

namespace hipjpeg {

bool gpu_entropy_eligible(const FrameInfo& f)
{
    if (f.progressive() || f.scans.size() != 1) return false;
    const ScanHeader& sc = f.scans[0];
    if (sc.ncomp != f.ncomp || sc.restart_interval != 0) return false;
    int bpm = 0;
    for (int i = 0; i < sc.ncomp; i++) {
        if (sc.comp_index[i] != i) return false;  // keep the MCU layout simple: components in frame order
        if (!sc.dc[sc.td[i]].present || !sc.ac[sc.ta[i]].present) return false;
        bpm += f.ncomp == 1 ? 1 : f.comp[i].h * f.comp[i].v;
    }
    if (bpm > 10) return false;
    if ((sc.data_end - sc.data_begin) >= (1ull << 28)) return false;  // bit positions are 32-bit
    return true;
}

size_t destuff_scan(const uint8_t* data, const ScanHeader& sc, uint8_t* out)
{
    const uint8_t* p = data + sc.data_begin;
    const uint8_t* end = data + sc.data_end;
    uint8_t* o = out;
    while (p < end) {
        const uint8_t* ff = static_cast<const uint8_t*>(memchr(p, 0xFF, (size_t)(end - p)));
        if (!ff) {
            memcpy(o, p, (size_t)(end - p));
            o += end - p;
            break;
        }
        memcpy(o, p, (size_t)(ff - p));
        o += ff - p;
        p = ff + 1;
        while (p < end && *p == 0xFF) p++;  // fill bytes
        if (p < end && *p == 0x00) {
            *o++ = 0xFF;
            p++;
        } else {
            break;  // a marker (cannot happen before data_end for restart-free scans) or the end
        }
    }
    const size_t n = (size_t)(o - out);
    memset(o, 0xFF, kStreamSlackBytes);
    return n;
}

static void expand_table(const HuffSpec& s, HuffDecodeTable* t)
{
    memset(t, 0, sizeof *t);
    if (!s.present) {
        for (int l = 0; l < 18; l++) t->maxcode[l] = -1;
        return;
    }
    memcpy(t->vals, s.vals, sizeof t->vals);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        t->valoff[l] = k - code;
        if (s.bits[l]) {
            if (l <= kHuffFastBits)
                for (int i = 0; i < s.bits[l]; i++) {
                    const int lo = (code + i) << (kHuffFastBits - l);
                    const uint16_t e = (uint16_t)((l << 8) | s.vals[k + i]);
                    for (int j = 0; j < (1 << (kHuffFastBits - l)); j++) t->fast[lo + j] = e;
                }
            k += s.bits[l];
            code += s.bits[l];
            t->maxcode[l] = code - 1;
        } else {
            t->maxcode[l] = -1;
        }
        code <<= 1;
    }
    t->maxcode[0] = -1;
    t->maxcode[17] = 0x7fffffff;
}

void build_gpu_tables(const ScanHeader& sc, HuffDecodeTable out[8])
{
    for (int i = 0; i < 4; i++) {
        expand_table(sc.dc[i], &out[i]);
        expand_table(sc.ac[i], &out[4 + i]);
    }
}

void fill_huff_image(const FrameInfo& f, uint32_t stream_bytes, HuffImage* im)
{
    memset(im, 0, sizeof *im);
    const ScanHeader& sc = f.scans[0];
    im->total_bits = stream_bytes * 8u;
    im->num_subseq = (im->total_bits + kSubseqBits - 1) / kSubseqBits;
    im->mcus_x = (uint32_t)(f.ncomp == 1 ? (f.comp[0].samp_w + 7) / 8 : f.mcus_x);
    const uint32_t mcus_y = (uint32_t)(f.ncomp == 1 ? (f.comp[0].samp_h + 7) / 8 : f.mcus_y);
    im->ncomp = (uint32_t)f.ncomp;
    int k = 0;
    for (int c = 0; c < f.ncomp; c++) {
        const int h = f.ncomp == 1 ? 1 : f.comp[c].h, v = f.ncomp == 1 ? 1 : f.comp[c].v;
        im->comp_h[c] = (uint8_t)h;
        im->comp_v[c] = (uint8_t)v;
        im->blocks_w[c] = (uint32_t)f.comp[c].blocks_w;
        for (int dy = 0; dy < v; dy++)
            for (int dx = 0; dx < h; dx++, k++) {
                im->k_comp[k] = (uint8_t)c;
                im->k_dx[k] = (uint8_t)dx;
                im->k_dy[k] = (uint8_t)dy;
                im->k_dc[k] = (uint8_t)sc.td[c];
                im->k_ac[k] = (uint8_t)(4 + sc.ta[c]);
            }
    }
    im->blocks_per_mcu = (uint32_t)k;
    im->total_blocks = im->mcus_x * mcus_y * (uint32_t)k;
}

namespace {
struct TablesRef {
    const HuffDecodeTable* t;
    const HuffDecodeTable& operator[](int i) const { return t[i]; }
};
}  // namespace

int emulate_gpu_entropy(const uint8_t* data, size_t size, const FrameInfo& f, int16_t* const coef[4], int* sync_passes)
{
    (void)size;
    const ScanHeader& sc = f.scans[0];
    std::vector<uint8_t> stream(destuffed_capacity(sc));
    const size_t n = destuff_scan(data, sc, stream.data());
    std::vector<HuffDecodeTable> tables(8);
    build_gpu_tables(sc, tables.data());
    HuffImage im;
    fill_huff_image(f, (uint32_t)n, &im);
    im.stream = stream.data();
    im.tables = tables.data();
    for (int c = 0; c < f.ncomp; c++) {
        im.coef[c] = coef[c];
        memset(coef[c], 0, (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h * 128);
    }
    TablesRef tr{tables.data()};
    const uint32_t ns = im.num_subseq;
    std::vector<SubseqState> cur(ns), nxt(ns);
    uint32_t err = 0;
    // pass 0
    for (uint32_t i = 0; i < ns; i++) cur[i] = decode_subsequence<false>(im, tr, i * kSubseqBits, (i + 1) * kSubseqBits, 0, 0, 0, &err);
    int passes = 0;
    for (;;) {
        bool changed = false;
        nxt[0] = cur[0];
        for (uint32_t i = 1; i < ns; i++) {
            const SubseqState& prev = cur[i - 1];
            nxt[i] = decode_subsequence<false>(im, tr, prev.end_bit, (i + 1) * kSubseqBits, prev.zk & 255, prev.zk >> 8, 0, &err);
            if (!same_sync_state(nxt[i], cur[i]) || nxt[i].nblocks != cur[i].nblocks) changed = true;
        }
        cur.swap(nxt);
        passes++;
        if (!changed) break;
        if (passes > (int)ns + 2) return 3;  // cannot happen: the wave of corrections advances one subsequence per pass
    }
    if (sync_passes) *sync_passes = passes;
    // block index of each subsequence's first symbol
    std::vector<uint32_t> first_block(ns);
    uint32_t acc = 0;
    for (uint32_t i = 0; i < ns; i++) {
        first_block[i] = acc;
        acc += cur[i].nblocks;
    }
    if (acc < im.total_blocks) return 2;
    // write pass
    for (uint32_t i = 0; i < ns; i++) {
        const uint32_t begin = i == 0 ? 0 : cur[i - 1].end_bit;
        const int z = i == 0 ? 0 : (cur[i - 1].zk & 255), k = i == 0 ? 0 : (cur[i - 1].zk >> 8);
        decode_subsequence<true>(im, tr, begin, (i + 1) * kSubseqBits, z, k, first_block[i], &err);
    }
    if (err) return 1;
    // DC integration, per component in MCU (scan) order
    const uint32_t mcus = im.total_blocks / im.blocks_per_mcu;
    int pred[4] = {0, 0, 0, 0};
    for (uint32_t m = 0; m < mcus; m++) {
        const uint32_t my = m / im.mcus_x, mx = m - my * im.mcus_x;
        for (uint32_t k = 0; k < im.blocks_per_mcu; k++) {
            const int c = im.k_comp[k];
            int16_t* blk = coef[c] + ((size_t)(my * im.comp_v[c] + im.k_dy[k]) * im.blocks_w[c] + (mx * im.comp_h[c] + im.k_dx[k])) * 64;
            pred[c] += blk[0];
            blk[0] = (int16_t)pred[c];
        }
    }
    return 0;
}

}  // namespace hipjpeg
