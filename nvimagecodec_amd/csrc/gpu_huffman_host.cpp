// gpu_huffman_host.cpp -- see gpu_huffman_host.h
#include "gpu_huffman_host.h"

#include <cstring>

namespace hipjpeg {

bool gpu_entropy_eligible(const FrameInfo& f)
{
    if (f.progressive() || f.scans.size() != 1) return false;
    const ScanHeader& sc = f.scans[0];
    if (sc.ncomp != f.ncomp || !sc.plain_stuffing) return false;
    {
        // restart intervals: every interval but the last must be closed by its marker (RST0..7 in order: plain_stuffing)
        const size_t mcus = (size_t)(f.ncomp == 1 ? ((f.comp[0].samp_w + 7) / 8) * ((f.comp[0].samp_h + 7) / 8) : f.mcus_x * f.mcus_y);
        const size_t expect = sc.restart_interval ? (mcus + sc.restart_interval - 1) / sc.restart_interval - 1 : 0;
        if (sc.rst_after.size() != expect) return false;
    }
    int bpm = 0;
    for (int i = 0; i < sc.ncomp; i++) {
        if (sc.comp_index[i] != i) return false;  // keep the MCU layout simple: components in frame order
        if (!sc.dc[sc.td[i]].present || !sc.ac[sc.ta[i]].present) return false;
        bpm += f.ncomp == 1 ? 1 : f.comp[i].h * f.comp[i].v;
    }
    if (bpm > 10) return false;
    if ((sc.data_end - sc.data_begin) >= (1ull << 28)) return false;  // bit positions are 32-bit
    const size_t words = gpu_pool_words(sc);
    return words != 0 && words <= (size_t)kMaxPoolWords;
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
        } else if (p < end && *p >= 0xD0 && *p <= 0xD7) {
            p++;  // restart marker: dropped; the interval behind it starts at this (byte-aligned) position
        } else {
            break;  // another marker (cannot happen before data_end) or the end
        }
    }
    const size_t n = (size_t)(o - out);
    memset(o, 0xFF, kStreamSlackBytes + ((4 - (n & 3)) & 3));  // whole 32-bit words stay readable past the end
    return n;
}

namespace {

// Walks the canonical code of `s`; calls fn(length, code, symbol) for every code.  False if the code is over-subscribed.
template <class Fn>
bool for_each_code(const HuffSpec& s, Fn fn)
{
    uint32_t code = 0;
    int k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < s.bits[l]; i++, k++, code++) {
            if (code >= (1u << l) || k >= 256) return false;
            fn(l, code, s.vals[k]);
        }
        code <<= 1;
    }
    return true;
}

// entries of one table: first level + second-level tables; 0 = malformed
size_t table_words(const HuffSpec& s, bool is_dc)
{
    if (!s.present) return 0;
    std::vector<uint8_t> has_sub(1u << kHuffFastBits, 0);
    size_t words = (1u << kHuffFastBits) + (is_dc ? 0u : kPairOffset);  // AC: first level + pair table
    bool bad = false;
    bool ok = for_each_code(s, [&](int l, uint32_t code, uint8_t sym) {
        if (is_dc && sym > 15) bad = true;
        if (l > kHuffFastBits) {
            const uint32_t prefix = code >> (l - kHuffFastBits);
            if (!has_sub[prefix]) {
                has_sub[prefix] = 1;
                words += 1u << kHuffSubBits;
            }
        }
    });
    return ok && !bad ? words : 0;
}

// Expands one table at pool[base...] (base is a multiple of 64); returns the number of entries used.
size_t expand_table(const HuffSpec& s, bool is_dc, uint16_t* pool, size_t base)
{
    uint16_t* first = pool + base;
    for (int i = 0; i < (1 << kHuffFastBits); i++) first[i] = (uint16_t)kEntryInvalid;
    size_t used = (1u << kHuffFastBits) + (is_dc ? 0u : kPairOffset);
    for_each_code(s, [&](int l, uint32_t code, uint8_t sym) {
        const uint32_t nb = sym & 15u, run = sym >> 4;
        // DC: the symbol is the size category.  AC: (run, size); size 0 is EOB except for run 15 (ZRL, 16 zeros) -- libjpeg
        // treats every other run with size 0 as EOB too (jdhuff.c decode_mcu: "if (r != 15) break").
        const uint32_t zadv = is_dc ? 1u : (nb ? run + 1 : (run == 15 ? 16u : 64u));
        const uint16_t e = (uint16_t)make_entry((uint32_t)l + nb, nb, zadv);
        if (l <= kHuffFastBits) {
            const uint32_t lo = code << (kHuffFastBits - l);
            for (uint32_t j = 0; j < (1u << (kHuffFastBits - l)); j++) first[lo + j] = e;
        } else {
            const uint32_t prefix = code >> (l - kHuffFastBits);
            if ((first[prefix] >> 9) != kZadvLong) {  // still "no such code": open a second-level table
                first[prefix] = (uint16_t)(((base + used) / 64) | (kZadvLong << 9));
                for (int j = 0; j < (1 << kHuffSubBits); j++) pool[base + used + j] = (uint16_t)kEntryInvalid;
                used += 1u << kHuffSubBits;
            }
            uint16_t* sub = pool + (size_t)(first[prefix] & 0x1FFu) * 64;
            const uint32_t lo = (code & ((1u << (l - kHuffFastBits)) - 1)) << (16 - l);
            for (uint32_t j = 0; j < (1u << (16 - l)); j++) sub[lo + j] = e;
        }
    });
    if (!is_dc) {
        // pair table (huffman_gpu_core.h): window w = a first symbol of t1 bits and, in the 10 - t1 bits behind it, a whole second one
        uint16_t* pair = first + kPairOffset;
        for (uint32_t w = 0; w < (1u << kHuffFastBits); w++) {
            pair[w] = 0;
            const uint32_t e1 = first[w], t1 = e1 & 31u, z1 = e1 >> 9;
            if (e1 == kEntryInvalid || z1 == kZadvLong || z1 == 64u || t1 + 2 > (uint32_t)kHuffFastBits) continue;
            const uint32_t e2 = first[(w << t1) & ((1u << kHuffFastBits) - 1)], t2 = e2 & 31u, z2 = e2 >> 9;
            if (e2 == kEntryInvalid || z2 == kZadvLong || t1 + t2 > (uint32_t)kHuffFastBits) continue;
            pair[w] = (uint16_t)make_pair_entry(t1 + t2, z2 == 64u ? 63u : z1 + z2, 64u - z1);
        }
    }
    return used;
}

}  // namespace

size_t gpu_pool_words(const ScanHeader& sc)
{
    size_t words = 0;
    bool dc_seen[4] = {false, false, false, false}, ac_seen[4] = {false, false, false, false};
    for (int i = 0; i < sc.ncomp; i++) {
        const int td = sc.td[i], ta = sc.ta[i];
        if (!dc_seen[td]) {
            dc_seen[td] = true;
            const size_t w = table_words(sc.dc[td], true);
            if (!w) return 0;
            words += w;
        }
        if (!ac_seen[ta]) {
            ac_seen[ta] = true;
            const size_t w = table_words(sc.ac[ta], false);
            if (!w) return 0;
            words += w;
        }
    }
    return words;
}

void build_gpu_pool(const ScanHeader& sc, HuffImage* im, uint16_t* pool)
{
    size_t dc_off[4] = {0, 0, 0, 0}, ac_off[4] = {0, 0, 0, 0};
    bool dc_seen[4] = {false, false, false, false}, ac_seen[4] = {false, false, false, false};
    size_t used = 0;
    // DC tables first, AC tables behind them: the walkers read the pair-table slot (kPairOffset entries behind the first level)
    // of whatever table they are at -- behind a DC table that is some other table's entries, ignored, but inside the pool
    for (int i = 0; i < sc.ncomp; i++) {
        const int td = sc.td[i];
        if (!dc_seen[td]) {
            dc_seen[td] = true;
            dc_off[td] = used;
            used += expand_table(sc.dc[td], true, pool, used);
        }
    }
    for (int i = 0; i < sc.ncomp; i++) {
        const int ta = sc.ta[i];
        if (!ac_seen[ta]) {
            ac_seen[ta] = true;
            ac_off[ta] = used;
            used += expand_table(sc.ac[ta], false, pool, used);
        }
    }
    im->pool_words = (uint32_t)used;
    for (uint32_t k = 0; k < im->blocks_per_mcu; k++) {
        const int c = im->k[k].comp;
        im->k[k].tdc = (uint16_t)dc_off[sc.td[c]];
        im->k[k].tac = (uint16_t)ac_off[sc.ta[c]];
    }
}

void fill_huff_image(const FrameInfo& f, uint32_t stream_bytes, HuffImage* im)
{
    memset(im, 0, sizeof *im);
    im->total_bits = stream_bytes * 8u;
    im->num_subseq = (im->total_bits + kSubseqBits - 1) / kSubseqBits;
    im->stream_words = (uint32_t)((((size_t)stream_bytes + 3) & ~(size_t)3) + kStreamSlackBytes) / 4;
    im->mcus_x = (uint32_t)(f.ncomp == 1 ? (f.comp[0].samp_w + 7) / 8 : f.mcus_x);
    im->mcus_y = (uint32_t)(f.ncomp == 1 ? (f.comp[0].samp_h + 7) / 8 : f.mcus_y);
    im->ncomp = (uint32_t)f.ncomp;
    int k = 0;
    for (int c = 0; c < f.ncomp; c++) {
        const int h = f.ncomp == 1 ? 1 : f.comp[c].h, v = f.ncomp == 1 ? 1 : f.comp[c].v;
        im->comp_h[c] = (uint8_t)h;
        im->comp_v[c] = (uint8_t)v;
        im->comp_k0[c] = (uint8_t)k;
        im->blocks_w[c] = (uint32_t)f.comp[c].blocks_w;
        for (int dy = 0; dy < v; dy++)
            for (int dx = 0; dx < h; dx++, k++) {
                HuffK& hk = im->k[k];
                hk.comp = (uint8_t)c;
                hk.blk0 = (uint32_t)(dy * f.comp[c].blocks_w + dx);
                hk.stride_y = (uint32_t)(v * f.comp[c].blocks_w);
                hk.stride_x = (uint8_t)h;
            }
    }
    im->blocks_per_mcu = (uint32_t)k;
    im->total_blocks = im->mcus_x * im->mcus_y * (uint32_t)k;
}

namespace {
// plain-memory accessors for decode_subsequence
struct HostEnv {
    const HuffImage* im;
    static constexpr uint32_t kCursorStep = 1;  // the cursor is the word index
    uint32_t cursor(uint32_t i) const { return i; }
    uint32_t fetch(uint32_t i) const { return word(i); }
    uint32_t word(uint32_t i) const
    {
        const uint8_t* p = im->stream + (size_t)i * 4;  // the slack behind the stream covers the reader's look-ahead
        return i < im->stream_words ? ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3] : ~0u;
    }
    uint32_t boundary(uint32_t i) const { return i < im->num_boundaries ? im->boundaries[i] : 0xFFFFFFFFu; }
    uint32_t lookup1(uint32_t t, uint32_t w) const { return im->pool[t + (w >> (32 - kHuffFastBits))]; }
    uint32_t lookup2(uint32_t e, uint32_t w) const { return im->pool[(e & 0x1FFu) * 64u + ((w >> 16) & ((1u << kHuffSubBits) - 1))]; }
    uint32_t lookup_pair(uint32_t t, uint32_t w) const
    {
        const size_t i = (size_t)t + kPairOffset + (w >> (32 - kHuffFastBits));
        return i < im->pool_words ? im->pool[i] : 0u;  // (behind the last DC table there may be nothing)
    }
    uint32_t tables(int k) const { return (uint32_t)im->k[k].tdc | ((uint32_t)im->k[k].tac << 16); }
    int16_t* block_ptr(int k, uint32_t mx, uint32_t my) const
    {
        const HuffK& hk = im->k[k];
        return im->coef[hk.comp] + ((size_t)hk.blk0 + (size_t)my * hk.stride_y + (size_t)mx * hk.stride_x) * 64;
    }
    int zigzag(int z) const { return kZigzagDeviceGpuHost[z]; }
    int16_t* buf;  // block buffer: 64 coefficients
    void put(int index, int value) const { buf[index] = (int16_t)value; }
};

// The window of cooperative_subsequence (huffman_gpu_core.h) on the host: what the 64 lanes of the kernel's wave look up at once.
struct HostWindow {
    const HostEnv* env;
    uint32_t edc[64], eac[64];
    uint32_t tables(uint32_t k) const { return env->tables((int)k); }
    uint32_t entry(uint32_t table, uint32_t w) const
    {
        uint32_t e = env->lookup1(table, w);
        if ((e >> 9) == kZadvLong) e = env->lookup2(e, w);
        return e;
    }
    void open(uint32_t pos, uint32_t ts)
    {
        for (uint32_t l = 0; l < 64; l++) {
            const uint32_t b = pos + l, bit = b & 31u;
            const uint32_t w0 = env->word(b >> 5), w1 = env->word((b >> 5) + 1);
            const uint32_t w = bit ? (w0 << bit) | (w1 >> (32u - bit)) : w0;
            edc[l] = entry(ts & 0xFFFFu, w);
            eac[l] = entry(ts >> 16, w);
        }
    }
    uint32_t dc(uint32_t rel) const { return edc[rel]; }
    uint32_t ac(uint32_t rel) const { return eac[rel]; }
};
}  // namespace

int emulate_gpu_entropy(const uint8_t* data, size_t size, const FrameInfo& f, int16_t* const coef[4], int* sync_passes)
{
    (void)size;
    const ScanHeader& sc = f.scans[0];
    std::vector<uint8_t> stream(destuffed_capacity(sc));
    const size_t n = destuff_scan(data, sc, stream.data());
    std::vector<uint16_t> pool(gpu_pool_words(sc));
    HuffImage im;
    fill_huff_image(f, (uint32_t)n, &im);
    build_gpu_pool(sc, &im, pool.data());
    std::vector<int16_t> dc_diff(im.total_blocks);
    std::vector<uint32_t> boundaries;
    for (uint32_t b : sc.rst_after) boundaries.push_back(b * 8u);
    im.stream = stream.data();
    im.pool = pool.data();
    im.dc_diff = dc_diff.data();
    im.boundaries = boundaries.data();
    im.num_boundaries = (uint32_t)boundaries.size();
    im.restart_interval = (uint32_t)sc.restart_interval;
    const bool rst = sc.restart_interval != 0;
    for (int c = 0; c < f.ncomp; c++) {
        im.coef[c] = coef[c];
        memset(coef[c], 0x5A, (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h * 128);  // every block must be written by the write pass
    }
    int16_t block_buffer[64] = {0};
    const HostEnv env{&im, block_buffer};
    const HuffGeom geom = make_geom(im);
    const uint32_t ns = im.num_subseq;
    std::vector<SubseqState> cur(ns), nxt(ns);
    uint32_t err = 0;
    // pass 0
    auto walk = [&](uint32_t begin, uint32_t limit, int z, int k) {
        return rst ? decode_subsequence<true>(geom, env, begin, limit, z, k, 0) : decode_subsequence<false>(geom, env, begin, limit, z, k);
    };
    for (uint32_t i = 0; i < ns; i++) cur[i] = walk(i * kSubseqBits, (i + 1) * kSubseqBits, 0, 0);
    int passes = 0;
    for (;;) {
        bool changed = false;
        if (ns) nxt[0] = cur[0];
        for (uint32_t i = 1; i < ns; i++) {
            const SubseqState& prev = cur[i - 1];
            nxt[i] = walk(prev.end_bit, (i + 1) * kSubseqBits, prev.zk & 255, prev.zk >> 8);
            if (pack_state(nxt[i]) != pack_state(cur[i])) changed = true;
        }
        cur.swap(nxt);
        passes++;
        if (!changed) break;
        if (passes > (int)ns + 2) return 3;  // cannot happen: the wave of corrections advances one subsequence per pass
    }
    if (sync_passes) *sync_passes = passes;
    if (!rst) {
        // The kernels follow the last links of a correction chain with the whole wave on one subsequence (cooperative_subsequence):
        // the same routine, here, must reproduce the lane decoder's end state for every subsequence -- from its true start state and
        // from the state pass 0 assumes (a trajectory through garbage).
        HostWindow win;
        win.env = &env;
        const uint32_t changes = cooperative_table_changes(win, geom.blocks_per_mcu);
        for (uint32_t i = 0; i < ns; i++) {
            const uint32_t begin = i ? cur[i - 1].end_bit : 0u, z = i ? (cur[i - 1].zk & 255u) : 0u, k = i ? (uint32_t)(cur[i - 1].zk >> 8) : 0u;
            if (pack_state(cooperative_subsequence(geom, win, changes, begin, (i + 1) * kSubseqBits, z, k)) != pack_state(cur[i])) return 4;
            if (pack_state(cooperative_subsequence(geom, win, changes, i * kSubseqBits, (i + 1) * kSubseqBits, 0, 0)) !=
                pack_state(walk(i * kSubseqBits, (i + 1) * kSubseqBits, 0, 0)))
                return 4;
        }
    }
    // block index of each subsequence's first symbol
    std::vector<uint32_t> first_block(ns);
    uint32_t acc = 0;
    for (uint32_t i = 0; i < ns; i++) {
        first_block[i] = acc;
        acc += cur[i].nblocks;
    }
    if (acc < im.total_blocks) return 2;
    // write pass, step 1: block start positions
    std::vector<uint32_t> block_pos(im.total_blocks, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < ns; i++) {
        const uint32_t begin = i == 0 ? 0 : cur[i - 1].end_bit;
        const int z = i == 0 ? 0 : (cur[i - 1].zk & 255), k = i == 0 ? 0 : (cur[i - 1].zk >> 8);
        auto rec = [&](uint32_t block, uint32_t pos) {
            if (block < im.total_blocks) block_pos[block] = pos;
        };
        if (rst) {
            uint32_t fault = 0;
            position_subsequence<true>(geom, env, begin, (i + 1) * kSubseqBits, z, k, first_block[i], rec, 0, &fault);
            if (fault) return 1;
        } else
            position_subsequence<false>(geom, env, begin, (i + 1) * kSubseqBits, z, k, first_block[i], rec);
    }
    // restart intervals: the first block of interval j+1 starts exactly at boundary j (see huff_blocks_kernel)
    if (geom.interval_blocks)
        for (uint32_t j = 0; j < im.num_boundaries && (uint64_t)(j + 1) * geom.interval_blocks < im.total_blocks; j++)
            if (block_pos[(j + 1) * geom.interval_blocks] != im.boundaries[j]) return 1;
    // step 2: every block on its own
    for (uint32_t b = 0; b < im.total_blocks; b++) {
        if (block_pos[b] == 0xFFFFFFFFu) return 2;
        int k;
        int16_t* dst = block_address(geom, env, b, &k);
        memset(block_buffer, 0, sizeof block_buffer);
        dc_diff[b] = (int16_t)decode_block(geom, env, block_pos[b], k, &err);
        memcpy(dst, block_buffer, 128);
    }
    if (err) return 1;
    // DC integration, per component in MCU (scan) order
    int pred[4] = {0, 0, 0, 0};
    uint32_t block = 0, mcu = 0;
    for (uint32_t my = 0; my < im.mcus_y; my++)
        for (uint32_t mx = 0; mx < im.mcus_x; mx++, mcu++) {
            if (rst && mcu % im.restart_interval == 0) pred[0] = pred[1] = pred[2] = pred[3] = 0;  // DC predictors start over
            for (uint32_t k = 0; k < im.blocks_per_mcu; k++, block++) {
                const int c = im.k[k].comp;
                pred[c] += dc_diff[block];
                env.block_ptr((int)k, mx, my)[0] = (int16_t)pred[c];
            }
        }
    return 0;
}

}  // namespace hipjpeg
