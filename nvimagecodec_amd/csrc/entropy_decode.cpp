// entropy_decode.cpp -- see entropy_decode.h.  ITU T.81 Annex F (sequential) and Annex G (progressive).
#include "entropy_decode.h"

#include <cstring>

namespace hipjpeg {

// transposed natural order: natural position n = r*8 + c is stored at c*8 + r
const uint8_t kZigzagDevice[64] = {
    0,  8,  1,  2,  9,  16, 24, 17, 10, 3,  4,  11, 18, 25, 32, 40, 33, 26, 19, 12, 5,  6,  13, 20, 27, 34, 41, 48, 56, 49, 42, 35,
    28, 21, 14, 7,  15, 22, 29, 36, 43, 50, 57, 58, 51, 44, 37, 30, 23, 31, 38, 45, 52, 59, 60, 53, 46, 39, 47, 54, 61, 62, 55, 63};

namespace {

constexpr int kLook = 9;

struct DecodeTable {
    uint16_t look[1 << kLook];    // (len << 8) | symbol ; 0 => code longer than kLook bits
    int16_t fast_ac[1 << kLook];  // (value << 8) | (run << 4) | total_bits ; 0 => not applicable
    int32_t maxcode[18];          // largest code of length l (left-aligned to l bits), -1 if none
    int32_t valoff[17];
    uint8_t vals[256];
    bool present = false;
};

void build_table(const HuffSpec& s, DecodeTable* t, bool is_ac)
{
    memset(t->look, 0, sizeof t->look);
    memset(t->fast_ac, 0, sizeof t->fast_ac);
    memcpy(t->vals, s.vals, sizeof t->vals);
    t->present = s.present;
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        t->valoff[l] = k - code;
        if (s.bits[l]) {
            if (l <= kLook) {
                for (int i = 0; i < s.bits[l]; i++) {
                    int lo = (code + i) << (kLook - l);
                    uint16_t e = (uint16_t)((l << 8) | s.vals[k + i]);
                    for (int j = 0; j < (1 << (kLook - l)); j++) t->look[lo + j] = e;
                }
            }
            k += s.bits[l];
            code += s.bits[l];
            t->maxcode[l] = code - 1;
        } else {
            t->maxcode[l] = -1;
        }
        code <<= 1;
    }
    t->maxcode[17] = 0x7fffffff;
    if (is_ac) {
        // one-lookup decode of (run, size, value) when code + magnitude bits fit in the window
        for (int i = 0; i < (1 << kLook); i++) {
            uint16_t e = t->look[i];
            if (!e) continue;
            int len = e >> 8, rs = e & 255, run = rs >> 4, mag = rs & 15;
            if (mag == 0 || len + mag > kLook) continue;
            int v = (i >> (kLook - len - mag)) & ((1 << mag) - 1);
            if (v < (1 << (mag - 1))) v += (int)((~0u) << mag) + 1;  // T.81 F.2.2.1 EXTEND
            if (v >= -128 && v <= 127) t->fast_ac[i] = (int16_t)((v * 256) + (run << 4) + (len + mag));
        }
    }
}

inline bool has_ff_byte(uint64_t v)
{
    uint64_t x = ~v;  // a 0xFF byte becomes 0x00
    return ((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull) != 0;
}

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;  // next bits of the stream, MSB first
    int cnt = 0;       // number of valid bits at the top of acc
    int pad = 0;       // how many of them (at the tail) are invented zeros
    int marker = 0;    // marker code that stopped the reader (p points at its 0xFF)

    BitReader(const uint8_t* b, const uint8_t* e) : p(b), end(e) {}

    inline void refill()
    {
        if (cnt > 56) return;
        if (!marker && p + 8 <= end) {
            uint64_t v;
            memcpy(&v, p, 8);
            if (!has_ff_byte(v)) {
                // plain bytes: OR-in is idempotent for the partially consumed tail, so no masking is needed
                acc |= __builtin_bswap64(v) >> cnt;
                p += (63 - cnt) >> 3;
                cnt |= 56;
                return;
            }
        }
        acc = cnt ? (acc & (~0ull << (64 - cnt))) : 0;  // drop speculative low bits before byte-wise appends
        while (cnt <= 56) {
            unsigned c = 0;
            if (!marker && p < end) {
                c = *p;
                if (c == 0xFF) {
                    const uint8_t* q = p + 1;
                    while (q < end && *q == 0xFF) q++;  // fill bytes
                    if (q >= end) {
                        marker = 0xD9;
                        p = end;
                        c = 0;
                        pad += 8;
                    } else if (*q == 0x00) {
                        p = q + 1;
                    } else {
                        marker = *q;
                        p = q - 1;
                        c = 0;
                        pad += 8;
                    }
                } else {
                    p++;
                }
            } else {
                if (!marker) marker = 0xD9;
                pad += 8;
            }
            acc |= (uint64_t)c << (56 - cnt);
            cnt += 8;
        }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(acc >> (64 - n)); }
    inline void skip(int n)
    {
        acc <<= n;
        cnt -= n;
    }
    inline uint32_t get(int n)
    {
        uint32_t v = (uint32_t)(acc >> 1 >> (63 - n));  // n may be 0
        skip(n);
        return v;
    }
    inline bool overran() const { return cnt < pad; }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// Decodes one Huffman symbol; reader must hold >= 16 bits.  Returns -1 on an invalid code.
inline int decode_symbol(BitReader& br, const DecodeTable& t)
{
    uint16_t e = t.look[br.peek(kLook)];
    if (e) {
        br.skip(e >> 8);
        return e & 255;
    }
    uint32_t code16 = br.peek(16);
    for (int l = kLook + 1; l <= 16; l++) {
        int32_t code = (int32_t)(code16 >> (16 - l));
        if (code <= t.maxcode[l]) {
            br.skip(l);
            return t.vals[(code + t.valoff[l]) & 255];
        }
    }
    return -1;
}

struct ScanTables {
    DecodeTable dc[4], ac[4];
};

inline void zero_block(int16_t* b) { memset(b, 0, 64 * sizeof(int16_t)); }

// ---- sequential block (T.81 F.2.2) ----
inline EntropyStatus decode_block_sequential(BitReader& br, const DecodeTable& dct, const DecodeTable& act, int& pred, int16_t* blk,
                                             uint32_t& mag_or)
{
    zero_block(blk);
    br.refill();
    int s = decode_symbol(br, dct);
    if (s < 0 || s > 15) return kEntropyCorrupt;
    int diff = 0;
    if (s) diff = extend((int)br.get(s), s);
    pred += diff;
    blk[0] = (int16_t)pred;
    uint32_t m = (uint32_t)(pred < 0 ? -pred : pred);
    for (int k = 1; k < 64;) {
        br.refill();
        int f = act.fast_ac[br.peek(kLook)];
        if (f) {
            k += (f >> 4) & 15;
            if (k > 63) return kEntropyCorrupt;
            br.skip(f & 15);
            int v = f >> 8;
            blk[kZigzagDevice[k++]] = (int16_t)v;
            m |= (uint32_t)(v < 0 ? -v : v);
            continue;
        }
        int rs = decode_symbol(br, act);
        if (rs < 0) return kEntropyCorrupt;
        int r = rs >> 4;
        s = rs & 15;
        if (s) {
            k += r;
            if (k > 63) return kEntropyCorrupt;
            int v = extend((int)br.get(s), s);
            blk[kZigzagDevice[k++]] = (int16_t)v;
            m |= (uint32_t)(v < 0 ? -v : v);
        } else {
            if (r != 15) break;
            k += 16;
        }
    }
    mag_or |= m;
    return kEntropyOk;
}

// The same block as a sparse record (entropy_decode.h): w = where the record goes; returns the byte behind it.
inline EntropyStatus decode_block_sequential_sparse(BitReader& br, const DecodeTable& dct, const DecodeTable& act, int& pred, uint8_t*& w)
{
    uint8_t* rec = w;
    w += 3;
    br.refill();
    int s = decode_symbol(br, dct);
    if (s < 0 || s > 15) return kEntropyCorrupt;
    int diff = 0;
    if (s) diff = extend((int)br.get(s), s);
    pred += diff;
    rec[1] = (uint8_t)(pred & 255);
    rec[2] = (uint8_t)((pred >> 8) & 255);
    int n = 0;
    for (int k = 1; k < 64;) {
        br.refill();
        int v;
        int f = act.fast_ac[br.peek(kLook)];
        if (f) {
            k += (f >> 4) & 15;
            if (k > 63) return kEntropyCorrupt;
            br.skip(f & 15);
            v = f >> 8;
        } else {
            int rs = decode_symbol(br, act);
            if (rs < 0) return kEntropyCorrupt;
            int r = rs >> 4;
            s = rs & 15;
            if (!s) {
                if (r != 15) break;
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) return kEntropyCorrupt;
            v = extend((int)br.get(s), s);
        }
        w[0] = kZigzagDevice[k++];
        w[1] = (uint8_t)(v & 255);
        w[2] = (uint8_t)((v >> 8) & 255);
        w += 3;
        n++;
    }
    rec[0] = (uint8_t)n;
    return kEntropyOk;
}

// ---- progressive pieces (T.81 G.1.2) ----
inline EntropyStatus decode_dc_first(BitReader& br, const DecodeTable& dct, int& pred, int16_t* blk, int al)
{
    br.refill();
    int s = decode_symbol(br, dct);
    if (s < 0 || s > 15) return kEntropyCorrupt;
    if (s) pred += extend((int)br.get(s), s);
    blk[0] = (int16_t)(pred * (1 << al));
    return kEntropyOk;
}

inline void decode_dc_refine(BitReader& br, int16_t* blk, int al)
{
    br.refill();
    if (br.get(1)) blk[0] = (int16_t)(blk[0] | (1 << al));
}

inline EntropyStatus decode_ac_first(BitReader& br, const DecodeTable& act, int16_t* blk, int ss, int se, int al, uint32_t& eobrun)
{
    if (eobrun) {
        eobrun--;
        return kEntropyOk;
    }
    for (int k = ss; k <= se; k++) {
        br.refill();
        int rs = decode_symbol(br, act);
        if (rs < 0) return kEntropyCorrupt;
        int r = rs >> 4, s = rs & 15;
        if (s) {
            k += r;
            if (k > 63) return kEntropyCorrupt;
            blk[kZigzagDevice[k]] = (int16_t)(extend((int)br.get(s), s) * (1 << al));
        } else if (r == 15) {
            k += 15;
        } else {
            eobrun = (1u << r) - 1;
            if (r) eobrun += br.get(r);
            break;
        }
    }
    return kEntropyOk;
}

inline void refine_nonzero(BitReader& br, int16_t* c, int p1, int m1)
{
    br.refill();
    if (br.get(1) && (*c & p1) == 0) *c = (int16_t)(*c >= 0 ? *c + p1 : *c + m1);
}

inline EntropyStatus decode_ac_refine(BitReader& br, const DecodeTable& act, int16_t* blk, int ss, int se, int al, uint32_t& eobrun)
{
    const int p1 = 1 << al, m1 = -(1 << al);
    int k = ss;
    if (eobrun == 0) {
        for (; k <= se; k++) {
            br.refill();
            int rs = decode_symbol(br, act);
            if (rs < 0) return kEntropyCorrupt;
            int r = rs >> 4, s = rs & 15, newval = 0;
            if (s) {
                if (s != 1) return kEntropyCorrupt;
                newval = br.get(1) ? p1 : m1;
            } else if (r != 15) {
                eobrun = 1u << r;
                if (r) eobrun += br.get(r);
                break;
            }
            // skip r zero-history coefficients, refining the nonzero ones passed on the way
            while (k <= se) {
                int16_t* c = &blk[kZigzagDevice[k]];
                if (*c != 0) {
                    refine_nonzero(br, c, p1, m1);
                } else if (--r < 0) {
                    break;
                }
                k++;
            }
            if (newval) {
                if (k > 63) return kEntropyCorrupt;
                blk[kZigzagDevice[k]] = (int16_t)newval;
            }
        }
    }
    if (eobrun > 0) {
        for (; k <= se; k++) {
            int16_t* c = &blk[kZigzagDevice[k]];
            if (*c != 0) refine_nonzero(br, c, p1, m1);
        }
        eobrun--;
    }
    return kEntropyOk;
}

// Consumes the RSTn marker that must sit at the current position.
EntropyStatus take_restart(BitReader& br, int& expected)
{
    if (br.overran()) return kEntropyTruncated;
    br.acc = 0;
    br.cnt = 0;
    br.pad = 0;
    if (!br.marker) {
        // reader had not reached the marker yet (only padding bits were left): find it
        const uint8_t* q = br.p;
        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF)) q++;
        if (q + 1 >= br.end) return kEntropyTruncated;
        br.p = q;
        br.marker = q[1];
    }
    if (br.marker != 0xD0 + expected) return kEntropyCorrupt;
    br.p += 2;
    br.marker = 0;
    expected = (expected + 1) & 7;
    return kEntropyOk;
}

EntropyStatus decode_scan(const uint8_t* data, const FrameInfo& f, const ScanHeader& sc, int16_t* const coef[4], uint32_t mag_or[4])
{
    ScanTables tabs;
    const bool prog = f.progressive();
    const bool need_dc = !prog || (sc.ss == 0 && sc.ah == 0);
    const bool need_ac = !prog || sc.ss > 0;
    for (int i = 0; i < sc.ncomp; i++) {
        if (need_dc) {
            if (!sc.dc[sc.td[i]].present) return kEntropyMissingTable;
            if (!tabs.dc[sc.td[i]].present) build_table(sc.dc[sc.td[i]], &tabs.dc[sc.td[i]], false);
        }
        if (need_ac) {
            if (!sc.ac[sc.ta[i]].present) return kEntropyMissingTable;
            if (!tabs.ac[sc.ta[i]].present) build_table(sc.ac[sc.ta[i]], &tabs.ac[sc.ta[i]], true);
        }
    }

    BitReader br(data + sc.data_begin, data + sc.data_end);
    int pred[4] = {0, 0, 0, 0};
    uint32_t eobrun = 0;
    int until_restart = sc.restart_interval, next_rst = 0;
    EntropyStatus st;

    if (sc.ncomp == 1) {
        // single-component scan: one block per MCU, covering only real samples (T.81 A.2.2)
        const int ci = sc.comp_index[0];
        const Component& k = f.comp[ci];
        const int nbx = (k.samp_w + 7) / 8, nby = (k.samp_h + 7) / 8;
        const DecodeTable& dct = tabs.dc[sc.td[0]];
        const DecodeTable& act = tabs.ac[sc.ta[0]];
        for (int by = 0; by < nby; by++) {
            int16_t* row = coef[ci] + (size_t)by * k.blocks_w * 64;
            for (int bx = 0; bx < nbx; bx++) {
                int16_t* blk = row + (size_t)bx * 64;
                if (sc.restart_interval && until_restart == 0) {
                    if ((st = take_restart(br, next_rst)) != kEntropyOk) return st;
                    pred[0] = 0;
                    eobrun = 0;
                    until_restart = sc.restart_interval;
                }
                if (!prog)
                    st = decode_block_sequential(br, dct, act, pred[0], blk, mag_or[ci]);
                else if (sc.ss == 0) {
                    st = kEntropyOk;
                    if (sc.ah == 0)
                        st = decode_dc_first(br, dct, pred[0], blk, sc.al);
                    else
                        decode_dc_refine(br, blk, sc.al);
                } else if (sc.ah == 0)
                    st = decode_ac_first(br, act, blk, sc.ss, sc.se, sc.al, eobrun);
                else
                    st = decode_ac_refine(br, act, blk, sc.ss, sc.se, sc.al, eobrun);
                if (st != kEntropyOk) return st;
                until_restart--;
            }
        }
    } else {
        for (int my = 0; my < f.mcus_y; my++) {
            for (int mx = 0; mx < f.mcus_x; mx++) {
                if (sc.restart_interval && until_restart == 0) {
                    if ((st = take_restart(br, next_rst)) != kEntropyOk) return st;
                    pred[0] = pred[1] = pred[2] = pred[3] = 0;
                    until_restart = sc.restart_interval;
                }
                for (int i = 0; i < sc.ncomp; i++) {
                    const int ci = sc.comp_index[i];
                    const Component& k = f.comp[ci];
                    const DecodeTable& dct = tabs.dc[sc.td[i]];
                    const DecodeTable& act = tabs.ac[sc.ta[i]];
                    for (int v = 0; v < k.v; v++) {
                        int16_t* blk = coef[ci] + ((size_t)(my * k.v + v) * k.blocks_w + (size_t)mx * k.h) * 64;
                        for (int h = 0; h < k.h; h++, blk += 64) {
                            if (!prog) {
                                st = decode_block_sequential(br, dct, act, pred[i], blk, mag_or[ci]);
                                if (st != kEntropyOk) return st;
                            } else if (sc.ah == 0) {
                                st = decode_dc_first(br, dct, pred[i], blk, sc.al);
                                if (st != kEntropyOk) return st;
                            } else {
                                decode_dc_refine(br, blk, sc.al);
                            }
                        }
                    }
                }
                until_restart--;
            }
        }
    }
    return br.overran() ? kEntropyTruncated : kEntropyOk;
}

}  // namespace

bool sparse_staging_applies(const FrameInfo& f)
{
    if (f.progressive() || f.scans.size() != 1 || f.ncomp < 1 || f.ncomp > 4) return false;
    return f.scans[0].ncomp == f.ncomp;  // every block is coded once, by this scan
}

size_t sparse_stream_capacity(const FrameInfo& f)
{
    size_t blocks = 0;
    for (int c = 0; c < f.ncomp; c++) blocks += (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h;
    return blocks * (4 + 3 + 63 * 3) + 64;
}

EntropyStatus decode_coefficients_sparse(const uint8_t* data, size_t size, const FrameInfo& f, uint8_t* out, size_t* out_size)
{
    (void)size;
    const ScanHeader& sc = f.scans[0];
    ScanTables tabs;
    for (int i = 0; i < sc.ncomp; i++) {
        if (!sc.dc[sc.td[i]].present || !sc.ac[sc.ta[i]].present) return kEntropyMissingTable;
        if (!tabs.dc[sc.td[i]].present) build_table(sc.dc[sc.td[i]], &tabs.dc[sc.td[i]], false);
        if (!tabs.ac[sc.ta[i]].present) build_table(sc.ac[sc.ta[i]], &tabs.ac[sc.ta[i]], true);
    }
    // offset tables first (zero = never coded), records behind them
    uint32_t* table[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t blocks = 0;
    for (int c = 0; c < f.ncomp; c++) {
        table[c] = reinterpret_cast<uint32_t*>(out) + blocks;
        blocks += (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h;
    }
    memset(out, 0, blocks * 4);
    uint8_t* w = out + blocks * 4;
    BitReader br(data + sc.data_begin, data + sc.data_end);
    int pred[4] = {0, 0, 0, 0};
    int until_restart = sc.restart_interval, next_rst = 0;
    EntropyStatus st;
    if (sc.ncomp == 1) {
        const int ci = sc.comp_index[0];
        const Component& k = f.comp[ci];
        const int nbx = (k.samp_w + 7) / 8, nby = (k.samp_h + 7) / 8;
        for (int by = 0; by < nby; by++)
            for (int bx = 0; bx < nbx; bx++) {
                if (sc.restart_interval && until_restart == 0) {
                    if ((st = take_restart(br, next_rst)) != kEntropyOk) return st;
                    pred[0] = 0;
                    until_restart = sc.restart_interval;
                }
                table[ci][(size_t)by * k.blocks_w + bx] = (uint32_t)(w - out);
                if ((st = decode_block_sequential_sparse(br, tabs.dc[sc.td[0]], tabs.ac[sc.ta[0]], pred[0], w)) != kEntropyOk) return st;
                until_restart--;
            }
    } else {
        for (int my = 0; my < f.mcus_y; my++)
            for (int mx = 0; mx < f.mcus_x; mx++) {
                if (sc.restart_interval && until_restart == 0) {
                    if ((st = take_restart(br, next_rst)) != kEntropyOk) return st;
                    pred[0] = pred[1] = pred[2] = pred[3] = 0;
                    until_restart = sc.restart_interval;
                }
                for (int i = 0; i < sc.ncomp; i++) {
                    const int ci = sc.comp_index[i];
                    const Component& k = f.comp[ci];
                    for (int v = 0; v < k.v; v++)
                        for (int hh = 0; hh < k.h; hh++) {
                            table[ci][(size_t)(my * k.v + v) * k.blocks_w + (size_t)mx * k.h + hh] = (uint32_t)(w - out);
                            if ((st = decode_block_sequential_sparse(br, tabs.dc[sc.td[i]], tabs.ac[sc.ta[i]], pred[i], w)) != kEntropyOk) return st;
                        }
                }
                until_restart--;
            }
    }
    if (br.overran()) return kEntropyTruncated;
    *out_size = (size_t)(w - out);
    return kEntropyOk;
}

EntropyStatus decode_coefficients(const uint8_t* data, size_t size, const FrameInfo& f, int16_t* const coef[4], uint32_t coef_or[4])
{
    uint32_t mag_or[4] = {0, 0, 0, 0};
    (void)size;
    // Sequential frames where one interleaved scan covers every component write each block exactly once
    // (decode_block_sequential zeroes it first), so the big memset is only needed otherwise.
    bool single_full_scan = !f.progressive() && f.scans.size() == 1 && f.scans[0].ncomp == f.ncomp && f.ncomp > 1;
    if (!single_full_scan) {
        for (int c = 0; c < f.ncomp; c++) memset(coef[c], 0, (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h * 64 * sizeof(int16_t));
    }
    bool covered[4] = {false, false, false, false};
    for (const ScanHeader& sc : f.scans) {
        EntropyStatus st = decode_scan(data, f, sc, coef, mag_or);
        if (st != kEntropyOk) return st;
        for (int i = 0; i < sc.ncomp; i++) covered[sc.comp_index[i]] = true;
    }
    for (int c = 0; c < f.ncomp; c++)
        if (!covered[c]) return kEntropyCorrupt;  // a component without any scan: incomplete file
    if (f.progressive()) {
        // successive approximation builds values up over several scans: measure the final magnitudes directly
        for (int c = 0; c < f.ncomp; c++) {
            const int16_t* p = coef[c];
            size_t n = (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h * 64;
            uint32_t m = 0;
            for (size_t i = 0; i < n; i++) m |= (uint32_t)(p[i] < 0 ? -p[i] : p[i]);
            mag_or[c] = m;
        }
    }
    if (coef_or)
        for (int c = 0; c < 4; c++) coef_or[c] = mag_or[c];
    return kEntropyOk;
}

}  // namespace hipjpeg
