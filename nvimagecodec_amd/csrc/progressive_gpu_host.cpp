// progressive_gpu_host.cpp -- see progressive_gpu_host.h
#include "progressive_gpu_host.h"

#include <algorithm>

#include <cstring>

#include "gpu_huffman_host.h"

namespace hipjpeg {

namespace {

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

inline size_t align64(size_t v) { return (v + 63) & ~(size_t)63; }

}  // namespace

size_t prog_table_words(const HuffSpec& s)
{
    if (!s.present) return 0;
    bool sub[256] = {false};
    size_t words = 256;
    const bool ok = for_each_code(s, [&](int l, uint32_t code, uint8_t) {
        if (l > 8 && !sub[code >> (l - 8)]) {
            sub[code >> (l - 8)] = true;
            words += 256;
        }
    });
    return ok ? words : 0;
}

void build_prog_table(const HuffSpec& s, uint16_t* out)
{
    memset(out, 0, 256 * sizeof(uint16_t));
    size_t used = 256;
    for_each_code(s, [&](int l, uint32_t code, uint8_t sym) {
        const uint16_t e = (uint16_t)prog_entry((uint32_t)l, sym);
        if (l <= 8) {
            const uint32_t lo = code << (8 - l);
            for (uint32_t j = 0; j < (1u << (8 - l)); j++) out[lo + j] = e;
        } else {
            const uint32_t prefix = code >> (l - 8);
            if (!(out[prefix] & kProgLong)) {
                out[prefix] = (uint16_t)(kProgLong | (used / 256));
                memset(out + used, 0, 256 * sizeof(uint16_t));
                used += 256;
            }
            uint16_t* sub = out + (size_t)(out[prefix] & 0x7FFFu) * 256;
            const uint32_t lo = (code & ((1u << (l - 8)) - 1)) << (16 - l);
            for (uint32_t j = 0; j < (1u << (16 - l)); j++) sub[lo + j] = e;
        }
    });
}

bool gpu_progressive_eligible(const FrameInfo& f)
{
    if (!f.progressive() || f.scans.empty() || f.scans.size() > (size_t)kProgMaxScans) return false;
    if (f.ncomp < 1 || f.ncomp > 4) return false;
    int coef_bits[4][64];
    for (auto& c : coef_bits)
        for (int& b : c) b = -1;
    int stages[4] = {0, 0, 0, 0};
    for (const ScanHeader& sc : f.scans) {
        if (!sc.plain_stuffing || sc.restart_interval != 0 || !sc.rst_after.empty()) return false;
        if ((sc.data_end - sc.data_begin) >= (1ull << 27)) return false;  // bit positions below 2^30: bit 31 flags end-of-band runs
        // a scan without a single entropy-coded byte (a file cut right behind an SOS header, a marker directly after one): no destuff
        // chunk would be queued for it, its stream descriptor would stay empty -- the host decoder names the error (TRUNCATED)
        if (sc.data_end <= sc.data_begin) return false;
        if (sc.ah != 0 && sc.al != sc.ah - 1) return false;
        for (int i = 0; i < sc.ncomp; i++) {
            const int c = sc.comp_index[i];
            if (sc.ss != 0 && coef_bits[c][0] < 0) return false;  // AC before the component's DC scan
            for (int k = sc.ss; k <= sc.se; k++) {
                if (sc.ah == 0 ? coef_bits[c][k] >= 0 : coef_bits[c][k] != sc.ah) return false;  // inconsistent progression
                coef_bits[c][k] = sc.al;
            }
        }
        if (sc.ss == 0) {
            if (sc.ah == 0)
                for (int i = 0; i < sc.ncomp; i++) {
                    const HuffSpec& t = sc.dc[sc.td[i]];
                    const size_t w = prog_table_words(t);
                    if (w == 0 || w > (size_t)kProgTableMax) return false;
                    int nv = 0;
                    for (int l = 1; l <= 16; l++) nv += t.bits[l];
                    for (int v = 0; v < nv; v++)
                        if (t.vals[v] > 15) return false;
                }
        } else {
            const int c = sc.comp_index[0];
            if (++stages[c] > kProgMaxStages) return false;
            const size_t w = prog_table_words(sc.ac[sc.ta[0]]);
            if (w == 0 || w > (size_t)kProgTableMax) return false;
        }
    }
    if (stages[0] + stages[1] + stages[2] + stages[3] > kProgMaxAcScans) return false;  // one wave per AC scan in the walker's workgroup
    for (int c = 0; c < f.ncomp; c++) {
        if (coef_bits[c][0] < 0) return false;  // a component without a DC scan: the host decoder reports the file as incomplete
        if ((size_t)((f.comp[c].samp_w + 7) / 8) * (size_t)((f.comp[c].samp_h + 7) / 8) >= (1u << 24)) return false;
    }
    return true;
}

ProgLdsShape prog_lds_shape(const FrameInfo& f)
{
    ProgLdsShape sh;
    unsigned per_comp[4] = {0, 0, 0, 0};
    for (const ScanHeader& sc : f.scans) {
        if (sc.ss == 0) {
            if (sc.ah != 0) continue;
            sh.dc_slots = std::max<unsigned>(sh.dc_slots, (unsigned)sc.ncomp);
            for (int i = 0; i < sc.ncomp; i++) sh.slot_words = std::max<unsigned>(sh.slot_words, (unsigned)prog_table_words(sc.dc[sc.td[i]]));
        } else {
            per_comp[sc.comp_index[0] & 3]++;
            sh.slot_words = std::max<unsigned>(sh.slot_words, (unsigned)prog_table_words(sc.ac[sc.ta[0]]));
        }
    }
    for (unsigned n : per_comp) {
        sh.ac_waves += n;
        sh.rings += n > 0 ? n - 1 : 0;
    }
    return sh;
}

size_t prog_pool_words(const FrameInfo& f)
{
    size_t words = 0;
    for (const ScanHeader& sc : f.scans) {
        if (sc.ss == 0) {
            if (sc.ah != 0) continue;
            bool seen[4] = {false, false, false, false};
            for (int i = 0; i < sc.ncomp; i++) {
                if (seen[sc.td[i]]) continue;
                seen[sc.td[i]] = true;
                words += align64(prog_table_words(sc.dc[sc.td[i]]));
            }
        } else {
            words += align64(prog_table_words(sc.ac[sc.ta[0]]));
        }
    }
    return words;
}

void fill_prog_image(const FrameInfo& f, ProgImage* im, uint16_t* pool)
{
    memset(im, 0, sizeof *im);
    im->num_scans = (uint32_t)f.scans.size();
    im->ncomp = (uint32_t)f.ncomp;
    im->mcus_x = (uint32_t)f.mcus_x;
    im->mcus_y = (uint32_t)f.mcus_y;
    for (int c = 0; c < f.ncomp; c++) {
        im->comp_h[c] = (uint32_t)f.comp[c].h;
        im->comp_v[c] = (uint32_t)f.comp[c].v;
        im->blocks_w[c] = (uint32_t)f.comp[c].blocks_w;
        im->blocks_h[c] = (uint32_t)f.comp[c].blocks_h;
        im->nbx[c] = (uint32_t)(f.comp[c].samp_w + 7) / 8;
        im->nby[c] = (uint32_t)(f.comp[c].samp_h + 7) / 8;
    }
    size_t used = 0;
    for (size_t s = 0; s < f.scans.size(); s++) {
        const ScanHeader& sc = f.scans[s];
        ProgScan& ps = im->scan[s];
        ps.ncomp = (uint32_t)sc.ncomp;
        ps.comp = (uint32_t)sc.comp_index[0];
        ps.ss = (uint32_t)sc.ss;
        ps.se = (uint32_t)sc.se;
        ps.ah = (uint32_t)sc.ah;
        ps.al = (uint32_t)sc.al;
        for (int i = 0; i < sc.ncomp; i++) ps.comps[i] = (uint32_t)sc.comp_index[i];
        if (sc.ss == 0) {
            im->dc_chain[im->dc_len++] = (uint32_t)s;
            if (sc.ncomp == 1) {
                ps.nblocks = im->nbx[ps.comp] * im->nby[ps.comp];
            } else {
                uint32_t bpm = 0;
                for (int i = 0; i < sc.ncomp; i++) bpm += (uint32_t)(f.comp[sc.comp_index[i]].h * f.comp[sc.comp_index[i]].v);
                ps.nblocks = im->mcus_x * im->mcus_y * bpm;
            }
            if (sc.ah == 0) {
                size_t at[4] = {0, 0, 0, 0};
                bool seen[4] = {false, false, false, false};
                for (int i = 0; i < sc.ncomp; i++) {
                    const int td = sc.td[i];
                    if (!seen[td]) {
                        seen[td] = true;
                        at[td] = used;
                        build_prog_table(sc.dc[td], pool + used);
                        used += align64(prog_table_words(sc.dc[td]));
                    }
                    ps.table[i] = (uint32_t)at[td];
                }
            }
        } else {
            const int c = sc.comp_index[0];
            ps.stage = im->chain_len[c];
            im->chain[c][im->chain_len[c]++] = (uint32_t)s;
            ps.nblocks = im->nbx[c] * im->nby[c];
            ps.table[0] = (uint32_t)used;
            build_prog_table(sc.ac[sc.ta[0]], pool + used);
            used += align64(prog_table_words(sc.ac[sc.ta[0]]));
        }
    }
    im->pool_words = (uint32_t)used;
}

// ---------------------------------------------------------------------------------------------------------- host emulation
namespace {

struct HostStream {
    const uint8_t* bytes = nullptr;
    uint32_t nwords = 0;
    uint32_t word(uint32_t i) const
    {
        if (i >= nwords) return ~0u;
        const uint8_t* p = bytes + (size_t)i * 4;
        return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
    }
};

inline uint32_t host_lookup(const uint16_t* table, uint32_t w)
{
    uint32_t e = table[w >> 24];
    if (e & kProgLong) e = table[(size_t)(e & 0x7FFFu) * 256 + ((w >> 16) & 255u)];
    return e;
}

// the scalar machine of prog_walk_ac on the host
struct HostWalker {
    HostStream st;
    const uint16_t* table;
    uint64_t* hist_all;   // history bitmaps of the component's blocks
    uint32_t* pos_all;    // block_pos of this scan
    uint32_t p = 0;
    uint32_t group = 0;
    int zpos[64];
    int nz = 0;
    uint32_t window() const
    {
        const uint32_t i = p >> 5, sh = p & 31;
        const uint32_t w0 = st.word(i), w1 = st.word(i + 1);
        return sh ? (w0 << sh) | (w1 >> (32 - sh)) : w0;
    }
    void advance(uint32_t n) { p += n; }
    uint32_t pos() const { return p; }
    // prog_walk_ac's view of the symbols "decoded ahead": here every offset is looked up when it is asked for.  Indices wrap at 64
    // like the lane select of the device's v_readlane, so a walk that relied on what an index >= 64 returns would show here too.
    uint32_t wv = 0;      // current window: bits [64 wv, 64 wv + 64)
    bool refine = false;
    uint32_t sym_base() const { return wv * 64u; }
    uint32_t bits_at(uint32_t d)
    {
        p = wv * 64u + (d & 63u);
        return window();
    }
    uint32_t sym_at(uint32_t d) { return host_lookup(table, bits_at(d)); }
    uint32_t fast_at(uint32_t d) { return prog_fast_entry(sym_at(d), refine); }
    template <bool REFINE>
    void sym_window(uint32_t at)
    {
        wv = at >> 6;
    }
    void group_begin(uint32_t g) { group = g; }
    uint64_t hist(int j) const { return hist_all[(size_t)group * kProgGroup + ((unsigned)j & 63u)]; }  // (the walk's first hist(-1) reads lane 63)
    void set_hist(int j, uint64_t h) { hist_all[(size_t)group * kProgGroup + j] = h; }
    void set_pos(int j, uint32_t v) { pos_all[(size_t)group * kProgGroup + j] = v; }
    void group_end(uint32_t) {}
    int zpos_next[64];
    int nz_next = 0;
    void zeros_prepare(uint64_t z)
    {
        nz_next = 0;
        for (int i = 0; i < 64; i++)
            if ((z >> i) & 1) zpos_next[nz_next++] = i;
    }
    void zeros_take()
    {
        nz = nz_next;
        memcpy(zpos, zpos_next, sizeof zpos);
    }
    uint32_t zeros_count() const { return (uint32_t)nz; }
    uint32_t zero_at(uint32_t t) const { return (uint32_t)(zpos[(t - 1u) & 63u] - (int)(t & 63u)); }  // t >= 1: the t-th zero
};

struct HostReplayEnv {
    const HostStream* streams;       // per scan
    const uint16_t* pool;
    const ProgImage* im;
    int16_t* block;                  // device-layout block being built
    uint32_t word(const ProgScan& sc, uint32_t i) const { return streams[&sc - im->scan].word(i); }
    uint32_t lookup(const ProgScan& sc, uint32_t w) const { return host_lookup(pool + sc.table[0], w); }
    int get(int zz) const { return block[kZigzagDeviceGpuHost[zz]]; }
    void put(int zz, int v) const { block[kZigzagDeviceGpuHost[zz]] = (int16_t)v; }
};

}  // namespace

int emulate_gpu_progressive(const uint8_t* data, size_t size, const FrameInfo& f, int16_t* const coef[4])
{
    (void)size;
    std::vector<uint16_t> pool(prog_pool_words(f) + 64);
    ProgImage im;
    fill_prog_image(f, &im, pool.data());
    im.pool = pool.data();
    // destuffed copies of every scan
    std::vector<std::vector<uint8_t>> bytes(f.scans.size());
    std::vector<HostStream> streams(f.scans.size());
    std::vector<uint32_t> total_bits(f.scans.size());
    for (size_t s = 0; s < f.scans.size(); s++) {
        bytes[s].resize(destuffed_capacity(f.scans[s]) + 8);
        const size_t n = destuff_scan(data, f.scans[s], bytes[s].data());
        streams[s].bytes = bytes[s].data();
        streams[s].nwords = (uint32_t)((((n + 3) & ~(size_t)3) + kStreamSlackBytes) / 4);
        total_bits[s] = (uint32_t)n * 8u;
    }
    for (int c = 0; c < f.ncomp; c++) memset(coef[c], 0, (size_t)f.comp[c].blocks_w * f.comp[c].blocks_h * 128);
    // ---- walk: block start positions of every AC scan, stage after stage per component
    std::vector<std::vector<uint32_t>> block_pos(f.scans.size());
    for (int c = 0; c < f.ncomp; c++) {
        const size_t nb = (size_t)im.nbx[c] * im.nby[c];
        std::vector<uint64_t> hist((nb + kProgGroup) & ~(size_t)(kProgGroup - 1), 0);
        for (int stg = 0; stg < (int)im.chain_len[c]; stg++) {
            const int s = im.chain[c][stg];
            ProgScan& sc = im.scan[s];
            block_pos[s].assign((nb + kProgGroup) & ~(size_t)(kProgGroup - 1), 0);
            sc.block_pos = block_pos[s].data();
            HostWalker w;
            w.st = streams[s];
            w.table = pool.data() + sc.table[0];
            w.hist_all = hist.data();
            w.pos_all = block_pos[s].data();
            w.refine = sc.ah != 0;
            if (!prog_walk_ac(w, sc.ss, sc.se, sc.ah, stg + 1 < (int)im.chain_len[c], sc.nblocks, total_bits[s])) return 1;
        }
    }
    // ---- replay: every block of every component
    for (int c = 0; c < f.ncomp; c++) {
        for (uint32_t by = 0; by < im.nby[c]; by++)
            for (uint32_t bx = 0; bx < im.nbx[c]; bx++) {
                HostReplayEnv env{streams.data(), pool.data(), &im, coef[c] + ((size_t)by * im.blocks_w[c] + bx) * 64};
                if (!prog_replay_block(env, im, c, by * im.nbx[c] + bx)) return 1;
            }
    }
    // ---- DC scans in file order
    for (int d = 0; d < (int)im.dc_len; d++) {
        const int s = im.dc_chain[d];
        const ProgScan& sc = im.scan[s];
        const HostStream& st = streams[s];
        uint32_t p = 0;
        auto window = [&]() {
            const uint32_t i = p >> 5, sh = p & 31;
            const uint32_t w0 = st.word(i), w1 = st.word(i + 1);
            return sh ? (w0 << sh) | (w1 >> (32 - sh)) : w0;
        };
        int pred[4] = {0, 0, 0, 0};
        bool ok = true;
        auto one = [&](int i, size_t index) {
            const int c = sc.comps[i];
            int16_t* dcp = coef[c] + index * 64;
            if (sc.ah == 0) {
                const uint32_t e = host_lookup(pool.data() + sc.table[i], window());
                const uint32_t len = e & 31u, sz = (e >> 5) & 255u;
                if (len == 0 || sz > 15) {
                    ok = false;
                    return;
                }
                p += len;
                int diff = 0;
                if (sz) {
                    const uint32_t v = window() >> (32 - sz);
                    p += sz;
                    diff = v < (1u << (sz - 1)) ? (int)v - (int)(1u << sz) + 1 : (int)v;
                }
                pred[i] += diff;
                *dcp = (int16_t)(pred[i] * (1 << sc.al));
            } else {
                if (window() >> 31) *dcp = (int16_t)(*dcp | (1 << sc.al));
                p += 1;
            }
        };
        if (sc.ncomp == 1) {
            const int c = sc.comps[0];
            for (uint32_t by = 0; by < im.nby[c] && ok; by++)
                for (uint32_t bx = 0; bx < im.nbx[c] && ok; bx++) one(0, (size_t)by * im.blocks_w[c] + bx);
        } else {
            for (uint32_t my = 0; my < im.mcus_y && ok; my++)
                for (uint32_t mx = 0; mx < im.mcus_x && ok; mx++)
                    for (uint32_t i = 0; i < sc.ncomp && ok; i++) {
                        const int c = sc.comps[i];
                        for (uint32_t dy = 0; dy < im.comp_v[c] && ok; dy++)
                            for (uint32_t dx = 0; dx < im.comp_h[c] && ok; dx++)
                                one(i, (size_t)(my * im.comp_v[c] + dy) * im.blocks_w[c] + mx * im.comp_h[c] + dx);
                    }
        }
        if (!ok || p > total_bits[s]) return 1;
    }
    return 0;
}

}  // namespace hipjpeg
