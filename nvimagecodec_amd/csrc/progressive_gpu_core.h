// progressive_gpu_core.h -- progressive (SOF2) scans on the GPU entropy stage: data structures and the parse logic shared by
// the kernels (progressive_gpu.hip) and their host emulation (progressive_gpu_host.cpp, used by the CPU tests).
//
// What can and what cannot be parallelised (ITU T.81 Annex G):
//   * DC refinement scans are one raw bit per block: position = block index.  Fully parallel.
//   * DC first / AC first scans are Huffman streams like a baseline scan.
//   * AC REFINEMENT scans interleave Huffman symbols with raw "correction" bits, one for every coefficient of the block that
//     earlier scans left non-zero and that the decoder passes while it counts zero-history coefficients.  How many bits
//     follow a symbol depends on the history of the very block being decoded, i.e. on the block INDEX -- and the index of
//     the block a bit belongs to is only known once everything in front of it has been parsed.  A decoder dropped into the
//     middle of such a scan cannot parse at all, so the self-synchronising subsequence trick of the baseline path
//     (huffman_gpu_core.h) does not apply: the parse of a refinement scan is a sequential chain.  In libjpeg's standard
//     script those scans carry ~3/4 of the bits of a q90 photo.
// Therefore the work is split in two:
//   walk   one WAVE per scan walks it sequentially, as a scalar machine (uniform control flow; per-block history bitmaps and the
//          symbols decoded ahead for a 64-bit window of the stream in registers read with v_readlane), and records nothing but WHERE
//          every block's data starts.  It never touches a coefficient: the correction bits after a symbol are counted from
//          the block's history bitmap with popcounts and a rank/select table, not looped over.  The scans of one component
//          run as a pipeline of waves (scan n+1 needs the history scan n leaves behind, 64 blocks at a time, through an LDS
//          ring), the components and the DC scans on other waves of the image's workgroup, all images at once -- the
//          parallelism that exists: images x components x pipeline stages.
//   replay one LANE per block replays all scans of its block from those positions (now every block is independent), builds
//          the 64 coefficients in LDS and stores the block once, as a whole 128-byte line.
// DC values go to the compact DC planes the IDCT kernels already read for GPU-decoded baseline images.
//
// All arithmetic is integer/bit exact; results are compared with the host entropy decoder and the oracle in tests/.
#pragma once
#include <cstdint>

#include "huffman_gpu_core.h"

#define HJ_LIKELY(x) __builtin_expect(!!(x), 1)
#define HJ_UNLIKELY(x) __builtin_expect(!!(x), 0)

namespace hipjpeg {

constexpr int kProgMaxScans = 24;     // scans per image the GPU path takes (libjpeg's script has 10; more -> host entropy stage)
constexpr int kProgMaxAcScans = 15;   // AC scans of an image the walker takes (a wave each + one for the DC scans: 1024 threads)
constexpr int kProgMaxStages = 6;     // AC scans per component = pipeline stages (= waves) of a walker workgroup
constexpr int kProgChains = 5;        // walker workgroups per image: one per component (up to 4) + one for the DC scans
constexpr int kProgGroup = 64;        // blocks per hand-over unit between pipeline stages (one lane per block)
constexpr int kProgRing = 4;          // groups a stage may run ahead of its successor
constexpr int kProgTableMax = 2560;   // uint16 entries of one scan's lookup table the kernels accept (first level + 9 second-level)
constexpr uint32_t kProgInRun = 0x80000000u;  // block_pos flag: the block lies inside an end-of-band run (no symbols of its own)

// Lookup-table entry (uint16): bits 0-4 code length (0 = no such code), bits 5-12 the symbol byte.  Bit 15 set = the code is
// longer than 8 bits: bits 0-14 = number of the 256-entry second-level table (in units of 256 entries from the table's
// start), indexed by the next 8 bits.
constexpr uint32_t kProgLong = 0x8000u;
HJ_HD constexpr uint32_t prog_entry(uint32_t len, uint32_t sym) { return len | (sym << 5); }

// NB every member that the kernels index with a run-time (wave-uniform) index is a 32-bit word: hipcc folds such an index into
// the offset of a scalar load, and a scalar load from an address that is not dword aligned returns the wrong dword on gfx950
// (DESIGN.md "A compiler hazard"; tests/test_code_object_hazards.py keeps watch).
struct alignas(16) ProgScan {
    const uint8_t* stream;    // destuffed entropy-coded bytes (written by the destuff kernels)
    uint32_t* block_pos;      // AC scans: where each block's data starts (bit offset; kProgInRun set inside an end-of-band run)
    uint32_t huff_image;      // the HuffImage the destuff kernels filled for this scan (total_bits, stream_words)
    uint32_t nblocks;         // AC scans: nbx * nby of the component; DC scans: blocks in the scan
    uint32_t table[4];        // pool offset (uint16 units) of the lookup table: AC scans [0]; DC first scans one per scan component
    uint32_t comps[4];        // DC scans: the components in scan order
    uint32_t ncomp, comp;     // comp: AC scans
    uint32_t ss, se, ah, al;
    uint32_t stage;           // AC scans: position in the component's chain (0 = first)
    uint32_t walk_ticks;      // written by the walk kernel: how long this scan's walk took, in 10 ns ticks (profiling aid)
    uint32_t wait_ticks;      // ... of which the wave spent waiting for its pipeline neighbours (history ring full / empty)
    uint32_t pad_ticks[3];
};

struct alignas(16) ProgImage {
    ProgScan scan[kProgMaxScans];
    uint32_t num_scans;
    uint32_t ncomp, mcus_x, mcus_y;
    uint32_t comp_h[4], comp_v[4];
    uint32_t blocks_w[4], blocks_h[4];  // allocation grid (MCU padded)
    uint32_t nbx[4], nby[4];            // real blocks: ceil(samp / 8)
    int16_t* coef[4];                   // coefficient blocks (device layout), written whole by the replay kernel
    int16_t* dc_plane[4];               // compact DC planes (raster over the allocation grid)
    const uint16_t* pool;               // lookup tables of all scans
    uint32_t pool_words;
    uint32_t status;                    // written by the kernels: 0 ok, 1 = the stream cannot be what it claims to be
    uint32_t chain_len[4];              // AC scans of component c, in file order ...
    uint32_t chain[4][kProgMaxStages];  // ... as indices into scan[]
    uint32_t dc_len, dc_chain[kProgMaxScans];  // DC scans in file order
    uint32_t pad[3];
};

// ---------------------------------------------------------------------------------------------------------- shared helpers
HJ_HD uint64_t prog_band_mask(int ss, int se)
{
    const uint64_t upto = se >= 63 ? ~0ull : ((1ull << (se + 1)) - 1);
    return upto & ~((1ull << ss) - 1);
}
HJ_HD uint64_t prog_from_mask(int k) { return k >= 64 ? 0ull : ~((1ull << k) - 1); }  // bits k..63
HJ_HD int prog_popc64(uint64_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(v);
#else
    return __builtin_popcountll(v);
#endif
}

// ---------------------------------------------------------------------------------------------------------- the walk
// "Fast" view of a lookup-table entry, made once per decoded-ahead symbol (vector work on the device), read by the walk's inner loops:
// a PLAIN coefficient symbol -- first scans: any size s > 0; refinement scans: s == 1 -- becomes (run r + 1) << 16 | bits the symbol
// takes (code + value bits, resp. code + sign bit); everything else (no such code, end of band, run of sixteen, a size a refinement scan
// may not carry) becomes a run no band can hold, so the ONE bound check of the inner loop also sorts those out.
// development aid (tools/walk_laps.sh): where a walking wave's time goes -- lap timers around the parts of prog_walk_scan
#if defined(HJ_WALK_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define HJ_WALK_LAP(w, id) (w).lap(id)
#define HJ_WALK_COUNT(n) ((n)++)
#else
#define HJ_WALK_LAP(w, id) ((void)0)
#define HJ_WALK_COUNT(n) ((void)0)
#endif

// The one non-plain symbol that is common -- "end of band, no run": how most blocks end -- keeps a tag of its own and its length, so that the
// walk recognises it by the fast view alone.
constexpr uint32_t kProgNotPlain = 64u << 16;
constexpr uint32_t kProgEndOfBandTag = 65u;  // fast view (kProgEndOfBandTag << 16) | code length
HJ_HD uint32_t prog_fast_entry(uint32_t e, bool refine)
{
    const uint32_t len = e & 31u, s = (e >> 5) & 15u, r = (e >> 9) & 15u;
    const bool plain = refine ? s == 1u : s != 0u;
    const uint32_t other = (s == 0u && r == 0u && len != 0u) ? (kProgEndOfBandTag << 16) | len : kProgNotPlain;
    return plain ? ((r + 1u) << 16) | (len + s) : other;
}
HJ_HD uint64_t prog_bit_set(uint64_t h, uint32_t t)  // t < 64
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("s_bitset1_b64 %0, %1" : "+s"(h) : "s"(t));  // the compiler spells it shift + or
    return h;
#else
    return h | (1ull << (t & 63u));
#endif
}

// Walks one AC scan and records the start of every block.  `W` supplies the machine.  Symbols are decoded ahead a WINDOW at a time:
// for all 64 bit offsets of an aligned 64-bit stretch of the stream at once, whatever code would start there.
//   uint32_t sym_base()            bit position the current window starts at
//   uint32_t fast_at(uint32_t d)   prog_fast_entry of the code that starts at bit sym_base() + d; d >= 64 returns anything
//   uint32_t sym_at(uint32_t d)    its lookup-table entry (0 = no such code), d < 64
//   uint32_t bits_at(uint32_t d)   the 32 bits of the stream from bit sym_base() + d on, MSB first, d < 64
//   void sym_window<REFINE>(p)     make the window that holds bit p the current one (REFINE: the fast views are a refinement scan's)
//   void group_begin(uint32_t g)   history bitmaps of blocks [64 g, 64 g + 64) become available (waits for the previous stage)
//   uint64_t hist(int j) / void set_hist(int j, uint64_t)   history bitmap of block j of the current group
//   void set_pos(int j, uint32_t)  block_pos of block j of the current group
//   void group_end(uint32_t g)     stores the positions, hands the bitmaps to the next stage
//   void zeros_prepare(uint64_t z) / void zeros_take()   rank/select over the set bits of z, prepared one block ahead, then made current:
//                                  uint32_t zeros_count(),  uint32_t zero_at(uint32_t t) = (position of the t-th set bit, t >= 1) - t
// Everything is wave-uniform on the device; the position is base + d, locals here.  A lone wave pays ~9 cycles per dependent scalar
// instruction and ~24 more per taken branch or VGPR->SGPR read (tools/scalar_chain_rate.hip), so what counts is the number of instructions
// on the path a PLAIN symbol takes and on the path from block to block:
//   * the coefficient symbols -- nearly all symbols of the scans that bound a batch's latency, the last refinement scans --
//     have an inner loop with ONE exit: window exhausted (d >= 64 is added to the run), end of band, run of sixteen, a broken stream and
//     "band complete" all show as a run past the band's end and are told apart behind the exit.  In a refinement scan the position of the
//     next coefficient is not tracked at all: with G[t] = zero_at(t) (one less than the number of non-zero-history coefficients in front
//     of the t-th zero-history one), a new coefficient behind r more zeros costs  code + sign + (G[zr + r + 1] - G_prev)  bits, and k is
//     G_prev + 1 + zr whenever somebody asks;
//   * everything else is ONE loop per group of 64 blocks, an iteration per event (block complete, window exhausted, run of sixteen, end
//     of band): the machine's vector state is carried by that loop alone.  (Nested block / symbol loops made the compiler copy the whole state -- twenty registers
//     -- into and out of every block, and wait for the lookups of the next window right behind their issue.)
// Returns false when the stream breaks the rules (the host decoder then takes the image and names the error).
// TRACK: another scan of the component follows and needs the history this one leaves behind (the last scan of a component -- the longest
// walks are such -- does not keep it up to date: two instructions per symbol less).
template <bool REFINE, bool TRACK, class W>
HJ_HD bool prog_walk_scan(W& w, int ss, int se, uint32_t nblocks, uint32_t total_bits)
{
    if (nblocks == 0) return true;
    const uint64_t band = prog_band_mask(ss, se);
    const uint32_t last = (uint32_t)se;
    uint32_t eobrun = 0;
    uint32_t base = w.sym_base();
    uint32_t d = 0;  // the position is base + d
    uint32_t g = 0, j = 0;
    uint32_t n = nblocks < (uint32_t)kProgGroup ? nblocks : (uint32_t)kProgGroup;
    // the current block
    uint64_t h = 0;
    uint32_t c = 0;             // first scans: position of the last coefficient placed (k - 1)
    uint32_t zr = 0, gprev = 0; // refinement scans: zero-history coefficients of the band already passed; G of the last one placed
    uint32_t nz = 0;
    bool skip = false;          // the block lies inside an end-of-band run: nothing to parse

    // Called from ONE place (a second call site would make the compiler merge the two in-flight rank/select tables with a copy, and wait
    // for the crossbar right behind its issue): the loop below starts one block early, on a block that does not exist.
    uint64_t hnext = 0;  // refinement scans: the history of the block after the current one (read once, for the preparation and for the block)
    auto block_start = [&]() {
        if (REFINE) {
            h = hnext;
            w.zeros_take();
            // the next block's, in the shadow of this block's walk (behind the group's last block: of whatever lane 0 holds, never used --
            // a condition here would cost a copy of the table and a wait for the crossbar)
            hnext = w.hist((int)j + 1);
            w.zeros_prepare(~hnext & band);
        } else {
            h = w.hist((int)j);
        }
        if (HJ_UNLIKELY(j == ~0u)) {
            skip = true;  // the block in front of the group's first: nothing but the preparation of block 0
        } else if (HJ_UNLIKELY(eobrun != 0)) {
            // inside an end-of-band run: a first scan has nothing for this block, a refinement scan one correction bit per coefficient of
            // the band that is already non-zero
            w.set_pos((int)j, (base + d) | kProgInRun);
            if (REFINE) d += (uint32_t)prog_popc64(h & band);
            eobrun--;
            skip = true;
        } else {
            w.set_pos((int)j, base + d);
            skip = false;
            if (REFINE) {
                nz = w.zeros_count();
                zr = 0;
                gprev = (uint32_t)ss - 1u;
            } else {
                c = (uint32_t)ss - 1u;
            }
        }
    };

    // Per group of 64 blocks: ONE loop, an iteration per event, left only at its bottom (no break, no return inside: every way out
    // of the middle costs flag registers and branches on the way of the events that stay).
    for (;;) {
        w.group_begin(g);
        j = ~0u - 1u;
        skip = true;
        uint32_t run = 1;  // 1 = walking, 0 = the group is complete, 2 = the stream broke the rules
        do {
            bool next = true;
            if (HJ_LIKELY(!skip)) {
                uint32_t placed;  // position of the last coefficient placed or passed (the next one to look at is placed + 1)
                uint32_t f;  // fast view of the symbol at the position (the one the coefficient loop stopped at)
                if (!REFINE) {
                    // plain coefficients: run of r zeros, then a coefficient of s bits
                    f = w.fast_at(d);
                    uint32_t kk = c + (f >> 16) + (d & ~63u);
                    while (HJ_LIKELY(kk <= last)) {  // (two symbols per turn: every second one saves the taken branch back to the top)
                        HJ_WALK_COUNT(w.lap_syms);
                        if (TRACK) h = prog_bit_set(h, kk);
                        d += f & 0xFFFFu;
                        c = kk;
                        f = w.fast_at(d);
                        kk = c + (f >> 16) + (d & ~63u);
                        if (HJ_UNLIKELY(kk > last)) break;
                        HJ_WALK_COUNT(w.lap_syms);
                        if (TRACK) h = prog_bit_set(h, kk);
                        d += f & 0xFFFFu;
                        c = kk;
                        f = w.fast_at(d);
                        kk = c + (f >> 16) + (d & ~63u);
                    }
                    placed = c;
                } else {
                    // plain new coefficients: behind r zero-history coefficients; code, sign bit, then one correction bit for every
                    // non-zero-history coefficient passed on the way
                    uint32_t q = d - gprev;
                    f = w.fast_at(d);
                    uint32_t t = zr + (f >> 16) + (d & ~63u);
                    while (HJ_LIKELY(t <= nz)) {  // (two symbols per turn: every second one saves the taken branch back to the top)
                        HJ_WALK_COUNT(w.lap_syms);
                        gprev = w.zero_at(t);
                        q += f & 0xFFFFu;
                        d = q + gprev;
                        zr = t;
                        if (TRACK) h = prog_bit_set(h, gprev + t);  // the coefficient's position
                        f = w.fast_at(d);
                        t = zr + (f >> 16) + (d & ~63u);
                        if (HJ_UNLIKELY(t > nz)) break;
                        HJ_WALK_COUNT(w.lap_syms);
                        gprev = w.zero_at(t);
                        q += f & 0xFFFFu;
                        d = q + gprev;
                        zr = t;
                        if (TRACK) h = prog_bit_set(h, gprev + t);
                        f = w.fast_at(d);
                        t = zr + (f >> 16) + (d & ~63u);
                    }
                    placed = gprev + zr;
                }
                HJ_WALK_LAP(w, 0);
                if (placed >= last) {
                    // the band is complete
                } else if (d >= 64u) {
                    w.template sym_window<REFINE>(base + d);
                    const uint32_t nb = w.sym_base();
                    d = base + d - nb;
                    base = nb;
                    next = false;
                    HJ_WALK_LAP(w, 2);
                } else if (HJ_LIKELY((f >> 16) == kProgEndOfBandTag)) {
                    // end of band, no run -- how most blocks end.  Refinement scans: + a correction bit for every non-zero-history
                    // coefficient in the rest of the band (the band has se - ss + 1 - nz of them, gprev + 1 - ss lie in front of the position)
                    d += f & 0xFFFFu;
                    if (REFINE) d += last - nz - gprev;
                } else {
                    const uint32_t e = w.sym_at(d);
                    const uint32_t len = e & 31u, r = (e >> 9) & 15u, s = (e >> 5) & 15u;
                    if (HJ_UNLIKELY(len == 0 || s != 0)) {
                        // no such code / a run past the band's end (resp. past the last zero) / refinement: a size other than 1
                        run = 2;
                        next = false;
                    } else if (r == 15) {
                        if (!REFINE) {
                            c += 16;
                            d += len;
                            next = c + 1u > last;
                        } else {
                            // sixteen zero-history coefficients are skipped (or the rest of the band, if it has fewer)
                            const uint32_t t = zr + 16;
                            if (t > nz) {
                                d += len + (last - nz - gprev);  // + the rest of the band (see below)
                            } else {
                                const uint32_t gz = w.zero_at(t);
                                d += len + gz - gprev;
                                gprev = gz;
                                zr = t;
                                next = gz + t >= last;
                            }
                        }
                    } else {
                        eobrun = (1u << r) - 1u + (r ? (w.bits_at(d) << len) >> (32 - r) : 0u);  // this block is the run's first
                        d += len + r;
                        // + a correction bit for every non-zero-history coefficient in the rest of the band: the band has se - ss + 1 - nz of
                        // them, gprev + 1 - ss lie in front of the position (position - ss places, zr of them zero-history)
                        if (REFINE) d += last - nz - gprev;
                    }
                }
                if (TRACK && next) w.set_hist((int)j, h);
                HJ_WALK_LAP(w, 1);
            }
            if (next) {
                j++;
                if (j == n)
                    run = 0;
                else
                    block_start();
                HJ_WALK_LAP(w, 3);
            }
        } while (run == 1);
        w.group_end(g);
        if (run == 2) return false;
        // once per 64 blocks (wave-uniform, nearly free): a walk that has left the data stops here instead of decoding the ones behind
        // the end as symbols for the rest of a forged frame; also keeps the 32-bit position far from kProgInRun's bit
        if (HJ_UNLIKELY(base + d > total_bits)) return false;
        g++;
        if (g * (uint32_t)kProgGroup >= nblocks) return true;
        n = (nblocks - g * kProgGroup) < (uint32_t)kProgGroup ? (nblocks - g * kProgGroup) : (uint32_t)kProgGroup;
    }
}

template <class W>
HJ_HD bool prog_walk_ac(W& w, int ss, int se, int ah, bool track, uint32_t nblocks, uint32_t total_bits)
{
    if (ah) return track ? prog_walk_scan<true, true>(w, ss, se, nblocks, total_bits) : prog_walk_scan<true, false>(w, ss, se, nblocks, total_bits);
    return track ? prog_walk_scan<false, true>(w, ss, se, nblocks, total_bits) : prog_walk_scan<false, false>(w, ss, se, nblocks, total_bits);
}

// ---------------------------------------------------------------------------------------------------------- the replay
// One block, all AC scans of its component, from the recorded positions.  `E` supplies per-lane accessors:
//   uint32_t word(const ProgScan&, uint32_t i)          32-bit word i of the scan's stream, MSB first (ones behind the end)
//   uint32_t lookup(const ProgScan&, uint32_t w)        lookup-table entry for the window w (next 32 bits, MSB first)
//   int get(int zz) / void put(int zz, int v)           coefficient at ZIGZAG index zz of the block being built
// Returns false for a stream that breaks the rules.  `block` = index of the block in scan order of its component.
struct ProgBits {  // MSB-first reader over E::word, positions in bits
    uint32_t hi, lo, nbits, next;
    template <class E>
    HJ_HD void start(const E& e, const ProgScan& sc, uint32_t pos)
    {
        const uint32_t i = pos >> 5, sh = pos & 31;
        const uint32_t w0 = e.word(sc, i), w1 = e.word(sc, i + 1);
        hi = (w0 << sh) | ((w1 >> 1) >> (sh ^ 31));
        lo = w1 << sh;
        nbits = 64 - sh;
        next = i + 2;
    }
    template <class E>
    HJ_HD void consume(const E& e, const ProgScan& sc, uint32_t c)  // c <= 31
    {
        if (c == 0) return;
        hi = (hi << c) | (lo >> (32 - c));
        lo <<= c;
        nbits -= c;
        if (nbits < 32) {
            const uint32_t f = e.word(sc, next++);
            hi |= f >> (nbits & 31);
            lo = nbits ? f << (32 - nbits) : f;  // nbits >= 1 here whenever c <= 31 and nbits was >= 32
            nbits += 32;
        }
    }
    HJ_HD uint32_t peek(uint32_t n) const { return n ? hi >> (32 - n) : 0u; }  // n <= 31
};

template <class E>
HJ_HD bool prog_replay_block(E& env, const ProgImage& im, int comp, uint32_t block)
{
    uint64_t nonzero = 0;  // bit zz set = coefficient zz is non-zero so far
    for (int st = 0; st < (int)im.chain_len[comp]; st++) {
        const ProgScan& sc = im.scan[im.chain[comp][st]];
        const uint32_t bp = ((const uint32_t*)sc.block_pos)[block];
        const int ss = sc.ss, se = sc.se, al = sc.al;
        const uint64_t band = prog_band_mask(ss, se);
        ProgBits br;
        if (sc.ah == 0) {
            if (bp & kProgInRun) continue;
            br.start(env, sc, bp);
            int k = ss;
            while (k <= se) {
                const uint32_t e = env.lookup(sc, br.hi);
                const uint32_t len = e & 31u, r = (e >> 9) & 15u, s = (e >> 5) & 15u;
                if (len == 0) return false;
                br.consume(env, sc, len);
                if (s) {
                    k += (int)r;
                    if (k > se) return false;
                    const uint32_t v = br.peek(s);
                    br.consume(env, sc, s);
                    const int val = v < (1u << (s - 1)) ? (int)v - (int)(1u << s) + 1 : (int)v;
                    env.put(k, val * (1 << al));
                    nonzero |= 1ull << k;
                    k++;
                } else if (r == 15) {
                    k += 16;
                } else {
                    break;  // end of band (the run length concerns the following blocks; the walk has dealt with it)
                }
            }
        } else {
            const int p1 = 1 << al, m1 = -(1 << al);
            br.start(env, sc, bp & ~kProgInRun);
            // one correction bit for coefficient zz (non-zero history): T.81 G.1.2.3
            auto correct = [&](int zz) {
                const uint32_t bit = br.peek(1);
                br.consume(env, sc, 1);
                if (bit) {
                    const int c = env.get(zz);
                    if ((c & p1) == 0) env.put(zz, c >= 0 ? c + p1 : c + m1);
                }
            };
            auto correct_range = [&](int from, int to) {  // every non-zero-history coefficient in [from, to)
                uint64_t m = nonzero & band & prog_from_mask(from) & ~prog_from_mask(to);
                while (m) {
#if defined(__HIP_DEVICE_COMPILE__)
                    const int zz = __ffsll((unsigned long long)m) - 1;
#else
                    const int zz = __builtin_ctzll(m);
#endif
                    m &= m - 1;
                    correct(zz);
                }
            };
            if (bp & kProgInRun) {
                correct_range(ss, se + 1);
                continue;
            }
            int k = ss;
            while (k <= se) {
                const uint32_t e = env.lookup(sc, br.hi);
                const uint32_t len = e & 31u, r = (e >> 9) & 15u, s = (e >> 5) & 15u;
                if (len == 0 || s > 1) return false;
                br.consume(env, sc, len);
                if (s == 1 || r == 15) {
                    int newval = 0;
                    if (s == 1) {
                        newval = br.peek(1) ? p1 : m1;
                        br.consume(env, sc, 1);
                    }
                    // the (r+1)-th zero-history coefficient at or behind k
                    uint64_t z = ~nonzero & band & prog_from_mask(k);
                    int t = se + 1;
                    for (uint32_t i = 0; i <= r && z; i++) {
#if defined(__HIP_DEVICE_COMPILE__)
                        const int p = __ffsll((unsigned long long)z) - 1;
#else
                        const int p = __builtin_ctzll(z);
#endif
                        z &= z - 1;
                        if (i == r) t = p;
                    }
                    if (t > se && s == 1) return false;  // fewer zero-history coefficients left than the run says
                    correct_range(k, t > se ? se + 1 : t);
                    if (s == 1) {
                        env.put(t, newval);
                        nonzero |= 1ull << t;  // positions below the new k are never looked at again in this scan
                    }
                    k = t + 1;
                } else {
                    br.consume(env, sc, r);  // the run length's extra bits (the walk has used them)
                    correct_range(k, se + 1);
                    break;
                }
            }
        }
    }
    return true;
}

}  // namespace hipjpeg
