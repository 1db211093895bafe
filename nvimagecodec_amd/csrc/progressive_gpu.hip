// progressive_gpu.hip -- gfx950 kernels for progressive (SOF2) scans on the GPU entropy stage.  Algorithm, data structures and
// the reasons for the split into a sequential WALK and a parallel REPLAY: progressive_gpu_core.h.
//
//   prog_walk_kernel    grid = images: one workgroup per image, a wave for the DC scans and a wave per AC scan (the last stages of
//                       the components first: they are the long ones and want SIMDs of their own).  The AC scans of a component are
//                       pipelined 64 blocks at a time through LDS rings that carry the blocks' history bitmaps from scan to scan.
//                       A wave runs as a scalar machine: control flow, position, history bitmap, zigzag state are uniform (SGPRs);
//                       what it indexes -- the symbols decoded ahead for the 64 bit offsets of the current window of the stream, 64
//                       history bitmaps, the rank/select table of the current block, the stream words -- sits in VGPRs and is read
//                       with v_readlane.  DC first scans: the same windows, one loop over the scan's blocks; DC refinement scans:
//                       64 blocks per step.
//   prog_replay_kernel  one lane per block: all AC scans of the block from the recorded positions, coefficients assembled in
//                       LDS, blocks stored as whole 128-byte lines (eight lanes per block).
#include <hip/hip_runtime.h>

#include "gpu_huffman.h"
#include "progressive_gpu.h"
#include "progressive_gpu_core.h"

namespace hipjpeg {

namespace {

#define HJ_LDS __attribute__((address_space(3)))
#define HJ_GLOBAL __attribute__((address_space(1)))
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// How long a wave that waits for its pipeline neighbour sleeps between two looks at the counter, in units of 64 cycles.  A poll is a dozen
// instructions through the compute unit's one scalar ALU, which the walking waves of up to three images share.
#ifndef HJ_WALK_SLEEP
#define HJ_WALK_SLEEP 2
#endif
constexpr int kWalkMaxWaves = 16;  // one workgroup per image: a wave for the DC scans and one per AC scan (gpu_progressive_eligible)
constexpr int kWalkThreads = 64 * kWalkMaxWaves;

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t lane_read(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// One workgroup walks ALL scans of an image: a wave for the DC scans, a wave per AC scan.  (A workgroup per (image, component)
// held 16 KB of LDS for the two scans of a chroma chain as for the four of luma -- 66 KB per image, three batches filled the
// compute units; now 27 KB per image and six batches' walks run side by side.  It did not buy throughput: the walk is scalar
// instructions, a compute unit has one scalar ALU for its four SIMDs, and three batches' walks already saturate it.)
// The history bitmaps a stage hands to its successor live in rings in dynamic LDS (one ring per hand-over, behind the tables).
struct WalkShared {
    uint32_t done[kWalkMaxWaves];   // groups the wave's scan has published
    uint32_t taken[kWalkMaxWaves];  // groups the wave's scan has taken over from its predecessor
    uint32_t abort_flag;
};
typedef unsigned long long WalkRing[kProgRing][kProgGroup];
static_assert(sizeof(WalkRing) == (size_t)kProgRing * kProgGroup * 8, "progressive_gpu_host.h kProgRingBytes");

// The words of one scan's stream, 192 of them at a time in three VGPRs.
struct WordStream {
    const HJ_GLOBAL uint32_t* g;
    uint32_t nwords;
    uint32_t lane;
    uint32_t W, X, Y;  // W: words vbase .. vbase + 63 (lane j = word vbase + j, MSB first), X: the 64 after those, Y: the 64 after X, as
    uint32_t vbase;    // loaded, still in flight -- requested 4096 bits before they are used

    // 64 words from word `base` on, as they lie in memory (the address is clamped instead of branched around, so that the
    // load needs no exec mask and nothing waits for it here); swap() turns them MSB first and fills in ones behind the end
    __device__ __forceinline__ uint32_t fetch_raw(uint32_t base) const
    {
        const uint32_t i = base + lane;
        if (nwords == 0) return ~0u;  // an empty stream (never queued by the host: gpu_progressive_eligible) reads as ones, not as g[-1]
        return g[i < nwords ? i : nwords - 1];
    }
    __device__ __forceinline__ uint32_t swap(uint32_t raw, uint32_t base) const { return base + lane < nwords ? __builtin_bswap32(raw) : ~0u; }
    __device__ __forceinline__ void open(const uint32_t* stream, uint32_t stream_words)
    {
        g = (const HJ_GLOBAL uint32_t*)stream;
        nwords = stream_words;
        load(0);
    }
    __device__ __forceinline__ void load(uint32_t base)
    {
        vbase = base;
        W = swap(fetch_raw(base), base);
        X = swap(fetch_raw(base + 64), base + 64);
        Y = fetch_raw(base + 128);
    }
    // word `index` (>= vbase + 64) becomes one of W's: the usual step (the next 64 words become the current ones) or a long skip
    __device__ __forceinline__ void seek(uint32_t index)
    {
        if (index - vbase < 128u) {
            vbase += 64;
            W = X;
            X = swap(Y, vbase + 64);
            Y = fetch_raw(vbase + 128);
        } else {
            load(index & ~63u);
        }
    }
};

// The 32 stream bits that start at each of the 64 bit offsets of window `target` (bits [64 target, 64 target + 64)): lane l = the bits
// from 64 target + l on, MSB first.  Three readlanes for the window's words, a funnel shift.
__device__ __forceinline__ uint32_t window_bits(WordStream& ws, uint32_t target)
{
    uint32_t i0 = 2u * target - ws.vbase;  // the window's words: i0, i0 + 1 and (for the codes that start in its second half) i0 + 2
    if (__builtin_expect(i0 >= 64u, 0)) {  // once per 32 windows
        ws.seek(2u * target);
        i0 = 2u * target - ws.vbase;
    }
    const uint32_t w0 = lane_read(ws.W, i0), w1 = lane_read(ws.W, i0 + 1u);
    const uint32_t w2 = i0 == 62u ? lane_read(ws.X, 0) : lane_read(ws.W, i0 + 2u);
    const uint32_t hi = ws.lane < 32u ? w0 : w1, lo = ws.lane < 32u ? w1 : w2;
    return (uint32_t)(((((unsigned long long)hi << 32) | lo) << (ws.lane & 31u)) >> 32);
}

// Reader of the DC first scans.  Like the AC walker it decodes ahead one aligned 64-bit window at a time -- for every bit offset of the
// window and every DC table of the scan, the code that would start there -- so that a DC symbol costs the walk two readlanes (entry,
// bits) instead of a window assembly and a table lookup in LDS it has to wait for.
struct DcWalker : WordStream {
    uint32_t wv;                  // the current window
    uint32_t d;                   // the position is 64 wv + d
    uint32_t win;                 // per lane l: the 32 bits from bit 64 wv + l on
    uint32_t FA, FB;              // per lane l: len | size << 5 of the code at bit 64 wv + l (0 = no such code), 16 bits per table slot:
                                  // slots 0 and 1 in FA, 2 and 3 in FB
    const HJ_LDS uint16_t* tables;
    uint32_t slot_words;
    uint32_t need;                // bit j: table slot j is looked up (slots that repeat an earlier slot's table are not)

    __device__ __forceinline__ uint32_t lookup(uint32_t j, uint32_t w) const
    {
        const HJ_LDS uint16_t* t = tables + j * slot_words;
        uint32_t e = t[w >> 24];
        if (e & kProgLong) e = t[(e & 0x7FFFu) * 256u + ((w >> 16) & 255u)];
        return e & 0x1FFFu;
    }
    __device__ __forceinline__ void decode(uint32_t target)
    {
        wv = target;
        win = window_bits(*this, target);
        FA = lookup(0, win);
        if (need & 2u) FA |= lookup(1, win) << 16;
        FB = 0;
        if (need & 4u) FB = lookup(2, win);
        if (need & 8u) FB |= lookup(3, win) << 16;
    }
    __device__ __forceinline__ void start(const uint32_t* stream, uint32_t stream_words)
    {
        open(stream, stream_words);
        d = 0;
        decode(0);
    }
    __device__ __forceinline__ uint32_t pos() const { return wv * 64u + d; }
    // One DC symbol with the table in slot shift16 / 16: false = no such code (the position then stays where it is).  *diff = the
    // difference it carries.  No way out of the middle, no branch but the window's: the caller collects the verdicts.
    __device__ __forceinline__ bool symbol(uint32_t shift16, int* diff)
    {
        if (__builtin_expect(d >= 64u, 0)) {
            const uint32_t at = wv * 64u + d;
            decode(at >> 6);
            d = at & 63u;
        }
        const unsigned long long both = ((unsigned long long)lane_read(FB, d) << 32) | lane_read(FA, d);
        const uint32_t f = (uint32_t)(both >> shift16) & 0xFFFFu;
        const uint32_t len = f & 31u, sz = f >> 5;  // sz <= 15: gpu_progressive_eligible() keeps other tables away
        // code (<= 16 bits) and value (<= 15 bits) lie in the same 32-bit window; sz == 0: v = 0 < 2^31 gives 0 - 1 + 1
        const uint32_t v = (uint32_t)(((unsigned long long)(lane_read(win, d) << len) << sz) >> 32);
        *diff = v < (1u << ((sz - 1u) & 31u)) ? (int)v - (int)(1u << sz) + 1 : (int)v;
        d += len + sz;
        return len != 0;
    }
};

// The scalar machine of prog_walk_ac (progressive_gpu_core.h) on one wave.
// Symbols are decoded ahead one 64-bit WINDOW of the stream at a time: lane l = whatever code starts at bit 64 wv + l (its 32 stream
// bits, its lookup-table entry, the entry's fast view) -- vector work: three readlanes for the window's words, a funnel shift, one or
// two table lookups in LDS.  The windows are aligned, so the next one is known before the walk gets there: its words are picked and its
// first-level lookups issued when the walk ENTERS the current window, and nothing waits for them until it leaves it.
struct DevWalker : WordStream {
    uint32_t wv;                       // the current window: bits [64 wv, 64 wv + 64)
    uint32_t cur, cur_fast, cur_win;   // per lane l, for the code at bit 64 wv + l: table entry, prog_fast_entry of it, the 32 bits from there on
    uint32_t nxt, nxt_win;             // window wv + 1: first-level entry (lookup in flight), bits
    bool refine;                       // a refinement scan (decides what counts as a plain symbol)
    const HJ_LDS uint16_t* table;      // the scan's full lookup table
    // per group
    uint32_t hlo, hhi, pos_out;        // per lane: block 64 g + lane
    uint32_t zpos, zraw_next;          // lane t >= 1: (position of the t-th zero-history coefficient of the current block) - t; the next block's positions
    uint32_t nz, nz_next;
    // pipeline
    WalkShared* sh;
    bool has_prev, has_next;                 // the scan refines what another scan of the component left / is refined by another
    HJ_LDS WalkRing* ring_in;                // history from the predecessor; ring_in + 1 = ring to the successor
    int self;                                // this scan's index into done[] / taken[] (predecessor self - 1, successor self + 1)
    uint32_t* block_pos;
    uint32_t nblocks;
    bool aborted;
    uint32_t waited;                         // 10 ns ticks spent in wait_for (profiling aid)

    // first-level decode of window `target`; the lookups stay in flight
    __device__ __forceinline__ void issue(uint32_t target)
    {
        const uint32_t win = window_bits(*this, target);
        nxt_win = win;
        nxt = table[win >> 24];
    }
    // window wv + 1 becomes the current one
    __device__ __forceinline__ void finish(bool refinement)
    {
        uint32_t e = nxt;
        if (e & kProgLong) e = table[(e & 0x7FFFu) * 256u + ((nxt_win >> 16) & 255u)];
        cur = e;
        cur_win = nxt_win;
        cur_fast = prog_fast_entry(e, refinement);
    }
    __device__ __forceinline__ void start(const uint32_t* stream, uint32_t stream_words)
    {
        open(stream, stream_words);
        issue(0);
        finish(refine);
        wv = 0;
        issue(1);
    }
    // prog_walk_ac's view (progressive_gpu_core.h)
    __device__ __forceinline__ uint32_t sym_base() const { return wv * 64u; }
    __device__ __forceinline__ uint32_t fast_at(uint32_t d) const { return lane_read(cur_fast, d); }
    __device__ __forceinline__ uint32_t sym_at(uint32_t d) const { return lane_read(cur, d); }
    __device__ __forceinline__ uint32_t bits_at(uint32_t d) const { return lane_read(cur_win, d); }
    template <bool REFINE>
    __device__ __forceinline__ void sym_window(uint32_t at)
    {
        const uint32_t target = at >> 6;
        if (target == wv) return;
        if (__builtin_expect(target != wv + 1u, 0)) issue(target);  // a long end-of-band run has skipped windows
        finish(REFINE);
        wv = target;
        issue(target + 1u);
    }
    __device__ __forceinline__ bool wait_for(const uint32_t* counter, uint32_t above)
    {
        unsigned long long w0 = 0;
        bool slept = false;
        while (uni(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) <= above) {
            if (uni(__hip_atomic_load(&sh->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) {
                aborted = true;
                return false;
            }
            if (!slept) {
                slept = true;
                w0 = wall_clock64();
            }
            __builtin_amdgcn_s_sleep(HJ_WALK_SLEEP);
        }
        if (slept) waited += (uint32_t)(wall_clock64() - w0);
        return true;
    }
    __device__ __forceinline__ void group_begin(uint32_t gi)
    {
        hlo = hhi = 0;
        pos_out = 0;
        if (has_prev && !aborted) {
            if (wait_for(&sh->done[self - 1], gi)) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const unsigned long long h = (*ring_in)[gi % kProgRing][lane];
                hlo = (uint32_t)h;
                hhi = (uint32_t)(h >> 32);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(&sh->taken[self], gi + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __device__ __forceinline__ unsigned long long hist(int j) const
    {
        return ((unsigned long long)lane_read(hhi, (uint32_t)j) << 32) | lane_read(hlo, (uint32_t)j);
    }
    __device__ __forceinline__ void set_hist(int j, unsigned long long h)
    {
        const bool me = lane == (uint32_t)j;  // one compare, two selects (this toolchain has no writelane builtin)
        hlo = me ? (uint32_t)h : hlo;
        hhi = me ? (uint32_t)(h >> 32) : hhi;
    }
    __device__ __forceinline__ void set_pos(int j, uint32_t v) { pos_out = lane == (uint32_t)j ? v : pos_out; }
    __device__ __forceinline__ void group_end(uint32_t gi)
    {
        const uint32_t b = gi * kProgGroup + lane;
        if (b < nblocks) ((HJ_GLOBAL uint32_t*)block_pos)[b] = pos_out;
        if (has_next && !aborted) {
            // the slot is free once the next stage has taken group gi - kProgRing
            if (gi >= (uint32_t)kProgRing && !wait_for(&sh->taken[self + 1], gi - kProgRing)) return;
            (*(ring_in + 1))[gi % kProgRing][lane] = ((unsigned long long)hhi << 32) | hlo;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(&sh->done[self], gi + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    // rank/select over the set bits of z, for the NEXT block: the crossbar write stays in flight until zeros_take().  The t-th zero
    // (t >= 1) lands in lane t.
    __device__ __forceinline__ void zeros_prepare(unsigned long long z)
    {
        const uint32_t zl = (uint32_t)z, zh = (uint32_t)(z >> 32);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(zh, __builtin_amdgcn_mbcnt_lo(zl, 0u));  // set bits of z below this lane
        nz_next = uni((uint32_t)__popcll(z));
        const bool mine = (z >> lane) & 1ull;
        const uint32_t target = mine ? rank : nz_next + (lane - rank);  // a full permutation: zeros first, in order
        zraw_next = (uint32_t)__builtin_amdgcn_ds_permute((int)(((target + 1u) & 63u) * 4u), (int)lane);
    }
    __device__ __forceinline__ void zeros_take()
    {
        zpos = zraw_next - lane;  // lane t, 1 <= t <= nz: (position of the t-th zero) - t
        nz = uni(nz_next);
    }
#if defined(HJ_WALK_PROFILE)
    // HJ_WALK_PROFILE=1: lap timers around the parts of prog_walk_scan (0 fast loops, 1 event handling, 2 window switches, 3 block starts);
    // =2: the fast loops alone, by the number of symbols a visit took (0, 1, 2, more)
    unsigned long long lap_last;
    uint32_t lap_sum[4], lap_n[4], lap_syms;
    __device__ __forceinline__ void lap(int id)
    {
        unsigned long long now = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(now));  // (keeps the two stamps apart: the second is taken when the first has arrived)
        const uint32_t dt = (uint32_t)(now - lap_last);
        if (HJ_WALK_PROFILE == 1) {
            lap_sum[id] += dt;
            lap_n[id]++;
        } else if (id == 0) {
            const uint32_t b = lap_syms < 3u ? lap_syms : 3u;
            lap_sum[0] += b == 0 ? dt : 0;
            lap_sum[1] += b == 1 ? dt : 0;
            lap_sum[2] += b == 2 ? dt : 0;
            lap_sum[3] += b == 3 ? dt : 0;
            lap_n[0] += b == 0;
            lap_n[1] += b == 1;
            lap_n[2] += b == 2;
            lap_n[3] += b == 3;
        }
        lap_syms = 0;
        lap_last = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(lap_last));
    }
#endif
    __device__ __forceinline__ uint32_t zeros_count() const { return nz; }
    __device__ __forceinline__ uint32_t zero_at(uint32_t t) const { return lane_read(zpos, t); }
};

// one scan's lookup table: pool (global) -> LDS slot, by one wave
__device__ __forceinline__ void stage_table(HJ_LDS uint16_t* slot, const uint16_t* pool, uint32_t offset, uint32_t words, uint32_t lane)
{
    const HJ_GLOBAL uint32_t* src = (const HJ_GLOBAL uint32_t*)(pool + offset);  // tables start at multiples of 64 entries
    HJ_LDS uint32_t* dst = (HJ_LDS uint32_t*)slot;
    for (uint32_t i = lane; i < words / 2; i += 64) dst[i] = src[i];
}

// entries of the table at pool + offset: first level + the second-level tables it refers to
__device__ __forceinline__ uint32_t table_words(const uint16_t* pool, uint32_t offset, uint32_t lane)
{
    const HJ_GLOBAL uint16_t* t = (const HJ_GLOBAL uint16_t*)(pool + offset);
    uint32_t m = 0;
    for (uint32_t i = lane; i < 256; i += 64) {
        const uint32_t e = t[i];
        if (e & kProgLong) m = max(m, e & 0x7FFFu);
    }
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    return (uni(m) + 1) * 256u;
}

// ---- DC scans -------------------------------------------------------------------------------------------------------------
// scan-order block b of a DC scan -> (component, index into its DC plane)
__device__ __forceinline__ void dc_block_address(const ProgImage& im, const ProgScan& sc, uint32_t b, uint32_t bpm, uint32_t* comp, uint32_t* index,
                                                 uint32_t* slot)
{
    if (sc.ncomp == 1) {
        const uint32_t c = sc.comps[0], nbx = im.nbx[c];
        const uint32_t by = b / nbx, bx = b - by * nbx;
        *comp = c;
        *slot = 0;
        *index = by * im.blocks_w[c] + bx;
        return;
    }
    const uint32_t mcu = b / bpm;
    uint32_t k = b - mcu * bpm;
    const uint32_t my = mcu / im.mcus_x, mx = mcu - my * im.mcus_x;
    uint32_t c = sc.comps[0], i = 0;
    for (; i + 1 < sc.ncomp; i++) {
        c = sc.comps[i];
        const uint32_t n = im.comp_h[c] * im.comp_v[c];
        if (k < n) break;
        k -= n;
    }
    c = sc.comps[i];
    const uint32_t h = im.comp_h[c], v = im.comp_v[c];
    const uint32_t dy = k / h, dx = k - dy * h;
    *comp = c;
    *slot = i;
    *index = (my * v + dy) * im.blocks_w[c] + mx * h + dx;
}

__device__ bool walk_dc_chain(ProgImage& im, const HuffImage* himgs, HJ_LDS uint16_t* tables, uint32_t slot_words, uint32_t lane)
{
    DcWalker w;
    w.lane = lane;
    w.tables = tables;
    w.slot_words = slot_words;
    for (int d = 0; d < (int)im.dc_len; d++) {
        const ProgScan& sc = im.scan[im.dc_chain[d]];
        const HuffImage& hi = himgs[sc.huff_image];
        const uint32_t total_bits = hi.total_bits;
        uint32_t bpm = 0;
        for (uint32_t i = 0; i < sc.ncomp; i++) bpm += im.comp_h[sc.comps[i]] * im.comp_v[sc.comps[i]];
        if (sc.ah != 0) {
            // refinement: bit b of the scan belongs to block b -- 64 blocks per step
            if (total_bits < sc.nblocks) return false;
            const HJ_GLOBAL uint32_t* g = (const HJ_GLOBAL uint32_t*)sc.stream;
            for (uint32_t base = 0; base < sc.nblocks; base += 64) {
                const uint32_t b = base + lane;
                if (b < sc.nblocks) {
                    const uint32_t wd = __builtin_bswap32(g[b >> 5]);
                    if ((wd >> (31 - (b & 31))) & 1u) {
                        uint32_t c, index, slot;
                        dc_block_address(im, sc, b, bpm, &c, &index, &slot);
                        HJ_GLOBAL int16_t* dst = (HJ_GLOBAL int16_t*)im.dc_plane[c] + index;
                        *dst = (int16_t)(*dst | (1 << sc.al));
                    }
                }
            }
            __threadfence_block();
            continue;
        }
        // first scan: the tables of its components, then symbol by symbol
        for (uint32_t i = 0; i < sc.ncomp; i++) stage_table(tables + i * slot_words, im.pool, sc.table[i], table_words(im.pool, sc.table[i], lane), lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // The walk is ONE loop over the scan's blocks in scan order, no way out of its middle (a broken stream is noticed once per 64
        // blocks).  What a block needs -- which component it belongs to, which table that component uses -- comes from a pattern of the
        // MCU, one block per lane, read with a readlane; the predictors live in the lanes of a register too.  The 64 values of a group are
        // collected in a register and stored together: every lane keeps the MCU position and the MCU coordinates of "its" block of the
        // group up to date by additions (no division, no descriptor loads in the loop) and looks the block's geometry up in the pattern.
        const uint32_t ncomp = uni(sc.ncomp);
        const uint32_t blocks_per_mcu = ncomp == 1 ? 1u : uni(bpm);  // (uniform anyway; said so, the MCU position stays in a scalar register)
        // per MCU position k (lane k < blocks_per_mcu):
        uint32_t pat = 0;             // 16 x table slot | component (position in the scan) << 8   -- what the walk reads
        uint32_t geo = 0, gbw = 0;    // h | v << 4 | dx << 8 | dy << 12 of the block; blocks_w of its component
        uint32_t gplo = 0, gphi = 0;  // the component's DC plane
        uint32_t need = 1;            // bit j: slot j holds a table of its own (the first component that uses the table owns the slot)
        {
            uint32_t first = 0;
            for (uint32_t i = 0; i < ncomp; i++) {
                uint32_t slot = i;
                for (uint32_t j = 0; j < i; j++)
                    if (uni(sc.table[j]) == uni(sc.table[i])) {
                        slot = j;
                        break;
                    }
                if (slot == i) need |= 1u << i;
                const uint32_t c = uni(sc.comps[i]);
                const uint32_t h = ncomp == 1 ? 1u : uni(im.comp_h[c]), v = ncomp == 1 ? 1u : uni(im.comp_v[c]);
                const unsigned long long plane = (unsigned long long)(uintptr_t)im.dc_plane[c];
                if (lane >= first && lane < first + h * v) {
                    const uint32_t r = lane - first, dy = r / h, dx = r - dy * h;
                    pat = (slot * 16u) | (i << 8);
                    geo = h | (v << 4) | (dx << 8) | (dy << 12);
                    gbw = uni(im.blocks_w[c]);
                    gplo = (uint32_t)plane;
                    gphi = (uint32_t)(plane >> 32);
                }
                first += h * v;
            }
        }
        w.need = need;
        w.start(reinterpret_cast<const uint32_t*>(sc.stream), hi.stream_words);
        const uint32_t al = uni(sc.al), nblocks = uni(sc.nblocks);
        // scan-order block b = MCU b / blocks_per_mcu, position b % blocks_per_mcu; MCUs in raster order, `across` per row (a scan of one
        // component: its real blocks in raster order, one per "MCU")
        const uint32_t across = ncomp == 1 ? uni(im.nbx[uni(sc.comps[0])]) : uni(im.mcus_x);
        uint32_t lk = lane % blocks_per_mcu, lm = lane / blocks_per_mcu;  // this lane's block of the current group: position, MCU ...
        uint32_t lmy = lm / across, lmx = lm - lmy * across;               // ... and the MCU's coordinates
        const uint32_t step_k = 64u % blocks_per_mcu, step_m = 64u / blocks_per_mcu;
        uint32_t predv = 0;  // lane i: predictor of the scan's i-th component
        uint32_t k = 0;      // position in the MCU
        bool bad = false;
        for (uint32_t b0 = 0; b0 < nblocks && !bad; b0 += 64) {
            const uint32_t cnt = nblocks - b0 < 64u ? nblocks - b0 : 64u;
            uint32_t outv = 0;
            for (uint32_t jj = 0; jj < cnt; jj++) {
                const uint32_t p = lane_read(pat, k);
                k = k + 1u == blocks_per_mcu ? 0u : k + 1u;
                int diff;
                bad |= !w.symbol(p & 255u, &diff);
                const uint32_t i = p >> 8;
                const uint32_t pred = lane_read(predv, i) + (uint32_t)diff;
                predv = lane == i ? pred : predv;
                outv = lane == jj ? pred : outv;
            }
            {
                const uint32_t g = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lk * 4u), (int)geo);
                const uint32_t bw = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lk * 4u), (int)gbw);
                const uint32_t plo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lk * 4u), (int)gplo);
                const uint32_t phi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(lk * 4u), (int)gphi);
                const uint32_t h = g & 15u, v = (g >> 4) & 15u, dx = (g >> 8) & 15u, dy = g >> 12;
                const uint32_t index = (lmy * v + dy) * bw + lmx * h + dx;
                HJ_GLOBAL int16_t* plane = (HJ_GLOBAL int16_t*)(uintptr_t)(((unsigned long long)phi << 32) | plo);
                if (lane < cnt) plane[index] = (int16_t)((int)outv * (1 << al));
                // on to the lane's block of the next group
                lk += step_k;
                const bool carry = lk >= blocks_per_mcu;
                lk -= carry ? blocks_per_mcu : 0u;
                lmx += step_m + (carry ? 1u : 0u);
                while (__builtin_amdgcn_ballot_w64(lmx >= across)) {  // (at most once for pictures wider than 64 MCUs)
                    const bool wrap = lmx >= across;
                    lmx -= wrap ? across : 0u;
                    lmy += wrap ? 1u : 0u;
                }
            }
            // (the position is checked once per group: a forged frame size cannot keep the wave walking through imaginary data)
            if (w.pos() > total_bits) bad = true;
        }
        if (bad) return false;
        if (w.pos() > total_bits) return false;
        __threadfence_block();
    }
    return true;
}

// One workgroup per image.  Wave 0 walks the DC scans; waves 1.. take the AC scans, component after component, the scans of a
// component in the order of the file (= pipeline stages).  Dynamic LDS: `dc_slots` table slots for the DC scans' tables, one
// slot per AC wave, then the rings.
__global__ __launch_bounds__(kWalkThreads) void prog_walk_kernel(ProgImage* __restrict__ images, const HuffImage* __restrict__ himgs, uint32_t slot_words,
                                                                 uint32_t dc_slots, uint32_t table_slots)
{
    __shared__ WalkShared sh;
    extern __shared__ unsigned long long dyn_lds[];  // table slots (uint16 entries), then the rings (8-byte aligned: slot_words % 64 == 0)
    HJ_LDS uint16_t* tables = (HJ_LDS uint16_t*)dyn_lds;
    HJ_LDS WalkRing* rings = (HJ_LDS WalkRing*)(tables + (size_t)table_slots * slot_words);
    ProgImage& im = images[blockIdx.x];
    // the wave number is the same in every lane, but only a readfirstlane tells the compiler so: everything the walk branches
    // on derives from it (which scan, its band, its table), and must live in SGPRs
    const int wave = (int)uni(threadIdx.x >> 6);
    const uint32_t lane = lane_id();
    if (threadIdx.x < kWalkMaxWaves) {
        sh.done[threadIdx.x] = 0;
        sh.taken[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) sh.abort_flag = 0;
    __syncthreads();
    bool ok = true;
    if (wave == 0) {
        const unsigned long long t0 = wall_clock64();
        ok = walk_dc_chain(im, himgs, tables, slot_words, lane);
#if defined(HJ_WALK_PROFILE)
        {  // what a lap books although nothing happens in it (the second stamp's own flight time)
            unsigned long long last = __builtin_readcyclecounter();
            uint32_t sum = 0;
            for (int i = 0; i < 256; i++) {
                const unsigned long long now = __builtin_readcyclecounter();
                sum += (uint32_t)(now - last);
                last = __builtin_readcyclecounter();
            }
            if (lane == 0 && im.dc_len) im.scan[im.dc_chain[0]].pad_ticks[1] = sum >> 8 >> 4;  // shown as "fast count"
        }
#endif
        if (lane == 0 && im.dc_len) {  // the whole DC chain, booked on its first scan
            im.scan[im.dc_chain[0]].walk_ticks = (uint32_t)(wall_clock64() - t0);
            im.scan[im.dc_chain[0]].pad_ticks[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    } else {
        // which component's chain, which stage of it; rings of the chains in front (a chain of n scans has n - 1 hand-overs)
        // Waves are dealt to the scans LAST stage first, component after component: the last refinement scans are the long ones
        // (DESIGN.md 3.5), the hardware deals a workgroup's waves to the four SIMDs in turn, and three long walks on one SIMD share
        // its issue slot while another SIMD idles.  `self` (the index into done[] / taken[]) stays the position in chain order.
        int a = -1, c = 0, ring_base = 0, self = 1;
        {
            int left = wave - 1;
            for (int level = 0; level < kProgMaxStages && a < 0; level++)
                for (int cc = 0; cc < 4; cc++) {
                    const int n = (int)uni(im.chain_len[cc]);
                    if (n <= level) continue;
                    if (left-- == 0) {
                        c = cc;
                        a = n - 1 - level;
                        break;
                    }
                }
            if (a < 0 || c >= (int)im.ncomp) return;
            for (int cc = 0; cc < c; cc++) {
                const int n = (int)uni(im.chain_len[cc]);
                ring_base += n > 0 ? n - 1 : 0;
                self += n;
            }
            self += a;
        }
        const int chain_len = (int)uni(im.chain_len[c]);
        const ProgScan& sc = im.scan[im.chain[c][a]];
        const HuffImage& hi = himgs[sc.huff_image];
        HJ_LDS uint16_t* slot = tables + (size_t)(dc_slots + (uint32_t)(wave - 1)) * slot_words;
        stage_table(slot, im.pool, sc.table[0], table_words(im.pool, sc.table[0], lane), lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        DevWalker w;
        w.lane = lane;
        w.sh = &sh;
        w.has_prev = a > 0;
        w.has_next = a + 1 < chain_len;
        w.ring_in = rings + (ring_base + a - 1);  // not dereferenced without a predecessor; ring_in + 1 = the ring to the successor
        w.self = self;
        w.block_pos = sc.block_pos;
        w.nblocks = sc.nblocks;
        w.aborted = false;
        w.waited = 0;
        w.nz = w.nz_next = 0;
        w.zpos = w.zraw_next = 0;
        w.table = slot;
        w.refine = uni(sc.ah) != 0;
        w.start(reinterpret_cast<const uint32_t*>(sc.stream), hi.stream_words);
#if defined(HJ_WALK_PROFILE)
        for (int i = 0; i < 4; i++) w.lap_sum[i] = w.lap_n[i] = 0;
        w.lap_syms = 0;
        w.lap_last = __builtin_readcyclecounter();
#endif
        const unsigned long long t0 = wall_clock64();
        ok = prog_walk_ac(w, (int)sc.ss, (int)sc.se, (int)sc.ah, w.has_next, sc.nblocks, hi.total_bits);
        if (lane == 0) {
            im.scan[im.chain[c][a]].walk_ticks = (uint32_t)(wall_clock64() - t0);
            im.scan[im.chain[c][a]].wait_ticks = w.waited;
            im.scan[im.chain[c][a]].pad_ticks[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID, all 32 bits
#if defined(HJ_WALK_PROFILE)
            // cycles (units of 4096) in the fast loops / event handling / window switches / block starts, and how many of each
            ProgScan& out = im.scan[im.chain[c][a]];
            out.wait_ticks = (w.lap_sum[0] >> 12) | ((w.lap_sum[1] >> 12) << 16);
            out.pad_ticks[0] = (w.lap_sum[2] >> 12) | ((w.lap_sum[3] >> 12) << 16);
            out.pad_ticks[1] = (w.lap_n[0] >> 4) | ((w.lap_n[1] >> 4) << 16);
            out.pad_ticks[2] = (w.lap_n[2] >> 4) | ((w.lap_n[3] >> 4) << 16);
#endif
        }
        if (w.aborted) ok = false;
        if (!ok && lane == 0) __hip_atomic_store(&sh.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (!ok && lane == 0) im.status = 1;  // benign race: every writer stores the same value
}

// ---- replay ---------------------------------------------------------------------------------------------------------------
constexpr int kRThreads = 256;
constexpr int kRBlockBytes = 144;  // 128 + 16: the lanes' buffers start in different banks

struct ReplayShared {
    __attribute__((aligned(16))) uint8_t blocks[kRThreads * kRBlockBytes];
    int16_t* dst[kRThreads];
    uint32_t zz[16];
    uint32_t wave_max[4];
    uint32_t err;
};

struct ReplayEnv {
    const HuffImage* himgs;
    const HJ_LDS uint16_t* tables;
    uint32_t slot_words;
    uint32_t buf;  // LDS byte address of this lane's block buffer
    const HJ_LDS uint8_t* zz;
    __device__ __forceinline__ uint32_t word(const ProgScan& sc, uint32_t i) const
    {
        const uint32_t n = himgs[sc.huff_image].stream_words;
        return i < n ? __builtin_bswap32(((const HJ_GLOBAL uint32_t*)sc.stream)[i]) : ~0u;
    }
    __device__ __forceinline__ uint32_t lookup(const ProgScan& sc, uint32_t w) const
    {
        const HJ_LDS uint16_t* t = tables + (uint32_t)sc.stage * slot_words;
        uint32_t e = t[w >> 24];
        if (e & kProgLong) e = t[(e & 0x7FFFu) * 256u + ((w >> 16) & 255u)];
        return e;
    }
    __device__ __forceinline__ int get(int k) const { return *(const HJ_LDS int16_t*)(uintptr_t)(buf + (uint32_t)zz[k] * 2u); }
    __device__ __forceinline__ void put(int k, int v) const { *(HJ_LDS int16_t*)(uintptr_t)(buf + (uint32_t)zz[k] * 2u) = (int16_t)v; }
};

// unit = {image, (component << 28) | first block of the allocation grid}
static_assert(sizeof(ReplayShared) <= 40 * 1024, "progressive_gpu_host.h kProgReplayStaticLds");
__global__ __launch_bounds__(kRThreads) void prog_replay_kernel(ProgImage* __restrict__ images, const HuffImage* __restrict__ himgs,
                                                                const HuffUnit* __restrict__ units, uint32_t slot_words)
{
    __shared__ ReplayShared sh;
    extern __shared__ uint16_t dyn_tables[];
    HJ_LDS uint16_t* tables = (HJ_LDS uint16_t*)dyn_tables;
    const HuffUnit u = units[blockIdx.x];
    ProgImage& im = images[u.image];
    const int c = (int)(u.first >> 28);
    const uint32_t first = u.first & 0x0FFFFFFFu;
    const int t = threadIdx.x;
    // the component's tables (all waves cooperate), the zigzag permutation, a zeroed buffer per lane
    for (int st = 0; st < (int)im.chain_len[c]; st++) {
        const ProgScan& sc = im.scan[im.chain[c][st]];
        const HJ_GLOBAL uint16_t* src16 = (const HJ_GLOBAL uint16_t*)(im.pool + sc.table[0]);
        uint32_t m = 0;
        const uint32_t e = src16[t];
        if (e & kProgLong) m = e & 0x7FFFu;
        for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
        if ((t & 63) == 0) sh.wave_max[t >> 6] = m;
        __syncthreads();
        const uint32_t words = (max(max(sh.wave_max[0], sh.wave_max[1]), max(sh.wave_max[2], sh.wave_max[3])) + 1) * 256u;
        const HJ_GLOBAL uint32_t* src = (const HJ_GLOBAL uint32_t*)src16;
        HJ_LDS uint32_t* dst = (HJ_LDS uint32_t*)(tables + (size_t)st * slot_words);
        for (uint32_t i = t; i < words / 2; i += kRThreads) dst[i] = src[i];
        __syncthreads();
    }
    if (t < 16) {
        constexpr uint8_t zz[64] = HJ_ZIGZAG_DEVICE_TABLE;
        sh.zz[t] = (uint32_t)zz[4 * t] | ((uint32_t)zz[4 * t + 1] << 8) | ((uint32_t)zz[4 * t + 2] << 16) | ((uint32_t)zz[4 * t + 3] << 24);
    }
    if (t == 0) sh.err = 0;
    HJ_LDS u32x4* my_buf = (HJ_LDS u32x4*)&sh.blocks[t * kRBlockBytes];
#pragma unroll
    for (int i = 0; i < kRBlockBytes / 16; i++) my_buf[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    const uint32_t bw = im.blocks_w[c], total = bw * im.blocks_h[c];
    const uint32_t a = first + (uint32_t)t;  // block of the allocation grid
    int16_t* dst = nullptr;
    if (a < total) {
        dst = im.coef[c] + (size_t)a * 64;
        const uint32_t by = a / bw, bx = a - by * bw;
        if (bx < im.nbx[c] && by < im.nby[c]) {  // blocks of the MCU padding carry no AC data
            ReplayEnv env;
            env.himgs = himgs;
            env.tables = tables;
            env.slot_words = slot_words;
            env.buf = (uint32_t)(uintptr_t)(HJ_LDS uint8_t*)&sh.blocks[t * kRBlockBytes];
            env.zz = (const HJ_LDS uint8_t*)sh.zz;
            if (!prog_replay_block(env, im, c, by * im.nbx[c] + bx)) sh.err = 1;
        }
    }
    sh.dst[t] = dst;
    __syncthreads();
    // eight lanes per block, 16 bytes each: every store instruction writes eight whole lines
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int blk = (t & ~63) + 8 * i + ((t & 63) >> 3), chunk = t & 7;
        int16_t* p = sh.dst[blk];
        if (p) {
            const u32x4 v = *(const HJ_LDS u32x4*)&sh.blocks[blk * kRBlockBytes + chunk * 16];
            __builtin_nontemporal_store(v, (HJ_GLOBAL u32x4*)p + chunk);
        }
    }
    if (t == 0 && sh.err) im.status = 1;
}

}  // namespace

int launch_prog_walk(ProgImage* images, const HuffImage* himgs, int nimages, unsigned slot_words, unsigned dc_slots, unsigned ac_waves, unsigned rings,
                     void* stream)
{
    if (nimages <= 0) return 0;
    if (ac_waves + 1 > (unsigned)kWalkMaxWaves) return (int)hipErrorInvalidValue;  // gpu_progressive_eligible() keeps such files away
    // LDS and wave slots decide how many walks are resident at once (a walk is one wave per scan: the chip is full of them long
    // before it is busy), so a workgroup gets exactly the waves, table slots and rings the batch's images need
    const unsigned table_slots = dc_slots + ac_waves;
    const unsigned lds = table_slots * slot_words * 2u + rings * (unsigned)sizeof(WalkRing);
    hipLaunchKernelGGL(prog_walk_kernel, dim3(nimages), dim3(64u * (ac_waves + 1)), lds, (hipStream_t)stream, images, himgs, slot_words, dc_slots, table_slots);
    return (int)hipGetLastError();
}

int launch_prog_replay(ProgImage* images, const HuffImage* himgs, const HuffUnit* units, int nunits, unsigned slot_words, void* stream)
{
    if (nunits <= 0) return 0;
    hipLaunchKernelGGL(prog_replay_kernel, dim3(nunits), dim3(kRThreads), slot_words * 2u * kProgMaxStages, (hipStream_t)stream, images, himgs, units,
                       slot_words);
    return (int)hipGetLastError();
}

}  // namespace hipjpeg
