// huffman_gpu_core.h -- data structures and the per-subsequence decode routine of the GPU entropy decoder.
// Compiles for both host and device: the HIP kernels (gpu_huffman.hip) and the host emulation used by the CPU tests
// (gpu_huffman_host.cpp) run the very same decode routine, parameterised only by how stream words and table entries are
// fetched (LDS on the device, plain memory on the host).
//
// Algorithm (self-synchronizing parallel Huffman decoding; Weissenberger & Schmidt, "Accelerating JPEG Decompression on
// GPUs", 2021 -- restated, not copied): the destuffed entropy-coded segment of a baseline scan is cut into subsequences of
// kSubseqBits bits.  Every subsequence is decoded by one lane.
//   pass 0     every lane decodes from the first bit of its subsequence assuming "start of a block, first block of an MCU";
//              it records where it stops (first symbol starting at or after the subsequence end) and in which decoder state.
//   pass t>0   lane i restarts from the end state lane i-1 recorded in pass t-1.  Huffman codes self-synchronize: after a
//              few symbols a decoder started in the wrong state falls into step with the true one, so end states stop
//              changing after a few passes.  When a whole pass changes nothing, every lane has decoded exactly what a
//              sequential decoder would have decoded in its subsequence (induction from subsequence 0, whose start state
//              is exact).
//   count      each pass also counts the blocks completed in the subsequence; an exclusive scan gives every lane the
//              index of its first block.
//   write      step 1: the lanes walk their subsequences once more and record where every block starts; step 2: one lane
//              per BLOCK decodes it from its start position into a buffer and the blocks are stored whole (column-major
//              block layout, DC position left zero); DC differences go to a compact per-image array in scan order.
//   dc         a per-component scan in MCU order turns DC differences into DC values, stored as one compact plane per
//              component (raster block order) that the IDCT kernels read next to the coefficient blocks.
// All decisions are integer/bit exact; the result is compared with the host entropy decoder and the oracle in tests/.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HJ_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define HJ_HD inline
#endif

namespace hipjpeg {

#ifndef HJ_SUBSEQ_BITS
#define HJ_SUBSEQ_BITS 1024
#endif
#ifndef HJ_SYNC_THREADS
#define HJ_SYNC_THREADS 256
#endif
constexpr int kSubseqBits = HJ_SUBSEQ_BITS;  // bits per subsequence (one lane's share of the stream; its LDS footprint)
constexpr int kSubseqWords = kSubseqBits / 32;
constexpr int kHuffFastBits = 10;      // first-level lookup width
constexpr int kHuffSubBits = 16 - kHuffFastBits;
constexpr int kStreamSlackBytes = 32;  // readable bytes after the last real byte of a destuffed stream
constexpr int kSyncThreads = HJ_SYNC_THREADS;  // lanes of a workgroup of the sync / position kernels
constexpr int kHuffOwn = kSyncThreads - 1;     // subsequences such a workgroup owns (one lane is the halo)
constexpr int kTailTaskBytes = 2 * kSyncThreads;  // per workgroup: the subsequences (uint16) it hands to the tail kernel
constexpr int kHuffBlocksPerWg = 256;   // lanes per workgroup of the block pass (one block per lane per round)
#ifndef HJ_MCUS_PER_WG
#define HJ_MCUS_PER_WG 128  // entropy stage per 256 x 1080p: 64 -> 2.12 ms, 128 -> 1.99, 256 -> 1.99, 512 -> 2.02 (tools/ab_full_flag.sh)
#endif
constexpr int kHuffMcusPerWg = HJ_MCUS_PER_WG;     // MCUs a workgroup of the block pass covers
constexpr int kDestuffChunk = 16384;   // raw bytes one workgroup of the destuff kernels handles
constexpr int kMaxPoolWords = 12288;   // uint16 entries of lookup tables per image the kernels accept (24 KB of LDS)

// Lookup-table entry (uint16), laid out for the state update the decoders do per symbol:
//   bits 0-4   total = code length + number of value bits (1..31)
//   bits 5-8   nb = number of value bits (size category)
//   bits 9-15  zadv = how far the zigzag position advances: 1 for a DC symbol, run + 1 for a coefficient, 16 for ZRL,
//              64 for EOB (any value that takes the position past 63 ends the block)
// "Code longer than kHuffFastBits" is zadv = 127 with bits 0-8 = pool offset / 64 of the second-level table, indexed by
// the next kHuffSubBits bits: it trips the same "position past 63" test as the end of a block, so the common path has one
// test for both.  "No such code" is total = 1, nb = 15 (impossible for a real symbol), zadv = 1.
HJ_HD constexpr uint32_t make_entry(uint32_t total, uint32_t nb, uint32_t zadv) { return total | (nb << 5) | (zadv << 9); }
// PAIR table of an AC table (directly behind its first level, same index): the walks that only track positions -- the
// synchronisation passes and the position pass, 2/3 of the stage's time -- take two AC symbols per step where the window holds
// both completely (half of all steps on a q90 photo).  Entry: bits 0-3 = bits of both symbols together (<= kHuffFastBits),
// bits 4-9 = zigzag advance of both (63 when the second one is EOB: enough to end any block from an AC position), bits 10-15 =
// the zigzag positions from which the pair may be taken: z < that many, i.e. 64 - the first symbol's own advance, so that the
// first symbol keeps the block open; 0 = no pair here (first symbol EOB / long code / no second symbol in the window).  The
// walker also wants the second symbol to start inside its range -- then the result is exactly that of two single steps.
constexpr uint32_t kPairOffset = 1u << kHuffFastBits;  // entries between an AC table's first level and its pair table
HJ_HD constexpr uint32_t make_pair_entry(uint32_t total, uint32_t zadv, uint32_t z_below) { return total | (zadv << 4) | (z_below << 10); }
constexpr uint32_t kEntryInvalid = 1u | (15u << 5) | (1u << 9);
constexpr uint32_t kZadvLong = 127;

// One block position k inside the MCU (k < blocks_per_mcu <= 10).
struct HuffK {
    uint32_t blk0;      // (dy * blocks_w + dx): block offset of this position inside MCU (0,0) of its component
    uint32_t stride_y;  // blocks between vertically adjacent MCUs: v * blocks_w
    uint16_t tdc, tac;  // pool offsets (uint16 units) of the first-level DC / AC tables
    uint8_t comp;       // component index
    uint8_t stride_x;   // blocks between horizontally adjacent MCUs: h
    uint8_t pad[2];
};

// Per-image description for the entropy kernels.
struct alignas(16) HuffImage {
    const uint8_t* stream;   // destuffed entropy-coded bytes, 4-byte aligned, `stream_words` 32-bit words readable
                             // (written by the destuff kernels together with total_bits, num_subseq, stream_words)
    const uint16_t* pool;    // lookup tables (pool_words entries)
    int16_t* coef[4];        // component coefficient blocks (device layout)
    int16_t* dc_diff;        // total_blocks DC differences in scan order (written by the write pass)
    int16_t* dc_plane[4];    // per component: DC values, one per block, raster order over the allocation grid (dc kernel)
    uint32_t total_bits;     // 8 * destuffed length
    uint32_t first_subseq;   // index of this image's first subsequence in the batch-wide arrays
    uint32_t num_subseq;     // ceil(total_bits / kSubseqBits)
    uint32_t total_blocks;   // mcus * blocks_per_mcu
    uint32_t mcus_x, mcus_y, blocks_per_mcu, ncomp;
    uint32_t pool_words, stream_words;
    uint32_t status;         // written by the kernels: 0 ok, 1 = invalid code inside the real data, 2 = block count mismatch
    uint32_t first_chunk;    // index of this image's first kDestuffChunk-byte chunk in the batch-wide drop-count array
    const uint8_t* raw;      // the scan's entropy-coded bytes as they are in the file (byte-stuffed), padded to 16 bytes
    uint32_t raw_bytes;
    uint32_t decoded_blocks; // blocks the converged decoders completed (scan kernel); < total_blocks = truncated stream
    uint32_t* block_pos;     // total_blocks bit positions: where every block starts (position pass -> block pass)
    // restart intervals (0 = none): the bit positions where intervals 1, 2, ... begin in the destuffed stream, and for every
    // subsequence the index of the first of them at or behind its first bit
    const uint32_t* boundaries;
    const uint32_t* sub_boundary;
    uint32_t restart_interval, num_boundaries;
    // Written by the tail / ripple kernels.  A periodic stream (stripes, a test pattern) can keep a decoder that started in the
    // wrong state on a stable wrong trajectory: corrections then travel through the image one subsequence at a time, strictly in
    // sequence -- the self-synchronising scheme has nothing to offer and the host decoder is far faster.  gave_up = 1: a chain of
    // the image exceeded its round budget; moved_pass = number of the last ripple launch in which a group's own last end state
    // still changed.  The host hands such images to the host entropy decoder instead of launching on (decoder_core.cpp resolve).
    uint32_t gave_up, moved_pass;
    // zero-copy input: the scan's bytes in the CALLER's page-locked host memory, as the device sees it (null: staged like everything else).
    // gather_raw_kernel copies them to `raw` in place of a DMA from the staging area.
    const uint8_t* raw_src;
    uint64_t pad2;
    HuffK k[10];
    uint32_t blocks_w[4];
    uint8_t comp_h[4], comp_v[4], comp_k0[4], pad1[4];  // comp_k0 = first position k of the component inside the MCU
};

// What a lane knows after decoding a subsequence.
struct SubseqState {
    uint32_t end_bit;   // position of the first symbol that starts at or after the subsequence end
    uint16_t zk;        // (k << 8) | z : position inside the MCU, zigzag index (0 = DC expected)
    uint16_t nblocks;   // blocks completed by symbols that started inside the subsequence
};

HJ_HD uint64_t pack_state(const SubseqState& s) { return ((uint64_t)s.end_bit) | ((uint64_t)s.zk << 32) | ((uint64_t)s.nblocks << 48); }
HJ_HD SubseqState unpack_state(uint64_t v)
{
    SubseqState s;
    s.end_bit = (uint32_t)v;
    s.zk = (uint16_t)(v >> 32);
    s.nblocks = (uint16_t)(v >> 48);
    return s;
}
constexpr uint64_t kSyncMask = 0x0000FFFFFFFFFFFFull;  // the part of a packed state the successor depends on

// zigzag index -> position inside a device-layout block (transposed natural order); same table as entropy_decode.cpp.
#define HJ_ZIGZAG_DEVICE_TABLE                                                                                                            \
    {0,  8,  1,  2,  9,  16, 24, 17, 10, 3,  4,  11, 18, 25, 32, 40, 33, 26, 19, 12, 5,  6,  13, 20, 27, 34, 41, 48, 56, 49, 42, 35, \
     28, 21, 14, 7,  15, 22, 29, 36, 43, 50, 57, 58, 51, 44, 37, 30, 23, 31, 38, 45, 52, 59, 60, 53, 46, 39, 47, 54, 61, 62, 55, 63}
static const uint8_t kZigzagDeviceGpuHost[64] = HJ_ZIGZAG_DEVICE_TABLE;

// The scalars the decode loop needs, copied out of HuffImage so that they live in registers across the coefficient stores.
struct HuffGeom {
    uint32_t total_bits, blocks_per_mcu, mcus_x, mcus_y;
    int16_t* dc_diff;
    const uint32_t* boundaries;
    uint32_t num_boundaries;
    uint32_t interval_blocks;  // blocks per restart interval (0: the scan has none)
};
HJ_HD HuffGeom make_geom(const HuffImage& im)
{
    HuffGeom g;
    g.total_bits = im.total_bits;
    g.blocks_per_mcu = im.blocks_per_mcu;
    g.mcus_x = im.mcus_x;
    g.mcus_y = im.mcus_y;
    g.dc_diff = im.dc_diff;
    g.boundaries = im.boundaries;
    g.num_boundaries = im.num_boundaries;
    g.interval_blocks = im.restart_interval * im.blocks_per_mcu;
    return g;
}

// MSB-first bit reader over 32-bit words fetched through Env::word(index).  The reader holds the 64 bits w0:w1 around the
// current position and a shift s: `hi` -- the next 32 bits of the stream (a symbol is at most 16 code + 15 value bits) -- is
// bits [s+31 : s] of w0:w1, one v_alignbit_b32.  Consuming c bits is s -= c; when s goes negative the pair moves on by one
// word.  The word that may be needed next is requested at the top of every step, before the table lookup, so that its latency
// overlaps the lookup instead of preceding it.
HJ_HD uint32_t funnel_shift_right(uint32_t high, uint32_t low, uint32_t s)  // bits [s+31 : s] of high:low, 0 <= s <= 31
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(high, low, s);
#else
    return (uint32_t)((((uint64_t)high << 32) | low) >> (s & 31));
#endif
}
// Words are addressed through a CURSOR the environment defines (Env::cursor(word index), Env::fetch(cursor), consecutive words
// Env::kCursorStep apart): a plain index on the host, an LDS byte address in the kernels whose lanes keep their stretch of the
// stream in LDS -- moving on by a word is then one add, not an index-to-address computation per symbol.
HJ_HD uint32_t extract_bits(uint32_t w, uint32_t offset, uint32_t width)  // bits [offset + width - 1 : offset] of w; offset <= 31, width <= 16
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(w, offset, width);
#else
    return (w >> offset) & ((1u << width) - 1u);
#endif
}
struct BitReader {
    uint32_t hi;      // the next 32 bits of the stream
    uint32_t w0, w1;  // the words hi is cut from
    uint32_t s;       // 0 <= s <= 31
    uint32_t next;    // cursor of the word behind w1
    uint32_t step;    // Env::kCursorStep
    template <class Env>
    HJ_HD void start(const Env& env, uint32_t pos)
    {
        const uint32_t i = pos >> 5, b = pos & 31;
        const uint32_t c0 = env.cursor(i);
        const uint32_t c1 = c0 + (b != 0 ? Env::kCursorStep : 0u);  // on a word boundary the window is w1 alone (s = 0)
        w0 = env.fetch(c0);
        w1 = env.fetch(c1);
        s = (0u - b) & 31u;
        step = Env::kCursorStep;
        next = c1 + Env::kCursorStep;
        hi = funnel_shift_right(w0, w1, s);
    }
    // drops c (1..31) bits; `fetched` = the word at `next`, taken when the position leaves w0
    HJ_HD void consume(uint32_t c, uint32_t fetched)
    {
        const uint32_t t = s - c;
        const bool need = (int32_t)t < 0;
        w0 = need ? w1 : w0;
        w1 = need ? fetched : w1;
        next += need ? step : 0u;
        s = t & 31u;
        hi = funnel_shift_right(w0, w1, s);
    }
};

// Restart markers (DRI): the scan is a chain of independent intervals.  The markers themselves are removed together with the
// byte stuffing; what is left of them is a list of BOUNDARIES, the bit positions (multiples of 8) where intervals begin.  In
// front of a boundary the encoder has filled the last byte with one-bits.  A decoder looks at them before every symbol:
// (at the start of a walk and whenever it has completed an MCU -- the only places where the true trajectory can meet a boundary)
//   * it is at or past the next boundary: whatever it believed, a fresh interval starts exactly there -- position, zigzag index
//     and MCU position are forced (for the true trajectory this changes nothing; a desynchronised one is exact from here on,
//     so no correction chain outlives an interval);
//   * it stands at the end of an MCU less than a byte in front of the boundary and only one-bits remain: padding, skip it
//     (no complete symbol consists of one-bits only, so real data is never skipped).
// Env::boundary(i) = bit position of boundary i, 0xFFFFFFFF behind the last one.
struct RestartCursor {
    uint32_t index, bound;
    template <class Env>
    HJ_HD void start(const Env& env, uint32_t first_candidate, uint32_t pos)
    {
        index = first_candidate;
        bound = env.boundary(index);
        while (bound < pos) bound = env.boundary(++index);
    }
    // returns true when pos/z/k were changed (the caller restarts its bit reader and reloads its table selection).
    // *overrun (optional) is set when the decoder had already read past the boundary: on the true trajectory of an intact
    // stream an interval ends exactly at its boundary, so this only happens to damaged data.
    template <class Env>
    HJ_HD bool normalise(const Env& env, uint32_t hi_bits, uint32_t* pos, int* z, int* k, uint32_t* overrun = nullptr)
    {
        if (*pos >= bound) {
            const bool changed = *pos != bound || *z != 0 || *k != 0;
            if (overrun && *pos != bound) *overrun = 1;
            *pos = bound;
            *z = 0;
            *k = 0;
            bound = env.boundary(++index);
            return changed;
        }
        const uint32_t gap = bound - *pos;
        if (*z == 0 && *k == 0 && gap < 8 && (hi_bits >> (32 - gap)) == (1u << gap) - 1) {
            *pos = bound;
            bound = env.boundary(++index);
            return true;
        }
        return false;
    }
};

// Synchronisation decode: the symbols that START in [begin, limit) (and before total_bits), beginning in state (z, k).
// Tracks the decoder state and counts the blocks completed; nothing is stored.
// Env supplies the memory accessors:
//   uint32_t cursor(uint32_t i), fetch(cursor), kCursorStep   32-bit word i of the stream, first byte in the most significant
//                                        position, through a cursor (see BitReader)
//   uint32_t tables(int k)               tdc | tac << 16 for block position k: where the first-level tables are, in whatever
//                                        unit lookup1 wants
//   uint32_t lookup1(uint32_t t, w)      first-level entry of table t for window w (index = top kHuffFastBits bits)
//   uint32_t lookup2(uint32_t e, w)      second-level entry behind first-level entry e (index = next kHuffSubBits bits)
//   uint32_t lookup_pair(uint32_t t, w)  pair-table entry of table t for window w (same index as lookup1, kPairOffset entries on)
// RST: the scan has restart intervals; boundary0 = index of the first boundary at or behind the subsequence's first bit.
template <bool RST, class Env>
HJ_HD SubseqState decode_subsequence(const HuffGeom& im, const Env& env, uint32_t begin, uint32_t limit, int z, int k, uint32_t boundary0 = 0)
{
    uint32_t pos = begin, nblocks = 0;
    const uint32_t end = limit < im.total_bits ? limit : im.total_bits;
    // a pair's first symbol has at most kHuffFastBits - 2 bits: from here on the second one might start outside the range
    const uint32_t pair_end = end > (uint32_t)kHuffFastBits - 2 ? end - ((uint32_t)kHuffFastBits - 2) : 0;  // pos + 8 < end
    const int bpm = (int)im.blocks_per_mcu;
    uint32_t tsel = env.tables(k);
    uint32_t tcur = z == 0 ? (tsel & 0xFFFFu) : (tsel >> 16);  // table of the next symbol: DC at the start of a block, AC after it
    BitReader br;
    br.start(env, pos);
    RestartCursor rc;
    if (RST) {
        // boundaries are looked at where the true trajectory meets them: at the start and whenever an MCU has been completed
        rc.start(env, boundary0, pos);
        if (rc.normalise(env, br.hi, &pos, &z, &k)) {
            tsel = env.tables(k);
            tcur = tsel & 0xFFFFu;
            br.start(env, pos);
        }
    }
    while (pos < end) {
        const uint32_t fetched = env.fetch(br.next);
        uint32_t e = env.lookup1(tcur, br.hi);
        const uint32_t pr = env.lookup_pair(tcur, br.hi);  // means something behind an AC table only (z != 0)
        const bool two = z != 0 && (uint32_t)z < (pr >> 10) && pos < pair_end;
        uint32_t tot = two ? (pr & 15u) : (e & 31u);
        tcur = tsel >> 16;
        z += two ? (int)((pr >> 4) & 63u) : (int)(e >> 9);
        bool mcu_done = false;
        if (z >= 64) {  // end of a block, or a code that continues in a second-level table
            if (!two && (e >> 9) == kZadvLong) {
                e = env.lookup2(e, br.hi);
                z += (int)(e >> 9) - (int)kZadvLong;
                tot = e & 31u;
            }
            if (z >= 64) {
                z = 0;
                nblocks++;
                if (++k == bpm) k = 0;
                tsel = env.tables(k);
                tcur = tsel & 0xFFFFu;
                mcu_done = k == 0;
            }
        }
        pos += tot;
        br.consume(tot, fetched);
        if (RST && mcu_done && rc.normalise(env, br.hi, &pos, &z, &k)) {
            tsel = env.tables(k);
            tcur = tsel & 0xFFFFu;
            br.start(env, pos);
        }
    }
    SubseqState st;
    st.end_bit = pos;
    st.zk = (uint16_t)((k << 8) | z);
    st.nblocks = (uint16_t)(nblocks > 0xFFFF ? 0xFFFF : nblocks);
    return st;
}

// One subsequence by a whole wave (the last links of a correction chain: gpu_huffman.hip coop_decode).  The WINDOW supplies, for
// the 64 bit offsets behind a position, the table entry of the symbol that WOULD start there -- as a DC symbol and as an AC symbol
// of one set of tables -- and this routine is the scalar walk over it: same state evolution as decode_subsequence<false>, symbol
// by symbol (a pair step there is two steps here).  Shared by the kernel and the host emulation (gpu_huffman_host.cpp), so that the
// CPU tests pin the window / limit / table-change logic.
//   void     win.open(pos, ts)     entries for the symbols starting at bits pos .. pos + 63, tables ts = tdc | tac << 16
//   uint32_t win.dc(rel), ac(rel)  the entry at offset rel < 64 (through the second level): total bits [4:0], zigzag advance [15:9]
//   uint32_t win.tables(k)         Env::tables
// changes: bit k set = MCU position k is followed by a position with other tables (cooperative_table_changes).
// HJ_UNIFORM marks values that are the same in every lane of the wave (a readfirstlane on the device: they live in SGPRs).
#if defined(__HIP_DEVICE_COMPILE__)
#define HJ_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#else
#define HJ_UNIFORM(x) ((uint32_t)(x))
#endif
template <class Win>
HJ_HD uint32_t cooperative_table_changes(const Win& win, uint32_t blocks_per_mcu)
{
    uint32_t changes = 0;
    for (uint32_t q = 0; q < blocks_per_mcu; q++)
        changes |= (HJ_UNIFORM(win.tables(q)) != HJ_UNIFORM(win.tables(q + 1 == blocks_per_mcu ? 0 : q + 1)) ? 1u : 0u) << q;
    return changes;
}
template <class Win>
HJ_HD SubseqState cooperative_subsequence(const HuffGeom& im, Win& win, uint32_t changes, uint32_t begin, uint32_t limit, uint32_t z, uint32_t k)
{
    const uint32_t end = HJ_UNIFORM(limit < im.total_bits ? limit : im.total_bits);
    const uint32_t bpm = HJ_UNIFORM(im.blocks_per_mcu);
    uint32_t pos = HJ_UNIFORM(begin), nblocks = 0;
    z = HJ_UNIFORM(z);
    k = HJ_UNIFORM(k);
    uint32_t ts = HJ_UNIFORM(win.tables(k));  // read again only where the tables change
    while (pos < end) {
        win.open(pos, ts);
        // symbols that start at offsets below `room` belong to this window (64 offsets, and the subsequence's end)
        uint32_t room = end - pos < 64u ? end - pos : 64u, rel = 0;
        if (z == 0) {  // the window opens a block: its DC symbol
            const uint32_t e = win.dc(0);
            rel = e & 31u;
            z = e >> 9;
        }
        // coefficients; a block that ends inside the window is followed by the next one's DC symbol right here, so that the loop
        // itself never asks which table applies -- on the device seven scalar instructions and one taken branch per symbol
        while (rel < room) {
            const uint32_t e = win.ac(rel);
            rel += e & 31u;
            z += e >> 9;
            if (z >= 64) {
                z = 0;
                nblocks++;
                const bool other_tables = (changes >> k) & 1u;
                k = k + 1 == bpm ? 0 : k + 1;
                if (other_tables) {
                    room = 0;  // the next block uses other tables: the window's entries no longer apply
                    ts = HJ_UNIFORM(win.tables(k));
                } else if (rel < room) {
                    const uint32_t d = win.dc(rel);
                    rel += d & 31u;
                    z = d >> 9;
                }
            }
        }
        pos += rel;
    }
    SubseqState st;
    st.end_bit = pos;
    st.zk = (uint16_t)((k << 8) | z);
    st.nblocks = (uint16_t)(nblocks > 0xFFFF ? 0xFFFF : nblocks);
    return st;
}

// Write pass, step 1 -- where the blocks start.  Same walk as decode_subsequence from the converged start state; calls
// rec(block, pos) for every block whose first (DC) symbol starts in [begin, limit): `block` = scan-order index, `pos` = bit
// position (behind the padding of a restart boundary, if one lies in front of it).  `block0` is the index of the block in
// progress at `begin` (the scan's prefix sum of completed blocks).
template <bool RST, class Env, class Rec>
HJ_HD void position_subsequence(const HuffGeom& im, const Env& env, uint32_t begin, uint32_t limit, int z, int k, uint32_t block0, const Rec& rec,
                                uint32_t boundary0 = 0, uint32_t* fault = nullptr)
{
    // Restart intervals are also where damage shows (the host decoder demands the marker exactly where an interval's last
    // MCU ends): *fault is set when a boundary is met with a block count other than (boundary number + 1) x blocks per interval,
    // or after the decoder has read past it.
    uint32_t pos = begin, block = block0;
    const uint32_t end = limit < im.total_bits ? limit : im.total_bits;
    const uint32_t pair_end = end > (uint32_t)kHuffFastBits - 2 ? end - ((uint32_t)kHuffFastBits - 2) : 0;  // see decode_subsequence
    const int bpm = (int)im.blocks_per_mcu;
    uint32_t tsel = env.tables(k);
    uint32_t tcur = z == 0 ? (tsel & 0xFFFFu) : (tsel >> 16);
    bool fresh = z == 0;  // the next symbol opens a block
    BitReader br;
    br.start(env, pos);
    RestartCursor rc;
    if (RST) {
        rc.start(env, boundary0, pos);
        const uint32_t met = rc.index;
        if (rc.normalise(env, br.hi, &pos, &z, &k, fault)) {
            tsel = env.tables(k);
            tcur = tsel & 0xFFFFu;
            br.start(env, pos);
            fresh = true;
        }
        if (fault && rc.index != met && block != (met + 1) * im.interval_blocks) *fault = 1;
    }
    // a block is recorded where its predecessor ends (inside the rare end-of-block branch), the one in progress at `begin` here
    if (fresh && pos < end) rec(block, pos);
    while (pos < end) {
        const uint32_t fetched = env.fetch(br.next);
        uint32_t e = env.lookup1(tcur, br.hi);
        const uint32_t pr = env.lookup_pair(tcur, br.hi);
        const bool two = z != 0 && (uint32_t)z < (pr >> 10) && pos < pair_end;
        if (!two && (e >> 9) == kZadvLong) e = env.lookup2(e, br.hi);
        const uint32_t tot = two ? (pr & 15u) : (e & 31u);
        pos += tot;
        br.consume(tot, fetched);
        z += two ? (int)((pr >> 4) & 63u) : (int)(e >> 9);
        tcur = tsel >> 16;
        if (z >= 64) {
            z = 0;
            block++;
            if (++k == bpm) k = 0;
            tsel = env.tables(k);
            tcur = tsel & 0xFFFFu;
            if (RST && k == 0) {
                const uint32_t met = rc.index;
                if (rc.normalise(env, br.hi, &pos, &z, &k, fault)) {
                    tsel = env.tables(k);
                    tcur = tsel & 0xFFFFu;
                    br.start(env, pos);
                }
                if (fault && rc.index != met && block != (met + 1) * im.interval_blocks) *fault = 1;
            }
            if (pos < end) rec(block, pos);
        }
    }
}

// Write pass, step 2 -- one block, from its start position: every block is an independent piece of work once step 1 has
// found where it begins.  Coefficients go to env.put(device-layout index, value) (a zero-initialised 64-entry buffer);
// returns the DC difference.  *error is set for an invalid code, a run past position 63, or a stream that ends inside the
// block.  k = the block's position inside its MCU (selects the tables).
// Env additionally supplies:  int zigzag(int z)  and  void put(int index, int value).
template <class Env>
HJ_HD int decode_block(const HuffGeom& im, const Env& env, uint32_t pos, int k, uint32_t* error)
{
    const uint32_t tsel = env.tables(k);
    int z = 0, dc = 0;
    uint32_t err = 0;
    BitReader br;
    br.start(env, pos);
    // one symbol: its table entry (through the second level if the code is long), the value its nb extra bits stand for
    // (values below 2^(nb-1) are the negative half, JPEG "EXTEND": with m = 2^nb - 1, v is in the lower half exactly when
    // 2v <= m, and the value is then v - m), and "no such code"
    struct Symbol {
        uint32_t total, zadv, nb;
        int val;
        bool bad;
    };
    auto read = [&](uint32_t table, uint32_t w) {
        uint32_t e = env.lookup1(table, w);
        if ((e >> 9) == kZadvLong) e = env.lookup2(e, w);
        Symbol s;
        s.total = e & 31u;
        s.zadv = e >> 9;
        const uint32_t nb_raw = (e >> 5) & 15u;
        s.bad = nb_raw >= s.total;
        s.nb = s.bad ? 0u : nb_raw;
        const uint32_t v = extract_bits(w, 32u - s.total, s.nb);
        const uint32_t m = (1u << s.nb) - 1u;
        s.val = (v << 1) <= m ? (int)v - (int)m : (int)v;
        return s;
    };
    // the block's first symbol is its DC difference -- taken out of the loop, which then knows it codes coefficients
    if (pos < im.total_bits) {
        const uint32_t fetched = env.fetch(br.next);
        const Symbol s = read(tsel & 0xFFFFu, br.hi);
        err |= (uint32_t)s.bad;
        dc = s.val;
        pos += s.total;
        br.consume(s.total, fetched);
        z = (int)s.zadv;
    }
    const uint32_t tac = tsel >> 16;
    while (pos < im.total_bits && z < 64) {
        const uint32_t fetched = env.fetch(br.next);
        const Symbol s = read(tac, br.hi);
        const uint32_t zpos = (uint32_t)z + s.zadv - 1;
        const bool coefficient = s.nb != 0;  // ZRL and EOB carry no value
        err |= (uint32_t)(s.bad | (coefficient & (zpos > 63)));
        if (coefficient & (zpos <= 63)) env.put(env.zigzag((int)(zpos & 63)), s.val);
        pos += s.total;
        br.consume(s.total, fetched);
        z += (int)s.zadv;
    }
    // the stream ended inside the block, or its last symbol reaches into the slack behind the data (the host decoder calls
    // that TRUNCATED, BitReader::overran in entropy_decode.cpp)
    if (z < 64 || pos > im.total_bits) err = 1;
    if (err) *error = 1;
    return dc;
}

// Address of scan-order block `block`: env.block_ptr(k, mx, my) with k = block % blocks_per_mcu, (mx, my) = its MCU.
template <class Env>
HJ_HD int16_t* block_address(const HuffGeom& im, const Env& env, uint32_t block, int* k_out)
{
    const uint32_t mcu = block / im.blocks_per_mcu, k = block - mcu * im.blocks_per_mcu;
    const uint32_t my = mcu / im.mcus_x, mx = mcu - my * im.mcus_x;
    *k_out = (int)k;
    return env.block_ptr((int)k, mx, my);
}

}  // namespace hipjpeg
