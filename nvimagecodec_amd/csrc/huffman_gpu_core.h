// huffman_gpu_core.h -- data structures and the per-subsequence decode routine of the GPU entropy decoder.
// Compiles for both host and device: the HIP kernels (gpu_huffman.hip) and the host emulation used by the CPU tests
// (gpu_huffman_host.cpp) run the very same code.
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
//   write      the lanes decode once more and write coefficients (column-major block layout, DC still differential).
//   dc         a per-component scan in MCU order turns DC differences into DC values.
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

constexpr int kSubseqBits = 1024;  // bits per subsequence (128 bytes)
constexpr int kHuffFastBits = 10;
constexpr int kStreamSlackBytes = 32;  // readable bytes after the last real byte of a destuffed stream

// One Huffman table as the decoder consumes it.
struct HuffDecodeTable {
    uint16_t fast[1 << kHuffFastBits];  // (length << 8) | symbol, 0 = code longer than kHuffFastBits (or invalid)
    int32_t maxcode[18];                // largest code of each length (right-aligned), -1 = none; [17] = sentinel
    int32_t valoff[17];
    uint8_t vals[256];
};

// Per-image description for the entropy kernels.
struct alignas(16) HuffImage {
    const uint8_t* stream;         // destuffed entropy-coded bytes (+ kStreamSlackBytes of 0xFF padding)
    const HuffDecodeTable* tables; // 8 slots: [0..3] DC tables by id, [4..7] AC tables by id
    int16_t* coef[4];              // component coefficient blocks (device layout)
    uint32_t total_bits;           // 8 * destuffed length
    uint32_t first_subseq;         // index of this image's first subsequence in the batch-wide arrays
    uint32_t num_subseq;
    uint32_t total_blocks;         // mcus * blocks_per_mcu
    uint32_t mcus_x, blocks_per_mcu, ncomp, pad0;
    uint32_t blocks_w[4];          // allocation grid width per component
    // per position k inside an MCU (k < blocks_per_mcu <= 10)
    uint8_t k_comp[12], k_dx[12], k_dy[12], k_dc[12], k_ac[12];
    uint8_t comp_h[4], comp_v[4];
    uint32_t status;               // written by the kernels: 0 ok, 1 = invalid code inside the real data, 2 = block count mismatch
    uint32_t pad1[3];
};

// What a lane knows after decoding a subsequence.
struct SubseqState {
    uint32_t end_bit;   // position of the first symbol that starts at or after the subsequence end
    uint16_t zk;        // (k << 8) | z : position inside the MCU, zigzag index (0 = DC expected)
    uint16_t nblocks;   // blocks completed by symbols that started inside the subsequence
};

HJ_HD bool same_sync_state(const SubseqState& a, const SubseqState& b) { return a.end_bit == b.end_bit && a.zk == b.zk; }

// zigzag index -> position inside a device-layout block (transposed natural order); same table as entropy_decode.cpp.
// Two copies: a __constant__ one for device code and a plain one for host code (the host shadow of a __constant__ variable
// holds no data).
#define HJ_ZIGZAG_DEVICE_TABLE                                                                                                            \
    {0,  8,  1,  2,  9,  16, 24, 17, 10, 3,  4,  11, 18, 25, 32, 40, 33, 26, 19, 12, 5,  6,  13, 20, 27, 34, 41, 48, 56, 49, 42, 35, \
     28, 21, 14, 7,  15, 22, 29, 36, 43, 50, 57, 58, 51, 44, 37, 30, 23, 31, 38, 45, 52, 59, 60, 53, 46, 39, 47, 54, 61, 62, 55, 63}
#if defined(__HIPCC__)
__device__ __constant__ static const uint8_t kZigzagDeviceGpuConst[64] = HJ_ZIGZAG_DEVICE_TABLE;
#endif
static const uint8_t kZigzagDeviceGpuHost[64] = HJ_ZIGZAG_DEVICE_TABLE;
HJ_HD int zigzag_to_device(int z)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return kZigzagDeviceGpuConst[z];
#else
    return kZigzagDeviceGpuHost[z];
#endif
}

// Next 64 bits of the stream starting at bit `pos`, MSB first.  The stream has kStreamSlackBytes readable bytes after its end.
HJ_HD uint64_t stream_window(const uint8_t* s, uint32_t pos)
{
    const uint8_t* p = s + (pos >> 3);
    uint64_t hi = ((uint64_t)p[0] << 56) | ((uint64_t)p[1] << 48) | ((uint64_t)p[2] << 40) | ((uint64_t)p[3] << 32) | ((uint64_t)p[4] << 24) |
                  ((uint64_t)p[5] << 16) | ((uint64_t)p[6] << 8) | (uint64_t)p[7];
    uint32_t sh = pos & 7;
    return sh ? ((hi << sh) | ((uint64_t)p[8] >> (8 - sh))) : hi;
}

// Decodes one Huffman symbol from the top of `w`.  Returns symbol, *len = code length; invalid code -> returns -1, *len = 1
// (a deterministic choice: desynchronised lanes may well run into bit patterns that are no code at all).
template <class TableRef>
HJ_HD int huff_symbol(const TableRef& t, uint64_t w, int* len)
{
    const uint32_t e = t.fast[(uint32_t)(w >> (64 - kHuffFastBits))];
    if (e) {
        *len = (int)(e >> 8);
        return (int)(e & 255);
    }
    const uint32_t code16 = (uint32_t)(w >> 48);
    for (int l = kHuffFastBits + 1; l <= 16; l++) {
        const int32_t code = (int32_t)(code16 >> (16 - l));
        if (code <= t.maxcode[l]) {
            *len = l;
            return t.vals[(code + t.valoff[l]) & 255];
        }
    }
    *len = 1;
    return -1;
}

// Decodes the symbols that START in [begin, limit) (and before total_bits), beginning in state (z, k).
//   WRITE == false: only tracks the state and counts completed blocks.
//   WRITE == true : also stores coefficients; `block` = index (in MCU order over the whole scan) of the block the first symbol
//                   belongs to.  DC differences go to position 0 of each block (the dc kernel integrates them later).
// `tables` is indexable by slot (0..7) and yields something with .fast/.maxcode/.valoff/.vals.
template <bool WRITE, class Tables>
HJ_HD SubseqState decode_subsequence(const HuffImage& im, const Tables& tables, uint32_t begin, uint32_t limit, int z, int k, uint32_t block,
                                     uint32_t* error)
{
    uint32_t pos = begin;
    uint32_t nblocks = 0;
    const uint32_t total_bits = im.total_bits;
    const int bpm = (int)im.blocks_per_mcu;
    int16_t* blk = nullptr;
    if (WRITE && block < im.total_blocks) {
        const uint32_t mcu = block / (uint32_t)bpm;  // == (block - k) / bpm
        const uint32_t my = mcu / im.mcus_x, mx = mcu - my * im.mcus_x;
        const int c = im.k_comp[k];
        blk = im.coef[c] + ((uint64_t)(my * im.comp_v[c] + im.k_dy[k]) * im.blocks_w[c] + (mx * im.comp_h[c] + im.k_dx[k])) * 64;
    }
    while (pos < limit && pos < total_bits) {
        const uint64_t w = stream_window(im.stream, pos);
        int len;
        if (z == 0) {
            const int s = huff_symbol(tables[im.k_dc[k]], w, &len);
            int nb = s;
            if (s < 0 || s > 15) {
                nb = 0;
                if (WRITE && blk) *error = 1;
            }
            if (WRITE && blk) {
                int v = 0;
                if (nb) {
                    v = (int)((w << len) >> (64 - nb));
                    if (v < (1 << (nb - 1))) v = v - (1 << nb) + 1;
                }
                blk[0] = (int16_t)v;
            }
            pos += (uint32_t)(len + nb);
            z = 1;
        } else {
            const int rs = huff_symbol(tables[im.k_ac[k]], w, &len);
            int r = 0, nb = 0;
            if (rs < 0) {
                if (WRITE && blk) *error = 1;
                z = 64;  // treat as end of block
            } else {
                r = rs >> 4;
                nb = rs & 15;
                if (nb == 0) {
                    z = (r == 15) ? z + 16 : 64;
                } else {
                    z += r;
                    if (z <= 63) {
                        if (WRITE && blk) {
                            int v = (int)((w << len) >> (64 - nb));
                            if (v < (1 << (nb - 1))) v = v - (1 << nb) + 1;
                            blk[zigzag_to_device(z)] = (int16_t)v;
                        }
                        z++;
                    } else if (WRITE && blk) {
                        *error = 1;  // run past the end of the block
                    }
                }
            }
            pos += (uint32_t)(len + nb);
        }
        if (z >= 64) {
            z = 0;
            nblocks++;
            block++;
            if (++k == bpm) k = 0;
            if (WRITE) {
                if (block < im.total_blocks) {
                    const uint32_t mcu = block / (uint32_t)bpm;
                    const uint32_t my = mcu / im.mcus_x, mx = mcu - my * im.mcus_x;
                    const int c = im.k_comp[k];
                    blk = im.coef[c] + ((uint64_t)(my * im.comp_v[c] + im.k_dy[k]) * im.blocks_w[c] + (mx * im.comp_h[c] + im.k_dx[k])) * 64;
                } else {
                    blk = nullptr;
                }
            }
        }
    }
    SubseqState st;
    st.end_bit = pos;
    st.zk = (uint16_t)((k << 8) | z);
    st.nblocks = (uint16_t)(nblocks > 0xFFFF ? 0xFFFF : nblocks);
    return st;
}

}  // namespace hipjpeg
