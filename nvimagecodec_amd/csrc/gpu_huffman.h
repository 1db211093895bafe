// gpu_huffman.h -- host-callable launchers of the GPU entropy kernels (gpu_huffman.hip); stream = hipStream_t as void*.
#pragma once
#include <cstdint>

#include "huffman_gpu_core.h"

namespace hipjpeg {

// sync/write kernels: `first` = first subsequence (inside the image) the workgroup owns (it owns kHuffOwn of them); dc kernel: `first` = component index
struct HuffUnit {
    uint32_t image;  // index into HuffImage[]
    uint32_t first;
};

// Byte-stuffing removal (count + compact): chunk_units[i] = {image, chunk index inside the image}; drops = one uint32 per chunk.
// Writes HuffImage::stream contents and total_bits / num_subseq / stream_words.
// count_on_device = false: `drops` already holds the per-chunk counts (from the host's marker walk)
// counters: 64 words the compact kernel clears (the stage's convergence counters)
// Zero-copy input: copies the scans of the images whose HuffImage::raw_src is set from the caller's pinned host memory to HuffImage::raw
// (same chunk units as the destuff kernels; images without raw_src are skipped).
int launch_gather_raw(const HuffImage* images, const HuffUnit* chunk_units, int nchunks, void* stream);
int launch_destuff(HuffImage* images, const HuffUnit* chunk_units, int nchunks, uint32_t* drops, bool count_on_device, unsigned int* counters,
                   void* stream);

// pool_bytes = dynamic LDS for the lookup tables: 2 * the largest HuffImage::pool_words of the batch
// units[i].first = first owned subsequence of workgroup i, a multiple of kHuffOwn; incoming = one uint64 per unit
// max_rounds / tail_tasks / tail_count: after max_rounds correction rounds the workgroup hands what is left (tail_tasks:
// kTailTaskBytes per unit, tail_count: one uint32 per unit) to the tail kernel, launched right behind it; tail_count == nullptr:
// the workgroup iterates to its fixpoint itself.
// first_pass == 0: a ripple launch (corrections across group borders), pass_id = its number (1, 2, ...): HuffImage::moved_pass
// receives it for every image in which a group's own last end state still changed.  HuffImage::gave_up is set for images with a
// group that exceeded its round budget (periodic streams); such images are skipped by later ripple launches.
int launch_huff_sync(HuffImage* images, const HuffUnit* units, int nunits, unsigned long long* states, unsigned long long* incoming,
                     unsigned int* changed, int first_pass, int max_rounds, uint16_t* tail_tasks, uint32_t* tail_count, unsigned pool_bytes, void* stream,
                     unsigned pass_id = 0);
int launch_huff_scan(HuffImage* images, const uint32_t* image_list, int nimages, const unsigned long long* states, uint32_t* first_block, void* stream);
// Write pass: position kernel over the sync units, then the block kernel over block_units ({image, first MCU}, kHuffMcusPerWg
// MCUs each).
// group_sums: four int32 per block unit -- the sums of the DC differences of its MCUs, per component (what the DC pass needs
// from the groups in front of a group).
// dc_only: the pixel kernels decode the blocks themselves (decode_kernels.hip FUSED builds); only the DC differences are read here.
int launch_huff_write(HuffImage* images, const HuffUnit* sync_units, int nsync_units, const HuffUnit* block_units, int nblock_units,
                      const unsigned long long* states, const uint32_t* first_block, int32_t* group_sums, unsigned pool_bytes, void* stream,
                      bool dc_only = false);
// DC differences -> DC planes.  Images without restart intervals: one workgroup per block unit (its base = the sums of the
// units in front of it); images with restart intervals: one workgroup per (image, component) in rst_units.
int launch_huff_dc(const HuffImage* images, const HuffUnit* rst_units, int nrst_units, const HuffUnit* block_units, int nblock_units,
                   const int32_t* group_sums, void* stream);

}  // namespace hipjpeg
