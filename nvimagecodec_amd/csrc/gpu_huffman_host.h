// gpu_huffman_host.h -- host-side preparation for the GPU entropy decoder (stream destuffing, table expansion, image
// descriptors) and a host emulation of the complete multi-pass algorithm, built from the same core as the kernels.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "huffman_gpu_core.h"
#include "jpeg_syntax.h"

namespace hipjpeg {

// Baseline/extended sequential frames with ONE scan that interleaves every component (or a single-component frame), no
// restart markers, lookup tables that fit kMaxPoolWords.  Everything else takes the host entropy stage.
bool gpu_entropy_eligible(const FrameInfo& f);

// Upper bound of the destuffed size (+ slack) for staging allocation.
inline size_t destuffed_capacity(const ScanHeader& sc) { return (((sc.data_end - sc.data_begin) + 3) & ~(size_t)3) + kStreamSlackBytes; }

// Removes byte stuffing (FF 00 -> FF) and fill bytes from the scan's entropy-coded segment; appends kStreamSlackBytes of
// 0xFF.  Returns the number of real bytes written.
size_t destuff_scan(const uint8_t* data, const ScanHeader& sc, uint8_t* out);

// Number of uint16 lookup-table entries the scan's Huffman tables expand to (first-level tables of the DC/AC tables the
// scan references + one 64-entry second-level table per 10-bit prefix that continues into longer codes); 0 = a table is
// missing or malformed (over-subscribed code, DC symbol > 15).  gpu_entropy_eligible() uses it.
size_t gpu_pool_words(const ScanHeader& sc);

// Expands the tables into `pool` (gpu_pool_words(sc) entries) and records the per-position table offsets in im->k[].tdc/tac.
// Call after fill_huff_image().
void build_gpu_pool(const ScanHeader& sc, HuffImage* im, uint16_t* pool);

// Fills every field except the pointers (stream, pool, coef, dc_diff), first_subseq and the table offsets.
void fill_huff_image(const FrameInfo& f, uint32_t stream_bytes, HuffImage* im);

// Runs pass 0, the synchronisation passes, the block-count scan, the write pass and the DC integration on the host, one
// "lane" after the other.  coef[c] = device-layout blocks (as entropy_decode.h).  Returns 0 on success, else the status the
// kernels would report; *sync_passes receives the number of passes until the fixpoint.
int emulate_gpu_entropy(const uint8_t* data, size_t size, const FrameInfo& f, int16_t* const coef[4], int* sync_passes);

}  // namespace hipjpeg
