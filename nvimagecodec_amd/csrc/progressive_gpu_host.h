// progressive_gpu_host.h -- host-side preparation for progressive scans on the GPU entropy stage (eligibility, lookup tables,
// image descriptors) and a host emulation of the walk + replay algorithm built from the same parse logic as the kernels
// (progressive_gpu_core.h).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "jpeg_syntax.h"
#include "progressive_gpu_core.h"

namespace hipjpeg {

// SOF2 frames whose every scan is plainly stuffed (no restart markers, no fill bytes), whose scan script is a consistent
// successive-approximation progression (every coefficient: first scan with Ah = 0, then Ah = previous Al, Al = Ah - 1), with
// at most kProgMaxScans scans, kProgMaxStages AC scans per component and lookup tables of at most kProgTableMax entries.
// Everything else keeps the host entropy stage.
bool gpu_progressive_eligible(const FrameInfo& f);

// uint16 entries of the lookup table for `s` (256 first-level + 256 per 8-bit prefix that continues); 0 = malformed.
size_t prog_table_words(const HuffSpec& s);
// Expands `s` at out[0 .. prog_table_words(s)).
void build_prog_table(const HuffSpec& s, uint16_t* out);

// Entries of all tables of the frame's scans (each table starts at a multiple of 64 entries).
size_t prog_pool_words(const FrameInfo& f);
// Fills every field of *im except the pointers (scan[].stream, scan[].block_pos, coef, dc_plane, pool) and scan[].huff_image;
// writes the lookup tables to pool[0 .. prog_pool_words(f)).
void fill_prog_image(const FrameInfo& f, ProgImage* im, uint16_t* pool);

// The kernels' algorithm on the host: walk every scan (block start positions), replay every block, integrate the DC scans.
// coef[c] = device-layout blocks with the DC coefficient in place (as entropy_decode.h).  Returns 0 on success, 1 when the
// walk or the replay rejects the stream.
int emulate_gpu_progressive(const uint8_t* data, size_t size, const FrameInfo& f, int16_t* const coef[4]);

}  // namespace hipjpeg
