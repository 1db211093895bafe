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

// What a frame asks of the walk / replay launches (they are sized for the batch's maxima, decoder_core.cpp enqueue_progressive): the
// largest lookup table in uint16 entries, table slots for its DC scans, AC scans (a wave each), hand-over rings; and the dynamic LDS
// such a launch takes.  The planner keeps the batch's need below the device's workgroup limit by leaving the most demanding
// images to the host entropy stage (a refused launch would fail the whole batch).
struct ProgLdsShape {
    unsigned slot_words = 0, dc_slots = 1, ac_waves = 0, rings = 0;
    void merge(const ProgLdsShape& o)
    {
        slot_words = slot_words > o.slot_words ? slot_words : o.slot_words;
        dc_slots = dc_slots > o.dc_slots ? dc_slots : o.dc_slots;
        ac_waves = ac_waves > o.ac_waves ? ac_waves : o.ac_waves;
        rings = rings > o.rings ? rings : o.rings;
    }
};
ProgLdsShape prog_lds_shape(const FrameInfo& f);
constexpr size_t kProgRingBytes = (size_t)kProgRing * kProgGroup * 8;   // sizeof(WalkRing), progressive_gpu.hip
constexpr size_t kProgReplayStaticLds = 40 * 1024;                        // upper bound of prog_replay_kernel's static LDS (ReplayShared)
inline size_t prog_slot_words(const ProgLdsShape& s) { return ((s.slot_words > 256u ? s.slot_words : 256u) + 63u) & ~(size_t)63; }
inline size_t prog_walk_lds_bytes(const ProgLdsShape& s) { return (size_t)(s.dc_slots + s.ac_waves) * prog_slot_words(s) * 2 + s.rings * kProgRingBytes; }
inline size_t prog_replay_lds_bytes(const ProgLdsShape& s) { return kProgReplayStaticLds + prog_slot_words(s) * 2 * kProgMaxStages; }

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
