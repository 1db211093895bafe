// progressive_gpu.h -- host-callable launchers of the progressive-scan kernels (progressive_gpu.hip); stream = hipStream_t as void*.
#pragma once
#include <cstdint>

#include "gpu_huffman.h"
#include "progressive_gpu_core.h"

namespace hipjpeg {

// slot_words = uint16 entries reserved per lookup table in LDS: the largest table of the batch (<= kProgTableMax).
// himgs = the HuffImage array the destuff kernels filled (one entry per scan: ProgScan::huff_image).
// slots = table slots of a walk workgroup: the longest AC chain of the batch, at least the components of its DC scans (<= kProgMaxStages)
// One workgroup per image: a wave for the DC scans + ac_waves waves (the most AC scans any image of the batch has), dc_slots table
// slots for the DC scans' tables + one per AC wave, `rings` hand-over rings (the most any image needs: sum over its components of
// AC scans - 1).
int launch_prog_walk(ProgImage* images, const HuffImage* himgs, int nimages, unsigned slot_words, unsigned dc_slots, unsigned ac_waves, unsigned rings,
                     void* stream);
// units[i] = {image, (component << 28) | first block of the component's allocation grid}; 256 blocks per unit.
int launch_prog_replay(ProgImage* images, const HuffImage* himgs, const HuffUnit* units, int nunits, unsigned slot_words, void* stream);

}  // namespace hipjpeg
