// progressive_gpu.h -- host-callable launchers of the progressive-scan kernels (progressive_gpu.hip); stream = hipStream_t as void*.
#pragma once
#include <cstdint>

#include "gpu_huffman.h"
#include "progressive_gpu_core.h"

namespace hipjpeg {

// slot_words = uint16 entries reserved per lookup table in LDS: the largest table of the batch (<= kProgTableMax).
// himgs = the HuffImage array the destuff kernels filled (one entry per scan: ProgScan::huff_image).
// slots = table slots of a walk workgroup: the longest AC chain of the batch, at least the components of its DC scans (<= kProgMaxStages)
// waves = waves per workgroup: the longest chain of scans any image of the batch has (a workgroup holds its wave slots until its last wave ends)
int launch_prog_walk(ProgImage* images, const HuffImage* himgs, int nimages, unsigned slot_words, unsigned slots, unsigned waves, void* stream);
// units[i] = {image, (component << 28) | first block of the component's allocation grid}; 256 blocks per unit.
int launch_prog_replay(ProgImage* images, const HuffImage* himgs, const HuffUnit* units, int nunits, unsigned slot_words, void* stream);

}  // namespace hipjpeg
