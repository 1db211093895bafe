// decode_kernels.h -- host-callable launchers of the decode kernels (implemented in decode_kernels.hip).
// `stream` is a hipStream_t passed as void* so that plain C++ translation units need no HIP headers.
#pragma once
#include "device_layout.h"

namespace hipjpeg {

// K1: IDCT of component blocks into u8 planes.  One WorkUnit = 128 blocks.  (Rounds 1-2 had three builds of K1 and K2 for three
// pass-1 arithmetics chosen per image; since round 3 the transform is the SIMD routine's int16 arithmetic for every image.)
int launch_idct_plane(const DecodeImage* images, const WorkUnit* units, int nunits, void* stream);
// K2: fused luma IDCT + chroma upsample (factors hs x vs, 0 = no chroma) + colour conversion + store.
// layout: which instantiation of the fused luma kernel (decode_kernels.hip luma_color_body): 0 = generic (format flags read at run
// time), 1 = COMMON (YCbCr source, interleaved RGB / BGR, fancy upsampling), 2 = COMMON with planar RGB / BGR output
constexpr int kNumLumaLayouts = 3;
int launch_luma_color(int layout, int hs, int vs, const DecodeImage* images, const WorkUnit* units, int nunits, void* stream);
// FUSED builds of K1 / K2 for images of the GPU entropy stage (baseline): the blocks are Huffman-decoded inside the kernel, into the LDS
// slots the IDCT reads, from the start positions the position pass recorded (HuffImage::block_pos) -- no coefficient block is written
// to or read from HBM.  himages = the batch's HuffImage array (DecodeImage::huff_index points into it), pool_bytes = dynamic LDS for
// the largest lookup-table pool of the batch.  Same work units as the plain builds.
struct HuffImage;
int launch_idct_plane_fused(const DecodeImage* images, const WorkUnit* units, int nunits, HuffImage* himages, unsigned pool_bytes, void* stream);
int launch_luma_color_fused(int layout, int hs, int vs, const DecodeImage* images, const WorkUnit* units, int nunits, HuffImage* himages, unsigned pool_bytes,
                            void* stream);
// K3: per-pixel colour stage from planes (replication upsampling) for uncommon sampling layouts.
int launch_generic_color(const DecodeImage* images, const WorkUnit* units, int nunits, void* stream);
// Four-component frames (CMYK / YCCK): upsampling + the reference's CMYK -> RGB step, one WorkUnit per pixel row.
int launch_cmyk_color(const DecodeImage* images, const WorkUnit* units, int nunits, void* stream);
// Geometry pass (region of interest + EXIF orientation): WorkUnit{image = TransformImage index, block_base = first output row}.
int launch_transform(const TransformImage* images, const WorkUnit* units, int nunits, void* stream);

}  // namespace hipjpeg
