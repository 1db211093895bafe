// decode_kernels.h -- host-callable launchers of the decode kernels (implemented in decode_kernels.hip).
// `stream` is a hipStream_t passed as void* so that plain C++ translation units need no HIP headers.
#pragma once
#include "device_layout.h"

namespace hipjpeg {

// Pass-1 arithmetic of the IDCT, chosen per image by the host (DecodeBatch::finalize): 24-bit-multiplier butterflies, the
// 32-bit-multiplier build for images flagged kFlagExactMul32, packed int16 dot products for images flagged kFlagFitsInt16
// (separate kernels keep the common case's register count down).
enum PlaneFlavour { kPlaneMul24 = 0, kPlaneExact = 1, kPlanePk16 = 2, kNumPlaneFlavours = 3 };
// K1: IDCT of component blocks into u8 planes.  One WorkUnit = 128 blocks.
int launch_idct_plane(int pass1, const DecodeImage* images, const WorkUnit* units, int nunits, void* stream);
// K2: fused luma IDCT + chroma upsample (factors hs x vs, 0 = no chroma) + colour conversion + store.
// flavour: which instantiation of the fused luma kernel (decode_kernels.hip luma_color_body)
// = pass-1 arithmetic (PlaneFlavour) x 3 + layout: 0 = generic (format flags read at run time), 1 = COMMON (YCbCr source, interleaved
// RGB / BGR, fancy upsampling), 2 = COMMON with planar RGB / BGR output
constexpr int kNumLumaLayouts = 3, kNumLumaFlavours = kNumPlaneFlavours * kNumLumaLayouts;
constexpr int luma_flavour(int pass1, int layout) { return pass1 * kNumLumaLayouts + layout; }
int launch_luma_color(int flavour, int hs, int vs, const DecodeImage* images, const WorkUnit* units, int nunits, void* stream);
// K3: per-pixel colour stage from planes (replication upsampling) for uncommon sampling layouts.
int launch_generic_color(const DecodeImage* images, const WorkUnit* units, int nunits, void* stream);
// Four-component frames (CMYK / YCCK): upsampling + the reference's CMYK -> RGB step, one WorkUnit per pixel row.
int launch_cmyk_color(const DecodeImage* images, const WorkUnit* units, int nunits, void* stream);
// Geometry pass (region of interest + EXIF orientation): WorkUnit{image = TransformImage index, block_base = first output row}.
int launch_transform(const TransformImage* images, const WorkUnit* units, int nunits, void* stream);

}  // namespace hipjpeg
