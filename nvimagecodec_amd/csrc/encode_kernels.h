// encode_kernels.h -- host-callable launcher of the encode kernel (encode_kernels.hip); stream = hipStream_t as void*.
#pragma once
#include "encode_layout.h"

namespace hipjpeg {
int launch_forward(const EncodeImage* images, const EncodeUnit* units, int nunits, void* stream);
// two-lanes-per-block kernel: interleaved or planar RGB/BGR input, three components, (hs, vs) = (2,2), (2,1) or (1,1)
int launch_forward_pair(int hs, int vs, bool planar, const EncodeImage* images, const EncodeUnit* units, int nunits, void* stream);
// pre-subsampled planar YCbCr input: one lane per block of a component plane; unit = {image, first block, -, component}
int launch_forward_planes(const EncodeImage* images, const EncodeUnit* units, int nunits, void* stream);
}
