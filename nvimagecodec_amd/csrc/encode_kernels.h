// encode_kernels.h -- host-callable launcher of the encode kernel (encode_kernels.hip); stream = hipStream_t as void*.
#pragma once
#include "encode_layout.h"

namespace hipjpeg {
int launch_forward(const EncodeImage* images, const EncodeUnit* units, int nunits, void* stream);
}
