// plugin_objects.h -- the factory objects whose function tables this extension registers with the framework.
// Their nvimgcodecDecoderDesc_t / nvimgcodecEncoderDesc_t live inside the objects and outlive the registration:
// the framework keeps the desc pointer (reference src/image_decoder_factory.cpp:26-34).
#pragma once
#include "../../include/nvimgcodec_abi.h"

namespace hipjpeg_ext {

class HipJpegDecoderPlugin {
public:
    explicit HipJpegDecoderPlugin(const nvimgcodecFrameworkDesc_t* framework);
    const nvimgcodecDecoderDesc_t* desc() const { return &desc_; }

private:
    static nvimgcodecStatus_t static_create(void* instance, nvimgcodecDecoder_t* decoder, const nvimgcodecExecutionParams_t* exec_params,
                                            const char* options);
    static nvimgcodecStatus_t static_destroy(nvimgcodecDecoder_t decoder);
    static nvimgcodecStatus_t static_can_decode(nvimgcodecDecoder_t decoder, nvimgcodecProcessingStatus_t* status,
                                                nvimgcodecCodeStreamDesc_t** code_streams, nvimgcodecImageDesc_t** images, int batch_size,
                                                const nvimgcodecDecodeParams_t* params);
    static nvimgcodecStatus_t static_decode(nvimgcodecDecoder_t decoder, nvimgcodecCodeStreamDesc_t** code_streams, nvimgcodecImageDesc_t** images,
                                            int batch_size, const nvimgcodecDecodeParams_t* params);
    nvimgcodecDecoderDesc_t desc_;
    const nvimgcodecFrameworkDesc_t* framework_;
};

class HipJpegEncoderPlugin {
public:
    explicit HipJpegEncoderPlugin(const nvimgcodecFrameworkDesc_t* framework);
    const nvimgcodecEncoderDesc_t* desc() const { return &desc_; }

private:
    static nvimgcodecStatus_t static_create(void* instance, nvimgcodecEncoder_t* encoder, const nvimgcodecExecutionParams_t* exec_params,
                                            const char* options);
    static nvimgcodecStatus_t static_destroy(nvimgcodecEncoder_t encoder);
    static nvimgcodecStatus_t static_can_encode(nvimgcodecEncoder_t encoder, nvimgcodecProcessingStatus_t* status, nvimgcodecImageDesc_t** images,
                                                nvimgcodecCodeStreamDesc_t** code_streams, int batch_size, const nvimgcodecEncodeParams_t* params);
    static nvimgcodecStatus_t static_encode(nvimgcodecEncoder_t encoder, nvimgcodecImageDesc_t** images, nvimgcodecCodeStreamDesc_t** code_streams,
                                            int batch_size, const nvimgcodecEncodeParams_t* params);
    nvimgcodecEncoderDesc_t desc_;
    const nvimgcodecFrameworkDesc_t* framework_;
};

}  // namespace hipjpeg_ext
