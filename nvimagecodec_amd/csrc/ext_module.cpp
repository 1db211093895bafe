// ext_module.cpp -- extension entry point: the one symbol nvImageCodec's plugin framework looks up after dlopen
// (reference src/plugin_framework.cpp:309-351, include/nvimgcodec.h:1356-1364; pattern of
// extensions/libjpeg_turbo/libjpeg_turbo_ext.cpp:28-100).
#include <new>

#include "plugin_common.h"
#include "plugin_objects.h"

namespace hipjpeg_ext {

struct HipJpegExtension {
    explicit HipJpegExtension(const nvimgcodecFrameworkDesc_t* fw) : framework(fw), decoder(fw), encoder(fw)
    {
        framework->registerEncoder(framework->instance, encoder.desc(), NVIMGCODEC_PRIORITY_HIGH);  // nvjpeg_ext.cpp:42
        // same slot in the priority chain as the reference's nvJPEG CUDA decoder (nvjpeg_ext.cpp:45):
        // ahead of libjpeg_turbo (NORMAL) and opencv (LOW), which stay available as fallbacks
        framework->registerDecoder(framework->instance, decoder.desc(), NVIMGCODEC_PRIORITY_HIGH);
    }
    ~HipJpegExtension()
    {
        framework->unregisterEncoder(framework->instance, encoder.desc());
        framework->unregisterDecoder(framework->instance, decoder.desc());
    }
    const nvimgcodecFrameworkDesc_t* framework;  // valid until destroy()
    HipJpegDecoderPlugin decoder;
    HipJpegEncoderPlugin encoder;
};

static nvimgcodecStatus_t extension_create(void* /*instance*/, nvimgcodecExtension_t* extension, const nvimgcodecFrameworkDesc_t* framework)
{
    try {
        if (!extension || !framework) return NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER;
        if (!framework->registerDecoder || !framework->unregisterDecoder || !framework->registerEncoder || !framework->unregisterEncoder) return NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER;
        HJ_LOG_TRACE(framework, kExtensionId, "extension_create");
        *extension = reinterpret_cast<nvimgcodecExtension_t>(new HipJpegExtension(framework));
        return NVIMGCODEC_STATUS_SUCCESS;
    } catch (const std::bad_alloc&) {
        return NVIMGCODEC_STATUS_EXTENSION_ALLOCATOR_FAILURE;
    } catch (...) {
        return NVIMGCODEC_STATUS_EXTENSION_INTERNAL_ERROR;
    }
}

static nvimgcodecStatus_t extension_destroy(nvimgcodecExtension_t extension)
{
    try {
        if (!extension) return NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER;
        delete reinterpret_cast<HipJpegExtension*>(extension);
        return NVIMGCODEC_STATUS_SUCCESS;
    } catch (...) {
        return NVIMGCODEC_STATUS_EXTENSION_INTERNAL_ERROR;
    }
}

}  // namespace hipjpeg_ext

extern "C" NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecExtensionModuleEntry(nvimgcodecExtensionDesc_t* ext_desc)
{
    if (!ext_desc) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    if (ext_desc->struct_type != NVIMGCODEC_STRUCTURE_TYPE_EXTENSION_DESC) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    ext_desc->struct_size = sizeof(nvimgcodecExtensionDesc_t);
    ext_desc->struct_next = nullptr;
    ext_desc->instance = nullptr;
    ext_desc->id = hipjpeg_ext::kExtensionId;
    ext_desc->version = hipjpeg_ext::kExtensionVersion;
    ext_desc->ext_api_version = NVIMGCODEC_EXT_API_VER;
    ext_desc->create = hipjpeg_ext::extension_create;
    ext_desc->destroy = hipjpeg_ext::extension_destroy;
    return NVIMGCODEC_STATUS_SUCCESS;
}
