// plugin_common.h -- helpers shared by the nvImageCodec plugin objects of this extension.
#pragma once
#include <sstream>
#include <string>
#include <vector>

#include "../../include/nvimgcodec_abi.h"

namespace hipjpeg_ext {

constexpr const char* kExtensionId = "hipjpeg_ext";
constexpr const char* kDecoderId = "hipjpeg_decoder";
constexpr const char* kEncoderId = "hipjpeg_encoder";
constexpr uint32_t kExtensionVersion = NVIMGCODEC_MAKE_VERSION(0, 1, 0);

// Logging goes through the framework's log hook (reference extensions/libjpeg_turbo/log.h:30-40).
inline void log_message(const nvimgcodecFrameworkDesc_t* fw, const char* codec_id, nvimgcodecDebugMessageSeverity_t severity,
                        const std::string& msg)
{
    if (!fw || !fw->log) return;
    nvimgcodecDebugMessageData_t data{NVIMGCODEC_STRUCTURE_TYPE_DEBUG_MESSAGE_DATA, sizeof(nvimgcodecDebugMessageData_t), nullptr,
                                      msg.c_str(), 0, "jpeg", codec_id, kExtensionVersion};
    fw->log(fw->instance, severity, NVIMGCODEC_DEBUG_MESSAGE_CATEGORY_GENERAL, &data);
}

#define HJ_LOG(fw, id, sev, expr)                              \
    do {                                                       \
        std::ostringstream _ss;                                \
        _ss << expr;                                           \
        ::hipjpeg_ext::log_message(fw, id, sev, _ss.str());    \
    } while (0)
#define HJ_LOG_ERROR(fw, id, expr) HJ_LOG(fw, id, NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_ERROR, expr)
#define HJ_LOG_WARNING(fw, id, expr) HJ_LOG(fw, id, NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_WARNING, expr)
#define HJ_LOG_DEBUG(fw, id, expr) HJ_LOG(fw, id, NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_DEBUG, expr)
#define HJ_LOG_TRACE(fw, id, expr) HJ_LOG(fw, id, NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_TRACE, expr)

// "<module>:<key>=<value> ..." option strings; an empty module applies to every plugin
// (same grammar as extensions/libjpeg_turbo/libjpeg_turbo_decoder.cpp:250-276).
template <typename F>
inline void for_each_option(const char* options, const char* module_name, F&& fn, std::vector<std::string>* addressed = nullptr)
{
    std::istringstream iss(options ? options : "");
    std::string token;
    while (std::getline(iss, token, ' ')) {
        auto colon = token.find(':');
        auto equal = token.find('=');
        if (colon == std::string::npos || equal == std::string::npos || colon > equal) continue;
        std::string module = token.substr(0, colon);
        if (!module.empty() && module != module_name) continue;
        if (addressed && !module.empty()) addressed->push_back(token.substr(colon + 1, equal - colon - 1));  // named this very module
        fn(token.substr(colon + 1, equal - colon - 1), token.substr(equal + 1));
    }
}

// Walks a struct_next chain looking for a given structure type.
template <typename T>
inline T* find_in_chain(void* next, nvimgcodecStructureType_t type)
{
    struct Head {
        nvimgcodecStructureType_t struct_type;
        size_t struct_size;
        void* struct_next;
    };
    Head* h = static_cast<Head*>(next);
    while (h && h->struct_type != type) h = static_cast<Head*>(h->struct_next);
    return reinterpret_cast<T*>(h);
}

}  // namespace hipjpeg_ext
