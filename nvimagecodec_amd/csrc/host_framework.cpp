// host_framework.cpp -- a small host harness that implements the application-side nvImageCodec C API
// (include/nvimgcodec_abi.h, "public API" section) for the JPEG path, so that the extension can be loaded, driven,
// tested and benchmarked on a box where the real nvImageCodec core cannot be built (it needs the CUDA toolkit).
//
// It mirrors, for this path only, the behaviour of the reference core (all paths relative to /root/reference):
//   registry + priorities        src/codec.cpp:119-124, src/codec_registry.cpp          (multimap<priority, factory>)
//   extension loading            src/plugin_framework.cpp:191-351                        (version gate, create(), entry symbol)
//   code streams / io streams    src/code_stream.cpp, src/mem_io_stream.h, src/std_file_io_stream.cpp
//   JPEG stream info             src/parsers/jpeg.cpp:202-361                            (what canDecode sees)
//   decode dispatch + fallback   src/image_generic_decoder.cpp:181-285, src/decoder_worker.cpp:158-307
//   bounce buffers               src/work.h:144-190
//   futures                      src/processing_results.cpp:34-146
//   default executor             src/default_executor.cpp:45-58, src/thread_pool.cpp
// It is NOT a re-implementation of the product's other codecs, parsers or tools.
#include <dirent.h>
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nvimgcodec_abi.h"
#include "../../include/hipjpeg.h"
#include "jpeg_syntax.h"

using namespace hipjpeg;

namespace {

// ---------------------------------------------------------------------------------------------- thread pool / executor
class WorkerPool {
public:
    WorkerPool(int device_id, int num_threads) : device_id_(device_id)
    {
        if (num_threads <= 0) num_threads = (int)std::thread::hardware_concurrency();
        if (num_threads <= 0) num_threads = 1;
        for (int t = 0; t < num_threads; t++) threads_.emplace_back([this, t] { run(t); });
    }
    ~WorkerPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    int size() const { return (int)threads_.size(); }
    void post(std::function<void(int)> fn)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back(std::move(fn));
        }
        cv_.notify_one();
    }

private:
    void run(int tid)
    {
        // pool threads of a GPU device have that device current (reference src/thread_pool.cpp:130)
        if (device_id_ >= 0) (void)hipSetDevice(device_id_);
        for (;;) {
            std::function<void(int)> fn;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                fn = std::move(q_.front());
                q_.pop_front();
            }
            try {
                fn(tid);
            } catch (...) {
                // plugin tasks must report failures through imageReady themselves
            }
        }
    }
    int device_id_;
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::function<void(int)>> q_;
    bool stop_ = false;
};

struct DefaultExecutor {
    explicit DefaultExecutor(int device_id, int num_threads) : pool(device_id, num_threads)
    {
        desc = {NVIMGCODEC_STRUCTURE_TYPE_EXECUTOR_DESC, sizeof(nvimgcodecExecutorDesc_t), nullptr, this, &launch, &get_num_threads};
    }
    static nvimgcodecStatus_t launch(void* instance, int /*device_id*/, int sample_idx, void* ctx, void (*task)(int, int, void*))
    {
        auto* self = static_cast<DefaultExecutor*>(instance);
        self->pool.post([=](int tid) { task(tid, sample_idx, ctx); });
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    static int get_num_threads(void* instance) { return static_cast<DefaultExecutor*>(instance)->pool.size(); }
    WorkerPool pool;
    nvimgcodecExecutorDesc_t desc;
};

int usable_cpus()
{
    int n = (int)std::thread::hardware_concurrency();
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            long q = atol(quota) / period;
            if (q >= 1 && q < n) n = (int)q;
        }
        fclose(f);
    }
    return n > 0 ? n : 1;
}

// ---------------------------------------------------------------------------------------------- io streams
struct IoStream {
    virtual ~IoStream() {}
    virtual nvimgcodecStatus_t read(size_t* out, void* buf, size_t bytes) = 0;
    virtual nvimgcodecStatus_t write(size_t* out, void* buf, size_t bytes) = 0;
    virtual nvimgcodecStatus_t seek(ptrdiff_t off, int whence) = 0;
    virtual nvimgcodecStatus_t tell(ptrdiff_t* off) = 0;
    virtual nvimgcodecStatus_t size(size_t* s) = 0;
    virtual nvimgcodecStatus_t reserve(size_t) { return NVIMGCODEC_STATUS_SUCCESS; }
    virtual nvimgcodecStatus_t flush() { return NVIMGCODEC_STATUS_SUCCESS; }
    virtual nvimgcodecStatus_t map(void** buffer, size_t, size_t)
    {
        *buffer = nullptr;  // streams that cannot hand out memory return NULL and the plugin falls back to read()
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    virtual nvimgcodecStatus_t unmap(void*, size_t) { return NVIMGCODEC_STATUS_SUCCESS; }
};

struct MemInStream : IoStream {
    MemInStream(const uint8_t* d, size_t n) : data(d), len(n) {}
    nvimgcodecStatus_t read(size_t* out, void* buf, size_t bytes) override
    {
        size_t n = pos < len ? std::min(bytes, len - pos) : 0;
        if (n) memcpy(buf, data + pos, n);
        pos += n;
        if (out) *out = n;
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    nvimgcodecStatus_t write(size_t*, void*, size_t) override { return NVIMGCODEC_STATUS_BAD_CODESTREAM; }
    nvimgcodecStatus_t seek(ptrdiff_t off, int whence) override
    {
        ptrdiff_t base = whence == SEEK_SET ? 0 : (whence == SEEK_CUR ? (ptrdiff_t)pos : (ptrdiff_t)len);
        ptrdiff_t np = base + off;
        if (np < 0 || (size_t)np > len) return NVIMGCODEC_STATUS_BAD_CODESTREAM;
        pos = (size_t)np;
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    nvimgcodecStatus_t tell(ptrdiff_t* off) override
    {
        *off = (ptrdiff_t)pos;
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    nvimgcodecStatus_t size(size_t* s) override
    {
        *s = len;
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    nvimgcodecStatus_t map(void** buffer, size_t offset, size_t sz) override
    {
        *buffer = (offset + sz <= len) ? const_cast<uint8_t*>(data) + offset : nullptr;
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    const uint8_t* data;
    size_t len, pos = 0;
};

struct FileStream : IoStream {
    FileStream(const char* name, const char* mode) { f = fopen(name, mode); }
    ~FileStream() override
    {
        if (f) fclose(f);
    }
    nvimgcodecStatus_t read(size_t* out, void* buf, size_t bytes) override
    {
        size_t n = fread(buf, 1, bytes, f);
        if (out) *out = n;
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    nvimgcodecStatus_t write(size_t* out, void* buf, size_t bytes) override
    {
        size_t n = fwrite(buf, 1, bytes, f);
        if (out) *out = n;
        return n == bytes ? NVIMGCODEC_STATUS_SUCCESS : NVIMGCODEC_STATUS_EXECUTION_FAILED;
    }
    nvimgcodecStatus_t seek(ptrdiff_t off, int whence) override { return fseek(f, off, whence) == 0 ? NVIMGCODEC_STATUS_SUCCESS : NVIMGCODEC_STATUS_BAD_CODESTREAM; }
    nvimgcodecStatus_t tell(ptrdiff_t* off) override
    {
        *off = ftell(f);
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    nvimgcodecStatus_t size(size_t* s) override
    {
        long cur = ftell(f);
        fseek(f, 0, SEEK_END);
        *s = (size_t)ftell(f);
        fseek(f, cur, SEEK_SET);
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    nvimgcodecStatus_t flush() override
    {
        fflush(f);
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    FILE* f = nullptr;
};

// host-memory sink driven by the user's resize callback (reference src/mem_io_stream.h:100-125)
struct MemOutStream : IoStream {
    MemOutStream(void* c, nvimgcodecResizeBufferFunc_t fn) : ctx(c), resize(fn) {}
    nvimgcodecStatus_t read(size_t*, void*, size_t) override { return NVIMGCODEC_STATUS_BAD_CODESTREAM; }
    nvimgcodecStatus_t reserve(size_t bytes) override
    {
        if (bytes > cap) {
            data = resize(ctx, bytes);
            cap = data ? bytes : 0;
        }
        return data ? NVIMGCODEC_STATUS_SUCCESS : NVIMGCODEC_STATUS_ALLOCATOR_FAILURE;
    }
    nvimgcodecStatus_t write(size_t* out, void* buf, size_t bytes) override
    {
        if (pos + bytes > cap && reserve(pos + bytes) != NVIMGCODEC_STATUS_SUCCESS) return NVIMGCODEC_STATUS_ALLOCATOR_FAILURE;
        memcpy(data + pos, buf, bytes);
        pos += bytes;
        written = std::max(written, pos);
        if (out) *out = bytes;
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    nvimgcodecStatus_t seek(ptrdiff_t off, int whence) override
    {
        ptrdiff_t base = whence == SEEK_SET ? 0 : (whence == SEEK_CUR ? (ptrdiff_t)pos : (ptrdiff_t)written);
        if (base + off < 0) return NVIMGCODEC_STATUS_BAD_CODESTREAM;
        pos = (size_t)(base + off);
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    nvimgcodecStatus_t tell(ptrdiff_t* off) override
    {
        *off = (ptrdiff_t)pos;
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    nvimgcodecStatus_t size(size_t* s) override
    {
        *s = written;
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    nvimgcodecStatus_t flush() override
    {
        // shrink the sink to the bytes actually written (reference src/mem_io_stream.h:114-120)
        data = resize(ctx, written);
        cap = written;
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    void* ctx;
    nvimgcodecResizeBufferFunc_t resize;
    uint8_t* data = nullptr;
    size_t cap = 0, pos = 0, written = 0;
};

nvimgcodecIoStreamDesc_t make_io_desc(IoStream* s)
{
    nvimgcodecIoStreamDesc_t d;
    memset(&d, 0, sizeof d);
    d.struct_type = NVIMGCODEC_STRUCTURE_TYPE_IO_STREAM_DESC;
    d.struct_size = sizeof d;
    d.instance = s;
    d.read = [](void* i, size_t* o, void* b, size_t n) { return static_cast<IoStream*>(i)->read(o, b, n); };
    d.write = [](void* i, size_t* o, void* b, size_t n) { return static_cast<IoStream*>(i)->write(o, b, n); };
    d.putc = [](void* i, size_t* o, unsigned char c) { return static_cast<IoStream*>(i)->write(o, &c, 1); };
    d.skip = [](void* i, size_t n) { return static_cast<IoStream*>(i)->seek((ptrdiff_t)n, SEEK_CUR); };
    d.seek = [](void* i, ptrdiff_t o, int w) { return static_cast<IoStream*>(i)->seek(o, w); };
    d.tell = [](void* i, ptrdiff_t* o) { return static_cast<IoStream*>(i)->tell(o); };
    d.size = [](void* i, size_t* s) { return static_cast<IoStream*>(i)->size(s); };
    d.reserve = [](void* i, size_t n) { return static_cast<IoStream*>(i)->reserve(n); };
    d.flush = [](void* i) { return static_cast<IoStream*>(i)->flush(); };
    d.map = [](void* i, void** b, size_t o, size_t n) { return static_cast<IoStream*>(i)->map(b, o, n); };
    d.unmap = [](void* i, void* b, size_t n) { return static_cast<IoStream*>(i)->unmap(b, n); };
    return d;
}

// ---------------------------------------------------------------------------------------------- EXIF orientation (APP1)
// Only the orientation tag (0x0112) of IFD0 is read; mapping = reference src/parsers/exif_orientation.h:36-57.
int exif_orientation_tag(const uint8_t* p, size_t n)
{
    if (n < 14 || memcmp(p, "Exif\0\0", 6) != 0) return 0;
    const uint8_t* t = p + 6;
    size_t tn = n - 6;
    bool le = t[0] == 'I' && t[1] == 'I';
    if (!le && !(t[0] == 'M' && t[1] == 'M')) return 0;
    auto u16 = [&](size_t o) -> unsigned { return o + 2 <= tn ? (le ? t[o] | (t[o + 1] << 8) : (t[o] << 8) | t[o + 1]) : 0; };
    auto u32 = [&](size_t o) -> unsigned {
        if (o + 4 > tn) return 0;
        return le ? (t[o] | (t[o + 1] << 8) | (t[o + 2] << 16) | ((unsigned)t[o + 3] << 24))
                  : (((unsigned)t[o] << 24) | (t[o + 1] << 16) | (t[o + 2] << 8) | t[o + 3]);
    };
    if (u16(2) != 42) return 0;
    size_t ifd = u32(4);
    unsigned cnt = u16(ifd);
    for (unsigned i = 0; i < cnt; i++) {
        size_t e = ifd + 2 + 12 * (size_t)i;
        if (e + 12 > tn) break;
        if (u16(e) == 0x0112) return (int)u16(e + 8);
    }
    return 0;
}

void set_orientation(nvimgcodecOrientation_t* o, int exif)
{
    int rot = 0, fx = 0, fy = 0;
    switch (exif) {
    case 2: fx = 1; break;
    case 3: rot = 180; break;
    case 4: fy = 1; break;
    case 5: rot = 90; fy = 1; break;
    case 6: rot = 270; break;
    case 7: rot = 270; fy = 1; break;
    case 8: rot = 90; break;
    default: break;
    }
    *o = {NVIMGCODEC_STRUCTURE_TYPE_ORIENTATION, sizeof(nvimgcodecOrientation_t), nullptr, rot, fx, fy};
}

nvimgcodecChromaSubsampling_t to_abi_css(hipjpegChromaSubsampling_t c)
{
    switch (c) {
    case HIPJPEG_CSS_444: return NVIMGCODEC_SAMPLING_444;
    case HIPJPEG_CSS_422: return NVIMGCODEC_SAMPLING_422;
    case HIPJPEG_CSS_420: return NVIMGCODEC_SAMPLING_420;
    case HIPJPEG_CSS_440: return NVIMGCODEC_SAMPLING_440;
    case HIPJPEG_CSS_411: return NVIMGCODEC_SAMPLING_411;
    case HIPJPEG_CSS_410: return NVIMGCODEC_SAMPLING_410;
    case HIPJPEG_CSS_GRAY: return NVIMGCODEC_SAMPLING_GRAY;
    case HIPJPEG_CSS_410V: return NVIMGCODEC_SAMPLING_410V;
    default: return NVIMGCODEC_SAMPLING_UNSUPPORTED;
    }
}

}  // namespace

// ================================================================================================ opaque handle types
struct nvimgcodecExtension {
    nvimgcodecInstance_t instance = nullptr;
    nvimgcodecExtensionDesc_t desc{};
    nvimgcodecExtension_t handle = nullptr;  // what the extension's create() returned
    void* dl = nullptr;
};

struct nvimgcodecDebugMessenger {
    nvimgcodecInstance_t instance = nullptr;
    nvimgcodecDebugMessengerDesc_t desc{};
};

struct nvimgcodecInstance {
    std::mutex m;
    std::multimap<float, const nvimgcodecDecoderDesc_t*> decoders;  // jpeg codec only
    std::multimap<float, const nvimgcodecEncoderDesc_t*> encoders;
    std::vector<nvimgcodecDebugMessenger*> messengers;
    std::vector<std::unique_ptr<nvimgcodecExtension>> owned_extensions;
    std::unique_ptr<nvimgcodecDebugMessenger> default_messenger;
    nvimgcodecFrameworkDesc_t fw{};
    std::string builtin_module;  // path of the extension library loaded from this library's own directory

    static nvimgcodecStatus_t log(void* inst, const nvimgcodecDebugMessageSeverity_t sev, const nvimgcodecDebugMessageCategory_t cat,
                                  const nvimgcodecDebugMessageData_t* data)
    {
        auto* self = static_cast<nvimgcodecInstance*>(inst);
        std::lock_guard<std::mutex> lk(self->m);
        for (auto* dm : self->messengers)
            if ((dm->desc.message_severity & sev) && (dm->desc.message_category & cat) && dm->desc.user_callback)
                dm->desc.user_callback(sev, cat, data, dm->desc.user_data);
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    template <typename Map, typename Desc>
    static nvimgcodecStatus_t do_register(nvimgcodecInstance* self, Map& map, const Desc* desc, float priority)
    {
        if (!desc || !desc->codec) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
        if (strcmp(desc->codec, "jpeg") != 0) return NVIMGCODEC_STATUS_SUCCESS;  // other codecs: accepted, never dispatched to
        std::lock_guard<std::mutex> lk(self->m);
        map.emplace(priority, desc);
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    template <typename Map, typename Desc>
    static nvimgcodecStatus_t do_unregister(nvimgcodecInstance* self, Map& map, const Desc* desc)
    {
        std::lock_guard<std::mutex> lk(self->m);
        for (auto it = map.begin(); it != map.end(); ++it)
            if (it->second == desc) {
                map.erase(it);
                return NVIMGCODEC_STATUS_SUCCESS;
            }
        return NVIMGCODEC_STATUS_SUCCESS;
    }
};

struct nvimgcodecCodeStream {
    nvimgcodecInstance_t instance = nullptr;
    std::unique_ptr<IoStream> io;
    nvimgcodecIoStreamDesc_t io_desc{};
    nvimgcodecCodeStreamDesc_t desc{};
    bool is_output = false;
    bool parsed = false;
    nvimgcodecStatus_t parse_status = NVIMGCODEC_STATUS_SUCCESS;
    nvimgcodecImageInfo_t info{};
    nvimgcodecJpegImageInfo_t jpeg_info{NVIMGCODEC_STRUCTURE_TYPE_JPEG_IMAGE_INFO, sizeof(nvimgcodecJpegImageInfo_t), nullptr,
                                        NVIMGCODEC_JPEG_ENCODING_UNKNOWN};
    bool has_jpeg_info = false;

    // Header parse, cached (reference src/code_stream.cpp:75-98); fills what src/parsers/jpeg.cpp:311-353 fills.
    nvimgcodecStatus_t ensure_parsed()
    {
        if (parsed) return parse_status;
        parsed = true;
        size_t n = 0;
        io->size(&n);
        std::vector<uint8_t> own;
        void* mapped = nullptr;
        io->map(&mapped, 0, n);
        const uint8_t* p = static_cast<const uint8_t*>(mapped);
        if (!p) {
            own.resize(n);
            size_t got = 0;
            io->seek(0, SEEK_SET);
            io->read(&got, own.data(), n);
            io->seek(0, SEEK_SET);
            if (got != n) return parse_status = NVIMGCODEC_STATUS_BAD_CODESTREAM;
            p = own.data();
        }
        FrameInfo f;
        ParseStatus ps = parse_jpeg(p, n, &f, /*headers_only=*/true);
        memset(&info, 0, sizeof info);
        info.struct_type = NVIMGCODEC_STRUCTURE_TYPE_IMAGE_INFO;
        info.struct_size = sizeof info;
        if (ps != kParseOk && f.width == 0) {
            // not a JPEG (or a SOF type we cannot even read dimensions from)
            if (ps == kParseUnsupported && f.sof != 0) {
                // arithmetic / lossless / 12-bit: still a jpeg code stream; dimensions unknown to this minimal parser
                strcpy(info.codec_name, "jpeg");
                jpeg_info.encoding = (nvimgcodecJpegEncoding_t)f.sof;
                has_jpeg_info = true;
                return parse_status = NVIMGCODEC_STATUS_SUCCESS;
            }
            return parse_status = NVIMGCODEC_STATUS_CODESTREAM_UNSUPPORTED;
        }
        strcpy(info.codec_name, "jpeg");
        info.sample_format = f.ncomp > 1 ? NVIMGCODEC_SAMPLEFORMAT_P_RGB : NVIMGCODEC_SAMPLEFORMAT_P_Y;
        info.chroma_subsampling = to_abi_css(classify_subsampling(f));
        info.color_spec = f.ncomp == 1 ? NVIMGCODEC_COLORSPEC_GRAY
                                       : (f.ncomp == 4 ? (f.adobe_transform == 2 ? NVIMGCODEC_COLORSPEC_YCCK : NVIMGCODEC_COLORSPEC_CMYK)
                                                       : NVIMGCODEC_COLORSPEC_SYCC);
        info.num_planes = (uint32_t)f.ncomp;
        for (int c = 0; c < f.ncomp; c++) {
            auto& pi = info.plane_info[c];
            pi.struct_type = NVIMGCODEC_STRUCTURE_TYPE_IMAGE_PLANE_INFO;
            pi.struct_size = sizeof pi;
            pi.width = (uint32_t)f.width;
            pi.height = (uint32_t)f.height;
            pi.num_channels = 1;
            pi.sample_type = f.precision <= 8 ? NVIMGCODEC_SAMPLE_DATA_TYPE_UINT8 : NVIMGCODEC_SAMPLE_DATA_TYPE_UINT16;
            pi.precision = (uint8_t)f.precision;
        }
        // EXIF orientation: walk the APPn segments before SOS
        int exif = 0;
        for (size_t pos = 2; pos + 4 <= n && p[pos] == 0xFF;) {
            int m = p[pos + 1];
            if (m == 0xDA || m == 0xD9) break;
            size_t L = ((size_t)p[pos + 2] << 8) | p[pos + 3];
            if (m == 0xE1 && pos + 2 + L <= n && !exif) exif = exif_orientation_tag(p + pos + 4, L - 2);
            pos += 2 + L;
        }
        set_orientation(&info.orientation, exif);
        info.region.struct_type = NVIMGCODEC_STRUCTURE_TYPE_REGION;
        info.region.struct_size = sizeof info.region;
        jpeg_info.encoding = (nvimgcodecJpegEncoding_t)f.sof;
        has_jpeg_info = true;
        return parse_status = NVIMGCODEC_STATUS_SUCCESS;
    }

    static nvimgcodecStatus_t get_info_thunk(void* inst, nvimgcodecImageInfo_t* out)
    {
        auto* self = static_cast<nvimgcodecCodeStream*>(inst);
        if (!out) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
        if (!self->is_output) {
            nvimgcodecStatus_t st = self->ensure_parsed();
            if (st != NVIMGCODEC_STATUS_SUCCESS) return st;
        }
        void* chain = out->struct_next;  // caller-provided extension structs are preserved and filled
        *out = self->info;
        out->struct_next = chain;
        if (self->has_jpeg_info) {
            struct Head {
                nvimgcodecStructureType_t t;
                size_t s;
                void* n;
            };
            for (Head* h = static_cast<Head*>(chain); h; h = static_cast<Head*>(h->n))
                if (h->t == NVIMGCODEC_STRUCTURE_TYPE_JPEG_IMAGE_INFO) reinterpret_cast<nvimgcodecJpegImageInfo_t*>(h)->encoding = self->jpeg_info.encoding;
        }
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    void finish_setup()
    {
        io_desc = make_io_desc(io.get());
        desc = {NVIMGCODEC_STRUCTURE_TYPE_CODE_STREAM_DESC, sizeof(nvimgcodecCodeStreamDesc_t), nullptr, this, &io_desc, &get_info_thunk};
    }
};

struct nvimgcodecImage {
    nvimgcodecInstance_t instance = nullptr;
    nvimgcodecImageInfo_t info{};
};

// Samples reported more than once (the reference's promise throws logic_error on a double set, src/processing_results.cpp:
// 104-115; here the first result wins and the event is counted so that tests can assert it never happens).
static std::atomic<int> g_double_reports{0};
extern "C" __attribute__((visibility("default"))) int hipjpegTestDoubleReports(void)
{
    const char* e = getenv("HIPJPEG_ENABLE_TEST_HOOKS");  // like the extension's hipjpegTest* entry points: test processes only
    return e && e[0] == '1' ? g_double_reports.load() : -1;
}

struct nvimgcodecFuture {
    explicit nvimgcodecFuture(size_t n) : status(n, NVIMGCODEC_PROCESSING_STATUS_UNKNOWN), remaining((int)n) {}
    void set(size_t i, nvimgcodecProcessingStatus_t s)
    {
        std::lock_guard<std::mutex> lk(m);
        if (status[i] != NVIMGCODEC_PROCESSING_STATUS_UNKNOWN && s != NVIMGCODEC_PROCESSING_STATUS_UNKNOWN) {
            g_double_reports.fetch_add(1);
            return;  // first result wins
        }
        status[i] = s;
        if (--remaining == 0) cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return remaining <= 0; });
    }
    std::mutex m;
    std::condition_variable cv;
    std::vector<nvimgcodecProcessingStatus_t> status;
    int remaining;
    std::shared_ptr<void> keepalive;  // per-call state (sample contexts) owned by the future
};

namespace {

// One plugin instance in the priority chain (reference src/decoder_worker.cpp: a worker owns one plugin decoder and
// forwards what it cannot handle to the next worker).
template <typename Desc, typename Handle>
struct ChainLink {
    const Desc* desc = nullptr;
    Handle handle = nullptr;
    bool create_failed = false;
    std::mutex call_mutex;  // plugin entry points are not assumed re-entrant
};

bool backend_allowed(const nvimgcodecExecutionParams_t& ep, nvimgcodecBackendKind_t kind)
{
    if (ep.num_backends == 0 || !ep.backends) return true;
    for (int i = 0; i < ep.num_backends; i++)
        if (ep.backends[i].kind == kind) return true;
    return false;
}

struct ExecState {
    nvimgcodecExecutionParams_t ep{};
    std::vector<nvimgcodecBackend_t> backends;
    std::unique_ptr<DefaultExecutor> executor;
    std::string options;
    void init(const nvimgcodecExecutionParams_t* user, const char* opts)
    {
        ep = *user;
        ep.struct_next = nullptr;
        if (user->backends && user->num_backends > 0) {
            backends.assign(user->backends, user->backends + user->num_backends);
            ep.backends = backends.data();
        } else {
            ep.backends = nullptr;
            ep.num_backends = 0;
        }
        if (ep.device_id == NVIMGCODEC_DEVICE_CURRENT) {
            int d = 0;
            ep.device_id = hipGetDevice(&d) == hipSuccess ? d : NVIMGCODEC_DEVICE_CPU_ONLY;  // resolved here, reference image_generic_decoder.cpp:61-62
        }
        if (!ep.executor) {
            int n = ep.max_num_cpu_threads > 0 ? ep.max_num_cpu_threads : usable_cpus();
            executor.reset(new DefaultExecutor(ep.device_id, n));
            ep.executor = &executor->desc;
        }
        options = opts ? opts : "";
    }
};

}  // namespace

struct nvimgcodecDecoder {
    nvimgcodecInstance_t instance = nullptr;
    ExecState exec;
    std::vector<std::unique_ptr<ChainLink<nvimgcodecDecoderDesc_t, nvimgcodecDecoder_t>>> chain;
    ~nvimgcodecDecoder()
    {
        for (auto& l : chain)
            if (l->handle) l->desc->destroy(l->handle);
    }
};

struct nvimgcodecEncoder {
    nvimgcodecInstance_t instance = nullptr;
    ExecState exec;
    std::vector<std::unique_ptr<ChainLink<nvimgcodecEncoderDesc_t, nvimgcodecEncoder_t>>> chain;
    ~nvimgcodecEncoder()
    {
        for (auto& l : chain)
            if (l->handle) l->desc->destroy(l->handle);
    }
};

namespace {

// ---------------------------------------------------------------------------------------------- dispatch with fallback
template <typename Codec>
struct Dispatch;  // per-call shared state

struct SampleCtx {
    void* dispatch = nullptr;
    size_t index = 0;
    size_t level = 0;
    nvimgcodecImage* image = nullptr;
    nvimgcodecCodeStream* stream = nullptr;
    nvimgcodecImageDesc_t image_desc{};
    nvimgcodecImageInfo_t effective_info{};  // what the plugin sees (device bounce buffer substituted when needed)
    void* bounce = nullptr;
    size_t bounce_bytes = 0;
    nvimgcodecProcessingStatus_t last_status = NVIMGCODEC_PROCESSING_STATUS_CODEC_UNSUPPORTED;
};

size_t image_bytes(const nvimgcodecImageInfo_t& info)
{
    size_t n = 0;
    for (uint32_t p = 0; p < info.num_planes && p < NVIMGCODEC_MAX_NUM_PLANES; p++) n += info.plane_info[p].row_stride * info.plane_info[p].height;
    return n;
}

template <bool kDecode>
struct DispatchT {
    using CodecHandle = typename std::conditional<kDecode, nvimgcodecDecoder, nvimgcodecEncoder>::type;
    CodecHandle* codec = nullptr;
    nvimgcodecFuture* future = nullptr;
    std::vector<SampleCtx> samples;
    nvimgcodecDecodeParams_t dparams{};
    nvimgcodecEncodeParams_t eparams{};
    nvimgcodecJpegEncodeParams_t jpeg_eparams{};

    static nvimgcodecStatus_t get_info(void* inst, nvimgcodecImageInfo_t* out)
    {
        auto* s = static_cast<SampleCtx*>(inst);
        void* chain = out->struct_next;
        *out = s->effective_info;
        out->struct_next = chain;
        return NVIMGCODEC_STATUS_SUCCESS;
    }
    static nvimgcodecStatus_t image_ready(void* inst, nvimgcodecProcessingStatus_t st)
    {
        auto* s = static_cast<SampleCtx*>(inst);
        static_cast<DispatchT*>(s->dispatch)->on_ready(s, st);
        return NVIMGCODEC_STATUS_SUCCESS;
    }

    void complete(SampleCtx* s, nvimgcodecProcessingStatus_t st)
    {
        if (s->bounce) {
            if (kDecode && st == NVIMGCODEC_PROCESSING_STATUS_SUCCESS) {
                // device bounce buffer -> the user's host buffer, on the image's stream (reference src/work.h:171-186)
                hipStream_t us = (hipStream_t)s->image->info.cuda_stream;
                if (hipMemcpyAsync(s->image->info.buffer, s->bounce, s->bounce_bytes, hipMemcpyDeviceToHost, us) != hipSuccess ||
                    hipStreamSynchronize(us) != hipSuccess)
                    st = NVIMGCODEC_PROCESSING_STATUS_FAIL;
            }
            (void)hipFree(s->bounce);
            s->bounce = nullptr;
        }
        future->set(s->index, st);
    }

    // a plugin reported on a sample: success ends it, anything else moves it one step down the chain
    void on_ready(SampleCtx* s, nvimgcodecProcessingStatus_t st)
    {
        if (st == NVIMGCODEC_PROCESSING_STATUS_SUCCESS) {
            complete(s, st);
            return;
        }
        s->last_status = st;
        std::vector<size_t> one{s->index};
        run_level(s->level + 1, one);
    }

    void run_level(size_t level, const std::vector<size_t>& idxs)
    {
        if (idxs.empty()) return;
        if (level >= codec->chain.size()) {
            for (size_t i : idxs) complete(&samples[i], samples[i].last_status);
            return;
        }
        auto& link = *codec->chain[level];
        std::vector<size_t> accepted, rejected;
        {
            std::lock_guard<std::mutex> lk(link.call_mutex);
            if (!link.handle && !link.create_failed) {
                // plugin objects are created lazily on first use (reference src/decoder_worker.cpp:63-93)
                nvimgcodecStatus_t st = link.desc->create(link.desc->instance, &link.handle, &codec->exec.ep, codec->exec.options.c_str());
                if (st != NVIMGCODEC_STATUS_SUCCESS || !link.handle) {
                    link.handle = nullptr;
                    link.create_failed = true;
                }
            }
        }
        if (!link.handle) {
            run_level(level + 1, idxs);
            return;
        }
        const bool gpu_backend = link.desc->backend_kind != NVIMGCODEC_BACKEND_KIND_CPU_ONLY;
        std::vector<nvimgcodecCodeStreamDesc_t*> cs;
        std::vector<nvimgcodecImageDesc_t*> im;
        for (size_t i : idxs) {
            SampleCtx& s = samples[i];
            s.level = level;
            s.effective_info = s.image->info;
            cs.push_back(&s.stream->desc);
            im.push_back(&s.image_desc);
        }
        std::vector<nvimgcodecProcessingStatus_t> st(idxs.size(), NVIMGCODEC_PROCESSING_STATUS_UNKNOWN);
        {
            std::lock_guard<std::mutex> lk(link.call_mutex);
            nvimgcodecStatus_t rc;
            if constexpr (kDecode)
                rc = link.desc->canDecode(link.handle, st.data(), cs.data(), im.data(), (int)idxs.size(), &dparams);
            else
                rc = link.desc->canEncode(link.handle, st.data(), im.data(), cs.data(), (int)idxs.size(), &eparams);
            if (rc != NVIMGCODEC_STATUS_SUCCESS) std::fill(st.begin(), st.end(), (nvimgcodecProcessingStatus_t)NVIMGCODEC_PROCESSING_STATUS_FAIL);
        }
        for (size_t k = 0; k < idxs.size(); k++) {
            if (st[k] == NVIMGCODEC_PROCESSING_STATUS_SUCCESS) {
                accepted.push_back(idxs[k]);
            } else {
                samples[idxs[k]].last_status = st[k];
                rejected.push_back(idxs[k]);
            }
        }
        if (!accepted.empty()) {
            cs.clear();
            im.clear();
            for (size_t i : accepted) {
                SampleCtx& s = samples[i];
                // GPU backends always see device memory (reference src/work.h:144-169, 192-232)
                if (gpu_backend && s.image->info.buffer_kind == NVIMGCODEC_IMAGE_BUFFER_KIND_STRIDED_HOST && !s.bounce) {
                    s.bounce_bytes = image_bytes(s.image->info);
                    if (hipMalloc(&s.bounce, s.bounce_bytes) == hipSuccess) {
                        if (!kDecode) {
                            hipStream_t us = (hipStream_t)s.image->info.cuda_stream;
                            (void)hipMemcpyAsync(s.bounce, s.image->info.buffer, s.bounce_bytes, hipMemcpyHostToDevice, us);
                            (void)hipStreamSynchronize(us);
                        }
                    } else {
                        s.bounce = nullptr;
                    }
                }
                if (s.bounce && gpu_backend) {
                    s.effective_info.buffer = s.bounce;
                    s.effective_info.buffer_kind = NVIMGCODEC_IMAGE_BUFFER_KIND_STRIDED_DEVICE;
                }
                cs.push_back(&s.stream->desc);
                im.push_back(&s.image_desc);
            }
            nvimgcodecStatus_t rc;
            {
                std::lock_guard<std::mutex> lk(link.call_mutex);
                if constexpr (kDecode)
                    rc = link.desc->decode(link.handle, cs.data(), im.data(), (int)accepted.size(), &dparams);
                else
                    rc = link.desc->encode(link.handle, im.data(), cs.data(), (int)accepted.size(), &eparams);
            }
            (void)rc;  // on a batch-level error the plugin has already reported every sample through imageReady
        }
        run_level(level + 1, rejected);
    }
};

template <bool kDecode, typename CodecHandle>
nvimgcodecStatus_t start_dispatch(CodecHandle* codec, const nvimgcodecCodeStream_t* streams, const nvimgcodecImage_t* images, int n,
                                  const void* params, nvimgcodecFuture_t* future)
{
    if (!codec || !streams || !images || !future || n < 0) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    auto disp = std::make_shared<DispatchT<kDecode>>();
    auto* fut = new nvimgcodecFuture((size_t)n);
    disp->codec = codec;
    disp->future = fut;
    if constexpr (kDecode) {
        disp->dparams = params ? *static_cast<const nvimgcodecDecodeParams_t*>(params)
                               : nvimgcodecDecodeParams_t{NVIMGCODEC_STRUCTURE_TYPE_DECODE_PARAMS, sizeof(nvimgcodecDecodeParams_t), nullptr, 0, 0};
    } else {
        disp->eparams = params ? *static_cast<const nvimgcodecEncodeParams_t*>(params)
                               : nvimgcodecEncodeParams_t{NVIMGCODEC_STRUCTURE_TYPE_ENCODE_PARAMS, sizeof(nvimgcodecEncodeParams_t), nullptr, 90.f, 0.f};
        // the chained JPEG parameters are copied: encoding continues after this call returns
        struct Head {
            nvimgcodecStructureType_t t;
            size_t s;
            void* n;
        };
        void* chain = disp->eparams.struct_next;
        disp->eparams.struct_next = nullptr;
        for (Head* h = static_cast<Head*>(chain); h; h = static_cast<Head*>(h->n))
            if (h->t == NVIMGCODEC_STRUCTURE_TYPE_JPEG_ENCODE_PARAMS) {
                disp->jpeg_eparams = *reinterpret_cast<nvimgcodecJpegEncodeParams_t*>(h);
                disp->jpeg_eparams.struct_next = nullptr;
                disp->eparams.struct_next = &disp->jpeg_eparams;
            }
    }
    disp->samples.resize((size_t)n);
    std::vector<size_t> all;
    for (int i = 0; i < n; i++) {
        SampleCtx& s = disp->samples[(size_t)i];
        s.dispatch = disp.get();
        s.index = (size_t)i;
        s.image = images[i];
        s.stream = streams[i];
        s.image_desc = {NVIMGCODEC_STRUCTURE_TYPE_IMAGE_DESC, sizeof(nvimgcodecImageDesc_t), nullptr, &s, &DispatchT<kDecode>::get_info,
                        &DispatchT<kDecode>::image_ready};
        all.push_back((size_t)i);
    }
    fut->keepalive = disp;
    *future = fut;
    if (n == 0) return NVIMGCODEC_STATUS_SUCCESS;
    disp->run_level(0, all);
    return NVIMGCODEC_STATUS_SUCCESS;
}

template <typename CodecHandle, typename Map>
void build_chain(CodecHandle* codec, Map& registry)
{
    // priority order, lower value first; backends filtered by the user's allow-list (reference test/decoder_worker_test.cpp:110-171)
    for (auto& kv : registry) {
        if (!backend_allowed(codec->exec.ep, kv.second->backend_kind)) continue;
        if (codec->exec.ep.device_id == NVIMGCODEC_DEVICE_CPU_ONLY && kv.second->backend_kind != NVIMGCODEC_BACKEND_KIND_CPU_ONLY) continue;
        using Link = typename std::remove_reference<decltype(*codec->chain[0])>::type;
        std::unique_ptr<Link> l(new Link());
        l->desc = kv.second;
        codec->chain.push_back(std::move(l));
    }
}

}  // namespace

// ================================================================================================ exported C API
extern "C" {

nvimgcodecStatus_t nvimgcodecGetProperties(nvimgcodecProperties_t* p)
{
    if (!p || p->struct_type != NVIMGCODEC_STRUCTURE_TYPE_PROPERTIES) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    p->version = NVIMGCODEC_VER;
    p->ext_api_version = NVIMGCODEC_EXT_API_VER;
    int v = 0;
    (void)hipRuntimeGetVersion(&v);
    p->cudart_version = (uint32_t)v;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecExtensionCreate(nvimgcodecInstance_t instance, nvimgcodecExtension_t* extension, nvimgcodecExtensionDesc_t* ext_desc)
{
    if (!instance || !ext_desc || !ext_desc->create || !ext_desc->destroy) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    // an extension built against a newer extension API than ours cannot be loaded (reference src/plugin_framework.cpp:191-220)
    if (ext_desc->ext_api_version > NVIMGCODEC_EXT_API_VER) return NVIMGCODEC_STATUS_IMPLEMENTATION_UNSUPPORTED;
    std::unique_ptr<nvimgcodecExtension> e(new nvimgcodecExtension());
    e->instance = instance;
    e->desc = *ext_desc;
    nvimgcodecStatus_t st = ext_desc->create(ext_desc->instance, &e->handle, &instance->fw);
    if (st != NVIMGCODEC_STATUS_SUCCESS) return st;
    nvimgcodecExtension* raw = e.get();
    {
        std::lock_guard<std::mutex> lk(instance->m);
        instance->owned_extensions.push_back(std::move(e));
    }
    if (extension) *extension = raw;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecExtensionDestroy(nvimgcodecExtension_t extension)
{
    if (!extension) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    nvimgcodecInstance_t inst = extension->instance;
    std::unique_ptr<nvimgcodecExtension> owned;
    {
        std::lock_guard<std::mutex> lk(inst->m);
        for (auto it = inst->owned_extensions.begin(); it != inst->owned_extensions.end(); ++it)
            if (it->get() == extension) {
                owned = std::move(*it);
                inst->owned_extensions.erase(it);
                break;
            }
    }
    if (!owned) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    nvimgcodecStatus_t st = owned->desc.destroy(owned->handle);
    if (owned->dl) dlclose(owned->dl);
    return st;
}

static void default_messenger_cb_install(nvimgcodecInstance* inst, uint32_t sev, uint32_t cat)
{
    inst->default_messenger.reset(new nvimgcodecDebugMessenger());
    inst->default_messenger->instance = inst;
    inst->default_messenger->desc = {NVIMGCODEC_STRUCTURE_TYPE_DEBUG_MESSENGER_DESC, sizeof(nvimgcodecDebugMessengerDesc_t), nullptr, sev, cat,
                                     [](const nvimgcodecDebugMessageSeverity_t s, const nvimgcodecDebugMessageCategory_t,
                                        const nvimgcodecDebugMessageData_t* d, void*) -> int {
                                         fprintf(stderr, "[nvimgcodec-host][%s][sev 0x%x] %s\n", d->codec_id ? d->codec_id : "core", (unsigned)s,
                                                 d->message ? d->message : "");
                                         return 0;
                                     },
                                     nullptr};
    inst->messengers.push_back(inst->default_messenger.get());
}

nvimgcodecStatus_t nvimgcodecInstanceCreate(nvimgcodecInstance_t* instance, const nvimgcodecInstanceCreateInfo_t* ci)
{
    if (!instance || !ci || ci->struct_type != NVIMGCODEC_STRUCTURE_TYPE_INSTANCE_CREATE_INFO) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    auto* inst = new nvimgcodecInstance();
    int hipv = 0;
    (void)hipRuntimeGetVersion(&hipv);
    inst->fw = {NVIMGCODEC_STRUCTURE_TYPE_FRAMEWORK_DESC,
                sizeof(nvimgcodecFrameworkDesc_t),
                nullptr,
                inst,
                "hipjpeg-host-harness",
                NVIMGCODEC_VER,
                NVIMGCODEC_EXT_API_VER,
                (uint32_t)hipv,
                &nvimgcodecInstance::log,
                [](void* i, const nvimgcodecEncoderDesc_t* d, float p) { auto* s = static_cast<nvimgcodecInstance*>(i); return nvimgcodecInstance::do_register(s, s->encoders, d, p); },
                [](void* i, const nvimgcodecEncoderDesc_t* d) { auto* s = static_cast<nvimgcodecInstance*>(i); return nvimgcodecInstance::do_unregister(s, s->encoders, d); },
                [](void* i, const nvimgcodecDecoderDesc_t* d, float p) { auto* s = static_cast<nvimgcodecInstance*>(i); return nvimgcodecInstance::do_register(s, s->decoders, d, p); },
                [](void* i, const nvimgcodecDecoderDesc_t* d) { auto* s = static_cast<nvimgcodecInstance*>(i); return nvimgcodecInstance::do_unregister(s, s->decoders, d); },
                [](void*, const nvimgcodecParserDesc_t*, float) { return NVIMGCODEC_STATUS_SUCCESS; },  // stream info comes from the built-in JPEG parser
                [](void*, const nvimgcodecParserDesc_t*) { return NVIMGCODEC_STATUS_SUCCESS; }};
    if (ci->create_debug_messenger) {
        if (ci->debug_messenger_desc) {
            inst->default_messenger.reset(new nvimgcodecDebugMessenger());
            inst->default_messenger->instance = inst;
            inst->default_messenger->desc = *ci->debug_messenger_desc;
            inst->messengers.push_back(inst->default_messenger.get());
        } else {
            default_messenger_cb_install(inst, ci->message_severity, ci->message_category);
        }
    }
    if (ci->load_extension_modules) {
        // 1) the extension that ships with this harness: libhipjpeg_ext.so in the directory this library was loaded from, opened like
        //    any other extension module (RTLD_LOCAL: the plugin's symbols stay out of the process's global scope, as in the reference's
        //    loader, src/library_loader.h:33)
        {
            Dl_info self;
            if (dladdr(reinterpret_cast<const void*>(&nvimgcodecInstanceCreate), &self) && self.dli_fname) {
                std::string path(self.dli_fname);
                const size_t slash = path.rfind('/');
                path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/libhipjpeg_ext.so";
                if (void* dl = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL)) {
                    auto entry = reinterpret_cast<nvimgcodecExtensionModuleEntryFunc_t>(dlsym(dl, "nvimgcodecExtensionModuleEntry"));
                    nvimgcodecExtensionDesc_t ed;
                    memset(&ed, 0, sizeof ed);
                    ed.struct_type = NVIMGCODEC_STRUCTURE_TYPE_EXTENSION_DESC;
                    ed.struct_size = sizeof ed;
                    nvimgcodecExtension_t xh = nullptr;
                    if (entry && entry(&ed) == NVIMGCODEC_STATUS_SUCCESS && nvimgcodecExtensionCreate(inst, &xh, &ed) == NVIMGCODEC_STATUS_SUCCESS)
                        inst->builtin_module = path;
                    else
                        dlclose(dl);
                }
            }
        }
        // 2) every regular file of the extension directories (reference src/plugin_framework.cpp:107-117,281-351)
        std::string paths = ci->extension_modules_path ? ci->extension_modules_path : "";
        if (paths.empty())
            if (const char* env = getenv("NVIMGCODEC_EXTENSIONS_PATH")) paths = env;
        size_t start = 0;
        while (start < paths.size()) {
            size_t end = paths.find(':', start);
            if (end == std::string::npos) end = paths.size();
            std::string dir = paths.substr(start, end - start);
            start = end + 1;
            DIR* d = dir.empty() ? nullptr : opendir(dir.c_str());
            if (!d) continue;
            while (dirent* de = readdir(d)) {
                if (de->d_name[0] == '~' || de->d_name[0] == '.') continue;
                std::string path = dir + "/" + de->d_name;
                struct stat sb;
                if (stat(path.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) continue;
                void* dl = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
                if (!dl) continue;
                auto entry = reinterpret_cast<nvimgcodecExtensionModuleEntryFunc_t>(dlsym(dl, "nvimgcodecExtensionModuleEntry"));
                char real_a[4096], real_b[4096];
                const bool same_as_builtin = !inst->builtin_module.empty() && realpath(path.c_str(), real_a) && realpath(inst->builtin_module.c_str(), real_b) &&
                                             strcmp(real_a, real_b) == 0;
                if (!entry || same_as_builtin) {
                    dlclose(dl);
                    continue;
                }
                nvimgcodecExtensionDesc_t xd;
                memset(&xd, 0, sizeof xd);
                xd.struct_type = NVIMGCODEC_STRUCTURE_TYPE_EXTENSION_DESC;
                xd.struct_size = sizeof xd;
                nvimgcodecExtension_t xh = nullptr;
                if (entry(&xd) == NVIMGCODEC_STATUS_SUCCESS && nvimgcodecExtensionCreate(inst, &xh, &xd) == NVIMGCODEC_STATUS_SUCCESS)
                    xh->dl = dl;
                else
                    dlclose(dl);
            }
            closedir(d);
        }
    }
    *instance = inst;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecInstanceDestroy(nvimgcodecInstance_t instance)
{
    if (!instance) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    while (true) {
        nvimgcodecExtension* e = nullptr;
        {
            std::lock_guard<std::mutex> lk(instance->m);
            if (!instance->owned_extensions.empty()) e = instance->owned_extensions.back().get();
        }
        if (!e) break;
        nvimgcodecExtensionDestroy(e);
    }
    delete instance;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecDebugMessengerCreate(nvimgcodecInstance_t instance, nvimgcodecDebugMessenger_t* out, const nvimgcodecDebugMessengerDesc_t* desc)
{
    if (!instance || !out || !desc) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    auto* dm = new nvimgcodecDebugMessenger();
    dm->instance = instance;
    dm->desc = *desc;
    std::lock_guard<std::mutex> lk(instance->m);
    instance->messengers.push_back(dm);
    *out = dm;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecDebugMessengerDestroy(nvimgcodecDebugMessenger_t dm)
{
    if (!dm) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    {
        std::lock_guard<std::mutex> lk(dm->instance->m);
        auto& v = dm->instance->messengers;
        v.erase(std::remove(v.begin(), v.end(), dm), v.end());
    }
    delete dm;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecFutureWaitForAll(nvimgcodecFuture_t f)
{
    if (!f) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    f->wait();
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecFutureDestroy(nvimgcodecFuture_t f)
{
    if (!f) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    f->wait();  // per-sample contexts are referenced by in-flight plugin callbacks until every result arrived
    delete f;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecFutureGetProcessingStatus(nvimgcodecFuture_t f, nvimgcodecProcessingStatus_t* st, size_t* size)
{
    if (!f || !size) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    std::lock_guard<std::mutex> lk(f->m);
    *size = f->status.size();
    if (st) std::copy(f->status.begin(), f->status.end(), st);
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecImageCreate(nvimgcodecInstance_t instance, nvimgcodecImage_t* image, const nvimgcodecImageInfo_t* info)
{
    if (!instance || !image || !info || info->struct_type != NVIMGCODEC_STRUCTURE_TYPE_IMAGE_INFO) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    if (!info->buffer) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    if (info->buffer_kind != NVIMGCODEC_IMAGE_BUFFER_KIND_STRIDED_DEVICE && info->buffer_kind != NVIMGCODEC_IMAGE_BUFFER_KIND_STRIDED_HOST)
        return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    auto* im = new nvimgcodecImage();
    im->instance = instance;
    im->info = *info;
    im->info.struct_next = nullptr;
    *image = im;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecImageDestroy(nvimgcodecImage_t image)
{
    if (!image) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    delete image;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecImageGetImageInfo(nvimgcodecImage_t image, nvimgcodecImageInfo_t* info)
{
    if (!image || !info) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    void* chain = info->struct_next;
    *info = image->info;
    info->struct_next = chain;
    return NVIMGCODEC_STATUS_SUCCESS;
}

static nvimgcodecStatus_t finish_input_stream(nvimgcodecCodeStream* cs, nvimgcodecCodeStream_t* out)
{
    cs->finish_setup();
    // Creation only asks "which parser takes this stream": for JPEG that is the SOI signature (reference
    // src/parsers/jpeg.cpp:130-145).  The marker walk happens on the first GetImageInfo and may still fail there
    // (test/parsers/jpeg_test.cpp Error_GetInfo_NoSOF).
    uint8_t soi[2] = {0, 0};
    size_t got = 0;
    cs->io->seek(0, SEEK_SET);
    cs->io->read(&got, soi, 2);
    cs->io->seek(0, SEEK_SET);
    if (got != 2 || soi[0] != 0xFF || soi[1] != 0xD8) {
        delete cs;
        return NVIMGCODEC_STATUS_CODESTREAM_UNSUPPORTED;
    }
    *out = cs;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecCodeStreamCreateFromFile(nvimgcodecInstance_t instance, nvimgcodecCodeStream_t* out, const char* file_name)
{
    if (!instance || !out || !file_name) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    auto* cs = new nvimgcodecCodeStream();
    cs->instance = instance;
    auto* fs = new FileStream(file_name, "rb");
    cs->io.reset(fs);
    if (!fs->f) {
        delete cs;
        return NVIMGCODEC_STATUS_BAD_CODESTREAM;
    }
    return finish_input_stream(cs, out);
}

nvimgcodecStatus_t nvimgcodecCodeStreamCreateFromHostMem(nvimgcodecInstance_t instance, nvimgcodecCodeStream_t* out, const unsigned char* data,
                                                         size_t length)
{
    if (!instance || !out || !data || length == 0) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    auto* cs = new nvimgcodecCodeStream();
    cs->instance = instance;
    cs->io.reset(new MemInStream(data, length));  // wraps the caller's memory, no copy (reference src/code_stream.cpp:44-64)
    return finish_input_stream(cs, out);
}

static nvimgcodecStatus_t finish_output_stream(nvimgcodecCodeStream* cs, const nvimgcodecImageInfo_t* info, nvimgcodecCodeStream_t* out)
{
    cs->is_output = true;
    cs->info = *info;
    cs->info.struct_next = nullptr;
    struct Head {
        nvimgcodecStructureType_t t;
        size_t s;
        void* n;
    };
    for (const Head* h = static_cast<const Head*>(info->struct_next); h; h = static_cast<const Head*>(h->n))
        if (h->t == NVIMGCODEC_STRUCTURE_TYPE_JPEG_IMAGE_INFO) {
            cs->jpeg_info.encoding = reinterpret_cast<const nvimgcodecJpegImageInfo_t*>(h)->encoding;
            cs->has_jpeg_info = true;
        }
    cs->finish_setup();
    *out = cs;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecCodeStreamCreateToFile(nvimgcodecInstance_t instance, nvimgcodecCodeStream_t* out, const char* file_name,
                                                    const nvimgcodecImageInfo_t* info)
{
    if (!instance || !out || !file_name || !info) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    auto* cs = new nvimgcodecCodeStream();
    cs->instance = instance;
    auto* fs = new FileStream(file_name, "wb");
    cs->io.reset(fs);
    if (!fs->f) {
        delete cs;
        return NVIMGCODEC_STATUS_BAD_CODESTREAM;
    }
    return finish_output_stream(cs, info, out);
}

nvimgcodecStatus_t nvimgcodecCodeStreamCreateToHostMem(nvimgcodecInstance_t instance, nvimgcodecCodeStream_t* out, void* ctx,
                                                       nvimgcodecResizeBufferFunc_t resize, const nvimgcodecImageInfo_t* info)
{
    if (!instance || !out || !resize || !info) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    auto* cs = new nvimgcodecCodeStream();
    cs->instance = instance;
    cs->io.reset(new MemOutStream(ctx, resize));
    return finish_output_stream(cs, info, out);
}

nvimgcodecStatus_t nvimgcodecCodeStreamDestroy(nvimgcodecCodeStream_t cs)
{
    if (!cs) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    delete cs;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecCodeStreamGetImageInfo(nvimgcodecCodeStream_t cs, nvimgcodecImageInfo_t* info)
{
    if (!cs || !info) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    return nvimgcodecCodeStream::get_info_thunk(cs, info);
}

nvimgcodecStatus_t nvimgcodecDecoderCreate(nvimgcodecInstance_t instance, nvimgcodecDecoder_t* decoder, const nvimgcodecExecutionParams_t* ep,
                                           const char* options)
{
    if (!instance || !decoder || !ep || ep->struct_type != NVIMGCODEC_STRUCTURE_TYPE_EXECUTION_PARAMS) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    auto* d = new nvimgcodecDecoder();
    d->instance = instance;
    d->exec.init(ep, options);
    {
        std::lock_guard<std::mutex> lk(instance->m);
        build_chain(d, instance->decoders);
    }
    *decoder = d;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecDecoderDestroy(nvimgcodecDecoder_t decoder)
{
    if (!decoder) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    delete decoder;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecDecoderCanDecode(nvimgcodecDecoder_t decoder, const nvimgcodecCodeStream_t* streams, const nvimgcodecImage_t* images,
                                              int n, const nvimgcodecDecodeParams_t* params, nvimgcodecProcessingStatus_t* status, int force_format)
{
    if (!decoder || !streams || !images || !status || !params) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    // Walk the chain: a sample is decodable if some decoder accepts it; with force_format == 0 a "could decode with other
    // parameters" answer (low bits 0b01) also counts (reference src/image_generic_decoder.cpp:84-132).
    for (int i = 0; i < n; i++) status[i] = NVIMGCODEC_PROCESSING_STATUS_CODEC_UNSUPPORTED;
    for (auto& link : decoder->chain) {
        std::lock_guard<std::mutex> lk(link->call_mutex);
        if (!link->handle && !link->create_failed) {
            if (link->desc->create(link->desc->instance, &link->handle, &decoder->exec.ep, decoder->exec.options.c_str()) != NVIMGCODEC_STATUS_SUCCESS) {
                link->handle = nullptr;
                link->create_failed = true;
            }
        }
        if (!link->handle) continue;
        for (int i = 0; i < n; i++) {
            if (status[i] == NVIMGCODEC_PROCESSING_STATUS_SUCCESS) continue;
            SampleCtx s;
            s.image = images[i];
            s.effective_info = images[i]->info;
            nvimgcodecImageDesc_t idesc{NVIMGCODEC_STRUCTURE_TYPE_IMAGE_DESC, sizeof(nvimgcodecImageDesc_t), nullptr, &s, &DispatchT<true>::get_info,
                                        nullptr};
            nvimgcodecCodeStreamDesc_t* cs = &streams[i]->desc;
            nvimgcodecImageDesc_t* ip = &idesc;
            nvimgcodecProcessingStatus_t st = NVIMGCODEC_PROCESSING_STATUS_UNKNOWN;
            if (link->desc->canDecode(link->handle, &st, &cs, &ip, 1, params) != NVIMGCODEC_STATUS_SUCCESS) continue;
            bool ok = st == NVIMGCODEC_PROCESSING_STATUS_SUCCESS || (!force_format && (st & 0x3) == 0x1);
            if (ok)
                status[i] = NVIMGCODEC_PROCESSING_STATUS_SUCCESS;
            else if (status[i] == NVIMGCODEC_PROCESSING_STATUS_CODEC_UNSUPPORTED)
                status[i] = st;
        }
    }
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecDecoderDecode(nvimgcodecDecoder_t decoder, const nvimgcodecCodeStream_t* streams, const nvimgcodecImage_t* images, int n,
                                           const nvimgcodecDecodeParams_t* params, nvimgcodecFuture_t* future)
{
    return start_dispatch<true>(decoder, streams, images, n, params, future);
}

nvimgcodecStatus_t nvimgcodecEncoderCreate(nvimgcodecInstance_t instance, nvimgcodecEncoder_t* encoder, const nvimgcodecExecutionParams_t* ep,
                                           const char* options)
{
    if (!instance || !encoder || !ep || ep->struct_type != NVIMGCODEC_STRUCTURE_TYPE_EXECUTION_PARAMS) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    auto* e = new nvimgcodecEncoder();
    e->instance = instance;
    e->exec.init(ep, options);
    {
        std::lock_guard<std::mutex> lk(instance->m);
        build_chain(e, instance->encoders);
    }
    *encoder = e;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecEncoderDestroy(nvimgcodecEncoder_t encoder)
{
    if (!encoder) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    delete encoder;
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecEncoderCanEncode(nvimgcodecEncoder_t encoder, const nvimgcodecImage_t* images, const nvimgcodecCodeStream_t* streams,
                                              int n, const nvimgcodecEncodeParams_t* params, nvimgcodecProcessingStatus_t* status, int force_format)
{
    if (!encoder || !streams || !images || !status || !params) return NVIMGCODEC_STATUS_INVALID_PARAMETER;
    for (int i = 0; i < n; i++) status[i] = NVIMGCODEC_PROCESSING_STATUS_CODEC_UNSUPPORTED;
    for (auto& link : encoder->chain) {
        std::lock_guard<std::mutex> lk(link->call_mutex);
        if (!link->handle && !link->create_failed) {
            if (link->desc->create(link->desc->instance, &link->handle, &encoder->exec.ep, encoder->exec.options.c_str()) != NVIMGCODEC_STATUS_SUCCESS) {
                link->handle = nullptr;
                link->create_failed = true;
            }
        }
        if (!link->handle) continue;
        for (int i = 0; i < n; i++) {
            if (status[i] == NVIMGCODEC_PROCESSING_STATUS_SUCCESS) continue;
            SampleCtx s;
            s.image = images[i];
            s.effective_info = images[i]->info;
            nvimgcodecImageDesc_t idesc{NVIMGCODEC_STRUCTURE_TYPE_IMAGE_DESC, sizeof(nvimgcodecImageDesc_t), nullptr, &s, &DispatchT<false>::get_info,
                                        nullptr};
            nvimgcodecCodeStreamDesc_t* cs = &streams[i]->desc;
            nvimgcodecImageDesc_t* ip = &idesc;
            nvimgcodecProcessingStatus_t st = NVIMGCODEC_PROCESSING_STATUS_UNKNOWN;
            if (link->desc->canEncode(link->handle, &st, &ip, &cs, 1, params) != NVIMGCODEC_STATUS_SUCCESS) continue;
            bool ok = st == NVIMGCODEC_PROCESSING_STATUS_SUCCESS || (!force_format && (st & 0x3) == 0x1);
            if (ok)
                status[i] = NVIMGCODEC_PROCESSING_STATUS_SUCCESS;
            else if (status[i] == NVIMGCODEC_PROCESSING_STATUS_CODEC_UNSUPPORTED)
                status[i] = st;
        }
    }
    return NVIMGCODEC_STATUS_SUCCESS;
}

nvimgcodecStatus_t nvimgcodecEncoderEncode(nvimgcodecEncoder_t encoder, const nvimgcodecImage_t* images, const nvimgcodecCodeStream_t* streams, int n,
                                           const nvimgcodecEncodeParams_t* params, nvimgcodecFuture_t* future)
{
    return start_dispatch<false>(encoder, streams, images, n, params, future);
}

}  // extern "C"
