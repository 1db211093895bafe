// hipimtrans -- batched JPEG transcoder / decode benchmark over the public nvimgcodec C API, for the MI355X build.
//
// The counterpart of the reference's sample application (example/nvimtrans/main.cpp:561-690): read a batch of files, parse,
// nvimgcodecDecoderDecode into device buffers, nvimgcodecEncoderEncode into files, with wall-clock timers around every stage
// and the same closing report ("Avg decoding speed (in images per sec)" ...).  Everything goes through the function tables of
// include/nvimgcodec_abi.h, i.e. through the priority chain into the hipjpeg_decoder / hipjpeg_encoder plugins -- this is
// the API route of DESIGN.md, measured without any Python in the way.
//
//   hipimtrans -i <file|dir> [-o <dir>] [-b batch] [-w warmup batches] [-r repeats] [-q quality] [-s 444|422|420|gray]
//              [-d device] [-t cpu threads] [-p batches in flight (decode only)] [--skip_encode] [--options "<plugin options>"] [-v]
//              [--jpeg_encoding baseline_dct|progressive_dct] [--optimized_huffman true|false]
//              [--devices a,b,...] [--checksums file]
// --devices a,b,... (decode only): ONE process drives several devices -- a decoder instance per entry (nvimgcodecDecoderCreate with that
// device_id: the reference keys its worker pools by device in the same way, src/default_executor.cpp:45-58), a host thread and a queue
// per entry; the input list is partitioned over the queues by greedy longest-processing-time on (MCU-padded coefficient bytes +
// file bytes), visited in decreasing size like the reference's generic decoder sorts a batch (src/image_generic_decoder.cpp:134-178);
// no device ever talks to another (no collective).  An id may appear twice: two queues on one card (how the mode is tested on a
// one-GPU box).  --checksums writes "<file name> <FNV-1a of the decoded RGB bytes>" per input for a checker to compare.
// -p N > 1 uses what the API offers for throughput: nvimgcodecDecoderDecode returns a future as soon as the batch is scheduled
// (include/nvimgcodec_abi.h; reference nvimgcodec.h:1455-1459), so the caller submits batch n+1 before it waits for batch n.
#include <dirent.h>
#include <hip/hip_runtime_api.h>
#include <sys/stat.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/nvimgcodec_abi.h"

namespace {

double wtime()
{
    timespec tp;
    clock_gettime(CLOCK_MONOTONIC, &tp);
    return tp.tv_nsec * 1e-9 + (double)tp.tv_sec;
}

#define CHECK_API(call)                                                                   \
    do {                                                                                  \
        nvimgcodecStatus_t _s = (call);                                                   \
        if (_s != NVIMGCODEC_STATUS_SUCCESS) {                                            \
            fprintf(stderr, "%s failed with status %d (%s:%d)\n", #call, (int)_s, __FILE__, __LINE__); \
            return EXIT_FAILURE;                                                          \
        }                                                                                 \
    } while (0)
#define CHECK_HIP(call)                                                                   \
    do {                                                                                  \
        hipError_t _e = (call);                                                           \
        if (_e != hipSuccess) {                                                           \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return EXIT_FAILURE;                                                          \
        }                                                                                 \
    } while (0)

struct Params {
    std::string input, output, options, checksums;
    std::vector<int> devices;
    int batch = 16, warmup = 1, repeats = 1, quality = 90, device = 0, threads = 0, verbose = 0, in_flight = 1;
    std::string subsampling = "420";
    bool progressive = false, optimized_huffman = false;  // nvimtrans --jpeg_encoding / --optimized_huffman (command_line_params.h:195-207)
    bool skip_encode = false;
};

bool is_dir(const std::string& p)
{
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

std::vector<std::string> list_inputs(const std::string& p)
{
    std::vector<std::string> out;
    if (!is_dir(p)) {
        out.push_back(p);
        return out;
    }
    if (DIR* d = opendir(p.c_str())) {
        while (dirent* e = readdir(d)) {
            std::string n = e->d_name;
            if (n.size() > 4 && (n.substr(n.size() - 4) == ".jpg" || n.substr(n.size() - 5) == ".jpeg")) out.push_back(p + "/" + n);
        }
        closedir(d);
    }
    std::sort(out.begin(), out.end());
    return out;
}

bool read_file(const std::string& path, std::vector<unsigned char>* data)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    data->resize((size_t)n);
    const bool ok = fread(data->data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

uint64_t fnv1a(const unsigned char* p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 1099511628211ull;
    }
    return h;
}

// ---- one process, several devices: a queue, a host thread and a decoder instance per --devices entry
struct QueueResult {
    size_t images = 0, failed = 0;
    double seconds = 0, bytes_in = 0;
    std::string error;
};

void run_device_queue(nvimgcodecInstance_t instance, const Params& p, int device, const std::vector<size_t>& mine,
                      const std::vector<std::vector<unsigned char>>& files, std::vector<uint64_t>* sums, QueueResult* res)
{
    auto fail = [&](const std::string& what) { res->error = what; };
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice");
    nvimgcodecExecutionParams_t ep{};
    ep.struct_type = NVIMGCODEC_STRUCTURE_TYPE_EXECUTION_PARAMS;
    ep.struct_size = sizeof ep;
    ep.device_id = device;
    ep.max_num_cpu_threads = p.threads;
    nvimgcodecDecoder_t decoder = nullptr;
    // A queue that keeps several batches in flight pipelines whole batches itself: the decoder's own cutting of a batch into pieces (for callers
    // that wait for every call) only adds pieces that queue for pages -- measured with two queues on one card, three batches in flight each:
    // 11-16 k images/s with the adaptive pieces, 64 k with whole batches.  A pipeline_chunks the user gave wins.
    std::string options = p.options;
    if (p.in_flight > 1 && options.find("pipeline_chunks") == std::string::npos) options += (options.empty() ? "" : " ") + std::string("hipjpeg_decoder:pipeline_chunks=1");
    if (nvimgcodecDecoderCreate(instance, &decoder, &ep, options.c_str()) != NVIMGCODEC_STATUS_SUCCESS) return fail("nvimgcodecDecoderCreate");
    nvimgcodecDecodeParams_t dparams{NVIMGCODEC_STRUCTURE_TYPE_DECODE_PARAMS, sizeof(nvimgcodecDecodeParams_t), nullptr, 1, 0};
    struct Slot {
        std::vector<nvimgcodecCodeStream_t> streams;
        std::vector<nvimgcodecImage_t> images;
        std::vector<void*> buffers;
        std::vector<size_t> bytes, sizes, index;
        nvimgcodecFuture_t future = nullptr;
        int n = 0;
    };
    std::vector<Slot> slots((size_t)p.in_flight);
    for (auto& sl : slots) {
        sl.buffers.assign((size_t)p.batch, nullptr);
        sl.bytes.assign((size_t)p.batch, 0);
    }
    std::vector<unsigned char> host;
    int head = 0, tail = 0, pending = 0;
    auto retire = [&]() -> bool {
        Slot& sl = slots[(size_t)tail];
        if (nvimgcodecFutureWaitForAll(sl.future) != NVIMGCODEC_STATUS_SUCCESS) return false;
        size_t count = 0;
        nvimgcodecFutureGetProcessingStatus(sl.future, nullptr, &count);
        std::vector<nvimgcodecProcessingStatus_t> st(count);
        nvimgcodecFutureGetProcessingStatus(sl.future, st.data(), &count);
        for (size_t i = 0; i < count; i++) {
            if (st[i] != NVIMGCODEC_PROCESSING_STATUS_SUCCESS) {
                res->failed++;
                continue;
            }
            if (sums) {  // outside what a throughput run measures: asked for by a checker
                host.resize(sl.sizes[i]);
                if (hipMemcpy(host.data(), sl.buffers[i], sl.sizes[i], hipMemcpyDeviceToHost) != hipSuccess) return false;
                (*sums)[sl.index[i]] = fnv1a(host.data(), host.size());
            }
        }
        nvimgcodecFutureDestroy(sl.future);
        for (auto cs : sl.streams) nvimgcodecCodeStreamDestroy(cs);
        for (auto im : sl.images) nvimgcodecImageDestroy(im);
        res->images += (size_t)sl.n;
        tail = (tail + 1) % p.in_flight;
        pending--;
        return true;
    };
    // -w warm-up passes over this queue's share first (pages and output buffers are sized on first use), then the clock starts
    double t0 = wtime();
    for (int rep = -p.warmup; rep < p.repeats && res->error.empty(); rep++) {
        if (rep == 0) {
            while (pending)
                if (!retire()) return fail("waiting for a batch");
            (void)hipDeviceSynchronize();
            res->images = 0;
            res->failed = 0;
            res->bytes_in = 0;
            t0 = wtime();
        }
        for (size_t cursor = 0; cursor < mine.size() && res->error.empty();) {
            if (pending == p.in_flight && !retire()) return fail("waiting for a batch");
            Slot& sl = slots[(size_t)head];
            sl.n = (int)std::min<size_t>((size_t)p.batch, mine.size() - cursor);
            sl.streams.assign((size_t)sl.n, nullptr);
            sl.images.assign((size_t)sl.n, nullptr);
            sl.sizes.assign((size_t)sl.n, 0);
            sl.index.assign((size_t)sl.n, 0);
            for (int i = 0; i < sl.n; i++) {
                const size_t fi = mine[cursor + (size_t)i];
                const std::vector<unsigned char>& file = files[fi];
                res->bytes_in += (double)file.size();
                if (nvimgcodecCodeStreamCreateFromHostMem(instance, &sl.streams[(size_t)i], file.data(), file.size()) != NVIMGCODEC_STATUS_SUCCESS)
                    return fail("nvimgcodecCodeStreamCreateFromHostMem");
                nvimgcodecImageInfo_t info{};
                info.struct_type = NVIMGCODEC_STRUCTURE_TYPE_IMAGE_INFO;
                info.struct_size = sizeof info;
                if (nvimgcodecCodeStreamGetImageInfo(sl.streams[(size_t)i], &info) != NVIMGCODEC_STATUS_SUCCESS) return fail("nvimgcodecCodeStreamGetImageInfo");
                const uint32_t w = info.plane_info[0].width, h = info.plane_info[0].height;
                info.sample_format = NVIMGCODEC_SAMPLEFORMAT_I_RGB;
                info.color_spec = NVIMGCODEC_COLORSPEC_SRGB;
                info.chroma_subsampling = NVIMGCODEC_SAMPLING_NONE;
                info.num_planes = 1;
                info.plane_info[0].num_channels = 3;
                info.plane_info[0].row_stride = (size_t)w * 3;
                info.plane_info[0].sample_type = NVIMGCODEC_SAMPLE_DATA_TYPE_UINT8;
                info.buffer_size = (size_t)w * 3 * h;
                info.buffer_kind = NVIMGCODEC_IMAGE_BUFFER_KIND_STRIDED_DEVICE;
                if (sl.bytes[(size_t)i] < info.buffer_size) {
                    if (sl.buffers[(size_t)i]) (void)hipFree(sl.buffers[(size_t)i]);
                    if (hipMalloc(&sl.buffers[(size_t)i], info.buffer_size) != hipSuccess) return fail("hipMalloc");
                    sl.bytes[(size_t)i] = info.buffer_size;
                }
                info.buffer = sl.buffers[(size_t)i];
                sl.sizes[(size_t)i] = info.buffer_size;
                sl.index[(size_t)i] = fi;
                if (nvimgcodecImageCreate(instance, &sl.images[(size_t)i], &info) != NVIMGCODEC_STATUS_SUCCESS) return fail("nvimgcodecImageCreate");
            }
            if (nvimgcodecDecoderDecode(decoder, sl.streams.data(), sl.images.data(), sl.n, &dparams, &sl.future) != NVIMGCODEC_STATUS_SUCCESS)
                return fail("nvimgcodecDecoderDecode");
            cursor += (size_t)sl.n;
            head = (head + 1) % p.in_flight;
            pending++;
        }
    }
    while (pending)
        if (!retire()) return fail("waiting for a batch");
    (void)hipDeviceSynchronize();
    res->seconds = wtime() - t0;
    for (auto& sl : slots)
        for (void* b : sl.buffers)
            if (b) (void)hipFree(b);
    nvimgcodecDecoderDestroy(decoder);
}

int run_multi_device(nvimgcodecInstance_t instance, const Params& p, const std::vector<std::string>& names)
{
    std::vector<std::vector<unsigned char>> files(names.size());
    std::vector<uint64_t> cost(names.size(), 0);
    for (size_t i = 0; i < names.size(); i++) {
        if (!read_file(names[i], &files[i])) {
            fprintf(stderr, "cannot read %s\n", names[i].c_str());
            return EXIT_FAILURE;
        }
        // work estimate from the header alone: MCU-padded coefficient bytes (what the device stage touches) + file bytes (what the
        // entropy stage walks) -- nvimagecodec_amd/sharding.py image_cost
        nvimgcodecCodeStream_t cs = nullptr;
        nvimgcodecImageInfo_t info{};
        info.struct_type = NVIMGCODEC_STRUCTURE_TYPE_IMAGE_INFO;
        info.struct_size = sizeof info;
        cost[i] = files[i].size();
        if (nvimgcodecCodeStreamCreateFromHostMem(instance, &cs, files[i].data(), files[i].size()) == NVIMGCODEC_STATUS_SUCCESS) {
            if (nvimgcodecCodeStreamGetImageInfo(cs, &info) == NVIMGCODEC_STATUS_SUCCESS) {
                const uint64_t w = info.plane_info[0].width, h = info.plane_info[0].height;
                const uint64_t chroma = info.chroma_subsampling == NVIMGCODEC_SAMPLING_GRAY  ? 0
                                        : info.chroma_subsampling == NVIMGCODEC_SAMPLING_444 ? 2 * w * h
                                        : info.chroma_subsampling == NVIMGCODEC_SAMPLING_420 ? w * h / 2
                                                                                             : w * h;
                cost[i] += 2 * (w * h + chroma);
            }
            nvimgcodecCodeStreamDestroy(cs);
        }
    }
    // greedy longest-processing-time partition, items in (cost descending, index ascending) order, ties to the lowest queue
    const size_t nq = p.devices.size();
    std::vector<size_t> order(names.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
    std::vector<std::vector<size_t>> queues(nq);
    std::vector<uint64_t> load(nq, 0);
    for (size_t i : order) {
        size_t best = 0;
        for (size_t q = 1; q < nq; q++)
            if (load[q] < load[best]) best = q;
        queues[best].push_back(i);
        load[best] += cost[i];
    }
    std::vector<uint64_t> sums(names.size(), 0);
    std::vector<QueueResult> results(nq);
    std::vector<std::thread> threads;
    const double t0 = wtime();
    for (size_t q = 0; q < nq; q++)
        threads.emplace_back(run_device_queue, instance, std::cref(p), p.devices[q], std::cref(queues[q]), std::cref(files),
                             p.checksums.empty() ? nullptr : &sums, &results[q]);
    for (auto& t : threads) t.join();
    (void)t0;
    double t = 0;  // the job's time is the slowest queue's (each queue's clock starts after its warm-up passes)
    for (size_t q = 0; q < nq; q++) t = std::max(t, results[q].seconds);
    size_t images = 0, failed = 0;
    double bytes_in = 0;
    for (size_t q = 0; q < nq; q++) {
        if (!results[q].error.empty()) {
            fprintf(stderr, "queue %zu (device %d): %s failed\n", q, p.devices[q], results[q].error.c_str());
            return EXIT_FAILURE;
        }
        printf("queue %zu on device %d: %zu images (%zu failed) in %f s = %f images per sec, %.1f MB/s of host bitstream bytes\n", q, p.devices[q],
               results[q].images, results[q].failed, results[q].seconds, results[q].images / results[q].seconds, results[q].bytes_in / results[q].seconds / 1e6);
        images += results[q].images;
        failed += results[q].failed;
        bytes_in += results[q].bytes_in;
    }
    printf("\nTotal images: %zu (failed: %zu) over %zu device queues, batch size %d, %d batches in flight per queue\n", images, failed, nq, p.batch, p.in_flight);
    printf("Total decoding time (parsing included, files preloaded): %f\n", t);
    printf("Avg decoding speed  (in images per sec): %f\n", images / t);
    printf("Host bitstream bytes read per sec (all queues): %.1f MB/s\n", bytes_in / t / 1e6);
    if (!p.checksums.empty()) {
        FILE* f = fopen(p.checksums.c_str(), "w");
        if (!f) return EXIT_FAILURE;
        for (size_t i = 0; i < names.size(); i++) {
            const size_t slash = names[i].rfind('/');
            fprintf(f, "%s %016llx\n", names[i].substr(slash == std::string::npos ? 0 : slash + 1).c_str(), (unsigned long long)sums[i]);
        }
        fclose(f);
    }
    return failed ? EXIT_FAILURE : EXIT_SUCCESS;
}

}  // namespace

int main(int argc, char** argv)
{
    Params p;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "-i") p.input = next();
        else if (a == "-o") p.output = next();
        else if (a == "-b") p.batch = atoi(next());
        else if (a == "-w") p.warmup = atoi(next());
        else if (a == "-r") p.repeats = atoi(next());
        else if (a == "-q") p.quality = atoi(next());
        else if (a == "-s") p.subsampling = next();
        else if (a == "-d") p.device = atoi(next());
        else if (a == "-t") p.threads = atoi(next());
        else if (a == "-p") p.in_flight = std::max(1, std::min(6, atoi(next())));
        else if (a == "--options") p.options = next();
        else if (a == "--skip_encode") p.skip_encode = true;
        else if (a == "--jpeg_encoding") p.progressive = std::string(next()) == "progressive_dct";
        else if (a == "--optimized_huffman") p.optimized_huffman = std::string(next()) == "true";
        else if (a == "--devices") {
            std::string list = next();
            for (size_t b = 0; b < list.size();) {
                size_t e = list.find(',', b);
                if (e == std::string::npos) e = list.size();
                p.devices.push_back(atoi(list.substr(b, e - b).c_str()));
                b = e + 1;
            }
        } else if (a == "--checksums") p.checksums = next();
        else if (a == "-v") p.verbose++;
        else {
            fprintf(stderr, "usage: %s -i <file|dir> [-o dir] [-b batch] [-w warmup] [-r repeats] [-q quality] [-s 444|422|420|gray] [-d device] "
                            "[-t threads] [-p batches in flight] [--skip_encode] [--jpeg_encoding baseline_dct|progressive_dct] [--optimized_huffman true|false] "
                            "[--options str] [--devices a,b,...] [--checksums file] [-v]\n", argv[0]);
            return EXIT_FAILURE;
        }
    }
    if (p.input.empty() || p.batch < 1) {
        fprintf(stderr, "an input (-i) and a positive batch size are needed\n");
        return EXIT_FAILURE;
    }
    if (p.output.empty()) p.skip_encode = true;
    std::vector<std::string> names = list_inputs(p.input);
    if (names.empty()) {
        fprintf(stderr, "no .jpg files under %s\n", p.input.c_str());
        return EXIT_FAILURE;
    }
    CHECK_HIP(hipSetDevice(p.device));

    nvimgcodecInstance_t instance = nullptr;
    nvimgcodecInstanceCreateInfo_t ci{};
    ci.struct_type = NVIMGCODEC_STRUCTURE_TYPE_INSTANCE_CREATE_INFO;
    ci.struct_size = sizeof ci;
    ci.load_builtin_modules = 1;
    ci.load_extension_modules = 1;
    ci.create_debug_messenger = p.verbose > 0;
    ci.message_severity = NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_ERROR | NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_FATAL |
                          (p.verbose > 1 ? NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_WARNING : 0);
    ci.message_category = NVIMGCODEC_DEBUG_MESSAGE_CATEGORY_ALL;
    CHECK_API(nvimgcodecInstanceCreate(&instance, &ci));

    if (!p.devices.empty()) {
        const int rc = run_multi_device(instance, p, names);
        nvimgcodecInstanceDestroy(instance);
        return rc;
    }

    nvimgcodecExecutionParams_t ep{};
    ep.struct_type = NVIMGCODEC_STRUCTURE_TYPE_EXECUTION_PARAMS;
    ep.struct_size = sizeof ep;
    ep.device_id = p.device;
    ep.max_num_cpu_threads = p.threads;
    nvimgcodecDecoder_t decoder = nullptr;
    nvimgcodecEncoder_t encoder = nullptr;
    CHECK_API(nvimgcodecDecoderCreate(instance, &decoder, &ep, p.options.c_str()));
    if (!p.skip_encode) CHECK_API(nvimgcodecEncoderCreate(instance, &encoder, &ep, p.options.c_str()));

    nvimgcodecDecodeParams_t dparams{NVIMGCODEC_STRUCTURE_TYPE_DECODE_PARAMS, sizeof(nvimgcodecDecodeParams_t), nullptr, 1, 0};
    nvimgcodecJpegEncodeParams_t jparams{NVIMGCODEC_STRUCTURE_TYPE_JPEG_ENCODE_PARAMS, sizeof(nvimgcodecJpegEncodeParams_t), nullptr, p.optimized_huffman ? 1 : 0};
    nvimgcodecJpegImageInfo_t jinfo{NVIMGCODEC_STRUCTURE_TYPE_JPEG_IMAGE_INFO, sizeof(nvimgcodecJpegImageInfo_t), nullptr,
                                    p.progressive ? NVIMGCODEC_JPEG_ENCODING_PROGRESSIVE_DCT_HUFFMAN : NVIMGCODEC_JPEG_ENCODING_BASELINE_DCT};
    nvimgcodecEncodeParams_t eparams{NVIMGCODEC_STRUCTURE_TYPE_ENCODE_PARAMS, sizeof(nvimgcodecEncodeParams_t), &jparams, (float)p.quality, 50.f};
    const nvimgcodecChromaSubsampling_t css = p.subsampling == "444"   ? NVIMGCODEC_SAMPLING_444
                                              : p.subsampling == "422" ? NVIMGCODEC_SAMPLING_422
                                              : p.subsampling == "gray" ? NVIMGCODEC_SAMPLING_GRAY
                                                                        : NVIMGCODEC_SAMPLING_420;

    if (p.in_flight > 1 && p.skip_encode) {
        // ---- decode only, several batches in flight: slot k holds the files, code streams, images, output buffers and future
        // of one batch; a slot is reused after its future has been waited for
        struct Slot {
            std::vector<std::vector<unsigned char>> files;
            std::vector<nvimgcodecCodeStream_t> streams;
            std::vector<nvimgcodecImage_t> images;
            std::vector<void*> buffers;
            std::vector<size_t> bytes;
            nvimgcodecFuture_t future = nullptr;
            int n = 0;
        };
        std::vector<Slot> slots(p.in_flight);
        for (auto& sl : slots) {
            sl.buffers.assign(p.batch, nullptr);
            sl.bytes.assign(p.batch, 0);
        }
        // the files are read once, before the clock starts: this mode measures the decode route, not the file system
        std::vector<std::vector<unsigned char>> preloaded(names.size());
        for (size_t i = 0; i < names.size(); i++)
            if (!read_file(names[i], &preloaded[i])) {
                fprintf(stderr, "cannot read %s\n", names[i].c_str());
                return EXIT_FAILURE;
            }
        const size_t total = names.size() * (size_t)p.repeats, warm_images = (size_t)p.warmup * p.batch;
        size_t submitted = 0, done = 0, failed = 0, cursor = 0;
        int head = 0, tail = 0, pending = 0;
        double t_begin = wtime();
        bool timing = warm_images == 0;
        auto retire = [&]() -> int {
            Slot& sl = slots[tail];
            CHECK_API(nvimgcodecFutureWaitForAll(sl.future));
            size_t count = 0;
            nvimgcodecFutureGetProcessingStatus(sl.future, nullptr, &count);
            std::vector<nvimgcodecProcessingStatus_t> st(count);
            nvimgcodecFutureGetProcessingStatus(sl.future, st.data(), &count);
            for (auto v : st) failed += v != NVIMGCODEC_PROCESSING_STATUS_SUCCESS;
            nvimgcodecFutureDestroy(sl.future);
            for (auto cs : sl.streams) nvimgcodecCodeStreamDestroy(cs);
            for (auto im : sl.images) nvimgcodecImageDestroy(im);
            done += (size_t)sl.n;
            tail = (tail + 1) % p.in_flight;
            pending--;
            return EXIT_SUCCESS;
        };
        while (submitted < total + warm_images) {
            if (pending == p.in_flight && retire() != EXIT_SUCCESS) return EXIT_FAILURE;
            if (!timing && done >= warm_images) {  // the warm-up batches are through: the clock starts here
                timing = true;
                t_begin = wtime();
            }
            Slot& sl = slots[head];
            sl.n = (int)std::min<size_t>((size_t)p.batch, total + warm_images - submitted);
            sl.streams.assign(sl.n, nullptr);
            sl.images.assign(sl.n, nullptr);
            for (int i = 0; i < sl.n; i++) {
                const std::vector<unsigned char>& file = preloaded[(cursor + i) % names.size()];
                CHECK_API(nvimgcodecCodeStreamCreateFromHostMem(instance, &sl.streams[i], file.data(), file.size()));
                nvimgcodecImageInfo_t info{};
                info.struct_type = NVIMGCODEC_STRUCTURE_TYPE_IMAGE_INFO;
                info.struct_size = sizeof info;
                CHECK_API(nvimgcodecCodeStreamGetImageInfo(sl.streams[i], &info));
                const uint32_t w = info.plane_info[0].width, h = info.plane_info[0].height;
                info.sample_format = NVIMGCODEC_SAMPLEFORMAT_I_RGB;
                info.color_spec = NVIMGCODEC_COLORSPEC_SRGB;
                info.chroma_subsampling = NVIMGCODEC_SAMPLING_NONE;
                info.num_planes = 1;
                info.plane_info[0].num_channels = 3;
                info.plane_info[0].row_stride = (size_t)w * 3;
                info.plane_info[0].sample_type = NVIMGCODEC_SAMPLE_DATA_TYPE_UINT8;
                info.buffer_size = (size_t)w * 3 * h;
                info.buffer_kind = NVIMGCODEC_IMAGE_BUFFER_KIND_STRIDED_DEVICE;
                if (sl.bytes[i] < info.buffer_size) {
                    if (sl.buffers[i]) CHECK_HIP(hipFree(sl.buffers[i]));
                    CHECK_HIP(hipMalloc(&sl.buffers[i], info.buffer_size));
                    sl.bytes[i] = info.buffer_size;
                }
                info.buffer = sl.buffers[i];
                CHECK_API(nvimgcodecImageCreate(instance, &sl.images[i], &info));
            }
            CHECK_API(nvimgcodecDecoderDecode(decoder, sl.streams.data(), sl.images.data(), sl.n, &dparams, &sl.future));
            cursor += (size_t)sl.n;
            submitted += (size_t)sl.n;
            head = (head + 1) % p.in_flight;
            pending++;
        }
        while (pending)
            if (retire() != EXIT_SUCCESS) return EXIT_FAILURE;
        CHECK_HIP(hipDeviceSynchronize());
        const double t = wtime() - t_begin;
        printf("\nTotal images: %zu (failed: %zu), batch size %d, %d batches in flight\n", total, failed, p.batch, p.in_flight);
        printf("Total decoding time (parsing included, files preloaded): %f\n", t);
        printf("Avg decoding speed  (in images per sec): %f\n", total / t);
        for (auto& sl : slots)
            for (void* b : sl.buffers)
                if (b) (void)hipFree(b);
        nvimgcodecDecoderDestroy(decoder);
        nvimgcodecInstanceDestroy(instance);
        return failed ? EXIT_FAILURE : EXIT_SUCCESS;
    }

    const size_t total_images = names.size() * (size_t)p.repeats;
    std::vector<std::vector<unsigned char>> file_data(p.batch);
    std::vector<void*> buffers(p.batch, nullptr);
    std::vector<size_t> buffer_bytes(p.batch, 0);
    double t_read = 0, t_parse = 0, t_decode = 0, t_encode = 0;
    size_t processed = 0, failed = 0, cursor = 0;
    int warm = 0;
    const double t_start = wtime();
    double t_timed_start = t_start;
    while (processed < total_images) {
        const int n = (int)std::min<size_t>((size_t)p.batch, total_images - processed);
        // ---- read
        double t0 = wtime();
        std::vector<std::string> current(n);
        for (int i = 0; i < n; i++) {
            current[i] = names[(cursor + i) % names.size()];
            if (!read_file(current[i], &file_data[i])) {
                fprintf(stderr, "cannot read %s\n", current[i].c_str());
                return EXIT_FAILURE;
            }
        }
        const double reading = wtime() - t0;
        // ---- parse: code streams, output images (interleaved RGB u8 on the device, row_stride = width * 3)
        t0 = wtime();
        std::vector<nvimgcodecCodeStream_t> in_streams(n, nullptr);
        std::vector<nvimgcodecImage_t> images(n, nullptr);
        std::vector<nvimgcodecImageInfo_t> infos(n);
        for (int i = 0; i < n; i++) {
            CHECK_API(nvimgcodecCodeStreamCreateFromHostMem(instance, &in_streams[i], file_data[i].data(), file_data[i].size()));
            nvimgcodecImageInfo_t info{};
            info.struct_type = NVIMGCODEC_STRUCTURE_TYPE_IMAGE_INFO;
            info.struct_size = sizeof info;
            CHECK_API(nvimgcodecCodeStreamGetImageInfo(in_streams[i], &info));
            const uint32_t w = info.plane_info[0].width, h = info.plane_info[0].height;
            info.sample_format = NVIMGCODEC_SAMPLEFORMAT_I_RGB;
            info.color_spec = NVIMGCODEC_COLORSPEC_SRGB;
            info.chroma_subsampling = NVIMGCODEC_SAMPLING_NONE;
            info.num_planes = 1;
            info.plane_info[0].num_channels = 3;
            info.plane_info[0].row_stride = (size_t)w * 3;
            info.plane_info[0].sample_type = NVIMGCODEC_SAMPLE_DATA_TYPE_UINT8;
            info.buffer_size = (size_t)w * 3 * h;
            info.buffer_kind = NVIMGCODEC_IMAGE_BUFFER_KIND_STRIDED_DEVICE;
            if (buffer_bytes[i] < info.buffer_size) {
                if (buffers[i]) CHECK_HIP(hipFree(buffers[i]));
                CHECK_HIP(hipMalloc(&buffers[i], info.buffer_size));
                buffer_bytes[i] = info.buffer_size;
            }
            info.buffer = buffers[i];
            info.cuda_stream = nullptr;
            infos[i] = info;
            CHECK_API(nvimgcodecImageCreate(instance, &images[i], &info));
        }
        const double parsing = wtime() - t0;
        // ---- decode
        t0 = wtime();
        nvimgcodecFuture_t future = nullptr;
        CHECK_API(nvimgcodecDecoderDecode(decoder, in_streams.data(), images.data(), n, &dparams, &future));
        CHECK_API(nvimgcodecFutureWaitForAll(future));
        CHECK_HIP(hipDeviceSynchronize());
        const double decoding = wtime() - t0;
        size_t count = 0;
        nvimgcodecFutureGetProcessingStatus(future, nullptr, &count);
        std::vector<nvimgcodecProcessingStatus_t> status(count);
        nvimgcodecFutureGetProcessingStatus(future, status.data(), &count);
        nvimgcodecFutureDestroy(future);
        std::vector<int> good;
        for (int i = 0; i < n; i++) {
            if (status[i] == NVIMGCODEC_PROCESSING_STATUS_SUCCESS)
                good.push_back(i);
            else {
                failed++;
                fprintf(stderr, "Error: something went wrong during decoding image %s (status 0x%x), it will not be encoded\n", current[i].c_str(),
                        (unsigned)status[i]);
            }
        }
        // ---- encode
        double encoding = 0;
        if (!p.skip_encode && !good.empty()) {
            std::vector<nvimgcodecCodeStream_t> out_streams;
            std::vector<nvimgcodecImage_t> enc_images;
            for (int i : good) {
                nvimgcodecImageInfo_t out_info = infos[i];
                strcpy(out_info.codec_name, "jpeg");
                out_info.chroma_subsampling = css;
                out_info.struct_next = &jinfo;  // nvimtrans main.cpp:138
                std::string base = current[i].substr(current[i].find_last_of('/') + 1);
                std::string path = p.output + "/" + base;
                nvimgcodecCodeStream_t cs = nullptr;
                CHECK_API(nvimgcodecCodeStreamCreateToFile(instance, &cs, path.c_str(), &out_info));
                out_streams.push_back(cs);
                enc_images.push_back(images[i]);
            }
            t0 = wtime();
            nvimgcodecFuture_t ef = nullptr;
            CHECK_API(nvimgcodecEncoderEncode(encoder, enc_images.data(), out_streams.data(), (int)enc_images.size(), &eparams, &ef));
            CHECK_API(nvimgcodecFutureWaitForAll(ef));
            CHECK_HIP(hipDeviceSynchronize());
            encoding = wtime() - t0;
            size_t ec = 0;
            nvimgcodecFutureGetProcessingStatus(ef, nullptr, &ec);
            std::vector<nvimgcodecProcessingStatus_t> es(ec);
            nvimgcodecFutureGetProcessingStatus(ef, es.data(), &ec);
            for (size_t k = 0; k < ec; k++)
                if (es[k] != NVIMGCODEC_PROCESSING_STATUS_SUCCESS) {
                    failed++;
                    fprintf(stderr, "Error: something went wrong during encoding image #%zu, it will not be saved\n", k);
                }
            nvimgcodecFutureDestroy(ef);
            for (auto cs : out_streams) nvimgcodecCodeStreamDestroy(cs);
        }
        for (auto cs : in_streams) nvimgcodecCodeStreamDestroy(cs);
        for (auto im : images) nvimgcodecImageDestroy(im);
        if (p.verbose) fprintf(stderr, "batch of %d: read %.2f ms, parse %.2f ms, decode %.2f ms, encode %.2f ms\n", n, reading * 1e3, parsing * 1e3,
                               decoding * 1e3, encoding * 1e3);
        if (warm < p.warmup) {
            warm++;  // the batch is processed again once the warm-up is over
            t_timed_start = wtime();
            continue;
        }
        cursor += (size_t)n;
        processed += (size_t)n;
        t_read += reading;
        t_parse += parsing;
        t_decode += decoding;
        t_encode += encoding;
        putchar('.');
        fflush(stdout);
    }
    const double total = wtime() - t_timed_start;
    for (void* b : buffers)
        if (b) (void)hipFree(b);
    if (decoder) nvimgcodecDecoderDestroy(decoder);
    if (encoder) nvimgcodecEncoderDestroy(encoder);
    nvimgcodecInstanceDestroy(instance);

    const double nb = (double)((total_images + p.batch - 1) / p.batch);
    auto report = [&](const char* what, double t) {
        printf("Total %s time: %f\n", what, t);
        printf("Avg %s time per image: %f\n", what, t / total_images);
        printf("Avg %s speed  (in images per sec): %f\n", what, t > 0 ? total_images / t : 0.0);
        printf("Avg %s time per batch: %f\n\n", what, t / nb);
    };
    printf("\nTotal images: %zu (failed: %zu), batch size %d\n", total_images, failed, p.batch);
    report(p.skip_encode ? "processing" : "transcoding", total);
    report("reading", t_read);
    report("parsing", t_parse);
    report("decoding", t_decode);
    if (!p.skip_encode) report("encoding", t_encode);
    return failed ? EXIT_FAILURE : EXIT_SUCCESS;
}
