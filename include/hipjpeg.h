/*
 * hipjpeg.h -- C-ABI of the MI355X-native JPEG hot path (libhipjpeg_ext.so).
 *
 * Plain pointers and sizes only; no C++ or torch types.  These are the entry points a host-language binding
 * (ctypes / cgo / JNI ...) uses, and the same library also exports the nvImageCodec plugin entry symbol
 * `nvimgcodecExtensionModuleEntry` (include/nvimgcodec_abi.h) through which an unmodified nvImageCodec loads it.
 *
 * Each group of functions names the reference interface it replaces (paths relative to /root/reference):
 *
 *   hipjpegGetImageInfo          nvjpegJpegStreamParse(Header)       extensions/nvjpeg/cuda_decoder.cpp:503-504,
 *                                / the framework's JPEG parser       src/parsers/jpeg.cpp:202-361
 *   hipjpegEntropyDecodeHost     nvjpegDecodeJpegHost                extensions/nvjpeg/cuda_decoder.cpp:527-530
 *   hipjpegDecodeBatch*          nvjpegDecodeJpegTransferToDevice +  extensions/nvjpeg/cuda_decoder.cpp:544-549
 *                                nvjpegDecodeJpegDevice               (one batched launch instead of one per image)
 *   hipjpegEncodeBatch*          nvjpegEncodeImage/RetrieveBitstream extensions/nvjpeg/cuda_encoder.cpp:362-381
 *
 * All device pointers are ordinary HIP device allocations; `stream` is a hipStream_t passed as void*.
 * Every function returns hipjpegStatus_t (0 = success).  The library never falls back to a CPU pixel path:
 * if no HIP device is usable, the device entry points return HIPJPEG_STATUS_NO_DEVICE.
 */
#ifndef HIPJPEG_H_
#define HIPJPEG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPJPEG_API __attribute__((visibility("default")))

typedef enum {
    HIPJPEG_STATUS_SUCCESS = 0,
    HIPJPEG_STATUS_INVALID_ARGUMENT = 1,
    HIPJPEG_STATUS_BAD_JPEG = 2,         /* not a JPEG or malformed marker segment */
    HIPJPEG_STATUS_UNSUPPORTED = 3,      /* valid JPEG outside this decoder's scope (arithmetic, 12-bit, CMYK, ...) */
    HIPJPEG_STATUS_TRUNCATED = 4,        /* entropy-coded data ends early */
    HIPJPEG_STATUS_CORRUPT = 5,          /* invalid Huffman code / restart marker / coefficient index */
    HIPJPEG_STATUS_ALLOC_FAILED = 6,
    HIPJPEG_STATUS_HIP_ERROR = 7,
    HIPJPEG_STATUS_NO_DEVICE = 8,
    HIPJPEG_STATUS_BUFFER_TOO_SMALL = 9,
    HIPJPEG_STATUS_INTERNAL_ERROR = 10   /* a C++ exception was caught at the C boundary (never propagated to the caller) */
} hipjpegStatus_t;

/* Output pixel layouts; numeric values equal hipjpeg::OutFormat (csrc/device_layout.h). They correspond to
 * nvimgcodecSampleFormat_t I_RGB, I_BGR, P_RGB, P_BGR, P_Y, P_YUV (type_convert.cpp:19-41 in the reference). */
typedef enum {
    HIPJPEG_OUTPUT_RGBI = 0,
    HIPJPEG_OUTPUT_BGRI = 1,
    HIPJPEG_OUTPUT_RGB_PLANAR = 2,
    HIPJPEG_OUTPUT_BGR_PLANAR = 3,
    HIPJPEG_OUTPUT_Y = 4,
    HIPJPEG_OUTPUT_YUV_PLANAR = 5
} hipjpegOutputFormat_t;

typedef enum {
    HIPJPEG_CSS_444 = 0,
    HIPJPEG_CSS_422 = 1,
    HIPJPEG_CSS_420 = 2,
    HIPJPEG_CSS_440 = 3,
    HIPJPEG_CSS_411 = 4,
    HIPJPEG_CSS_410 = 5,
    HIPJPEG_CSS_GRAY = 6,
    HIPJPEG_CSS_410V = 7,
    HIPJPEG_CSS_UNKNOWN = -1
} hipjpegChromaSubsampling_t;

#define HIPJPEG_FLAG_FANCY_UPSAMPLING 1u /* libjpeg do_fancy_upsampling (plugin option fancy_upsampling, default on) */
#define HIPJPEG_FLAG_GPU_HUFFMAN 2u      /* entropy-decode eligible streams on the GPU (sequential Huffman with one interleaved scan, with or
                                            without restart intervals; progressive SOF2 with up to 24 scans); the host then only finds the
                                            scans.  Other streams keep the host entropy stage. */

typedef struct {
    int32_t width, height, num_components;
    int32_t sof_marker;  /* 0xC0 baseline, 0xC1 extended, 0xC2 progressive, other = unsupported type */
    int32_t color_model; /* 0 gray, 1 YCbCr, 2 RGB, 3 CMYK, 4 YCCK */
    int32_t subsampling; /* hipjpegChromaSubsampling_t */
    int32_t restart_interval, num_scans;
    int32_t h[4], v[4];
    int32_t blocks_w[4], blocks_h[4]; /* MCU-padded block grid per component */
    int32_t samp_w[4], samp_h[4];     /* true component size in samples */
    uint64_t coef_bytes;              /* bytes of int16 coefficient storage for the whole image */
} hipjpegImageInfo_t;

typedef struct {
    void* plane[3];     /* device pointers; interleaved formats use plane[0] only */
    uint32_t pitch[3];  /* bytes per row */
} hipjpegOutput_t;

/* Optional per-image geometry: region of interest and EXIF orientation, applied on the device after decoding.
 * The output buffer then holds the region (all zeros = whole image) of the decoded image, brought upright according to
 * `orientation` (EXIF tag 0x0112 values 1..8; 0 or 1 = leave as stored): its size is (x1-x0) x (y1-y0), swapped for
 * orientations 5..8.  Mirrors nvimgcodecImageInfo_t::region (ref include/nvimgcodec.h:446-455; crop semantics of
 * extensions/libjpeg_turbo/jpeg_mem.cpp:206-240: exactly the pixels of the full decode) and nvimgcodecOrientation_t as the
 * nvJPEG plugin applies it (ref extensions/nvjpeg/cuda_decoder.cpp:443-478, type_convert.cpp:43-64).
 * Not available for HIPJPEG_OUTPUT_YUV_PLANAR (subsampled planes): such images report HIPJPEG_STATUS_UNSUPPORTED. */
typedef struct {
    int32_t x0, y0, x1, y1; /* region in stored-image coordinates, end exclusive */
    int32_t orientation;    /* EXIF orientation 1..8 (0 = 1) */
} hipjpegTransform_t;

typedef struct hipjpegHandle* hipjpegHandle_t;

HIPJPEG_API const char* hipjpegStatusString(hipjpegStatus_t status);
HIPJPEG_API int hipjpegVersion(void);
/* ---- test hooks.  Every hipjpegTest* entry point answers only in a process started with HIPJPEG_ENABLE_TEST_HOOKS=1 (read once) and
 * refuses (INVALID_ARGUMENT / -1) anywhere else: a production process cannot have a throw armed inside the library. ---- */
/* Test hook (fault injection): the `countdown`-th passage of the named host-code site from now on throws a C++ exception
 * inside the library, once; site NULL or "" disarms.  Sites: "plan", "entropy_stage", "finalize", "transfer", "launch",
 * "resolve", "marshal".  The boundary must turn it into a status code / per-sample FAIL; tests/test_gpu_plugin.py relies on it. */
HIPJPEG_API hipjpegStatus_t hipjpegTestSetFault(const char* site, int countdown);
/* Test hook: how many times a plugin has reported a sample that had already been reported (must stay 0: exactly one
 * imageReady per sample, reference src/processing_results.cpp:104-115). Counted by this library's host harness. */
HIPJPEG_API int hipjpegTestDoubleReports(void);
/* Test hook: how many images of the handle's last settled batch the GPU entropy stage handed back to the host entropy decoder
 * (damaged streams, and periodic streams whose corrections would have to travel through the image subsequence by subsequence). */
HIPJPEG_API int32_t hipjpegTestHostFallbacks(hipjpegHandle_t handle);
/* Test hook: work units of the handle's current batch per kernel -- plane_units[0]: the plane IDCT kernel (K1); luma_units[3]: the fused
 * luma kernel (K2) by layout (0 generic, 1 the everyday interleaved kernel, 2 the everyday planar kernel; csrc/decode_kernels.h). */
HIPJPEG_API hipjpegStatus_t hipjpegTestKernelFlavours(hipjpegHandle_t handle, int32_t plane_units[1], int32_t luma_units[3]);
/* Test hook: how many of those work units went to the FUSED kernel builds (blocks Huffman-decoded inside the pixel kernels,
 * HIPJPEG_FUSED_DECODE=1); negative without a handle. */
HIPJPEG_API int32_t hipjpegTestFusedUnits(hipjpegHandle_t handle);

/* Test hook (host only): the parser's per-chunk counts of the bytes that byte-stuffing removal drops from scan `scan_index` (16,384-byte
 * chunks of the entropy-coded segment; the GPU entropy stage's compact kernel works from them).  Returns the number of chunks, or a
 * negative value when the file does not parse / has no such scan; at most `capacity` counts are written. */
HIPJPEG_API int32_t hipjpegTestScanChunkDrops(const uint8_t* data, size_t length, int scan_index, uint32_t* drops, int32_t capacity);

/* ---- host-only entry points (usable without a GPU) ---- */
HIPJPEG_API hipjpegStatus_t hipjpegGetImageInfo(const uint8_t* data, size_t length, hipjpegImageInfo_t* info);

/* Huffman-decode every scan into quantized coefficient blocks in the device layout (block raster order per
 * component, 64 int16 per block stored column-major).  comp_offsets[c] receives the int16 offset of component c
 * inside `coef`; qtables[c*64 .. ] the (column-major) quantisation table of component c. */
HIPJPEG_API hipjpegStatus_t hipjpegEntropyDecodeHost(const uint8_t* data, size_t length, int16_t* coef, size_t coef_capacity_bytes,
                                                     uint64_t comp_offsets[4], uint16_t qtables[256]);

/* The host entropy stage's zero-run-compressed output (what crosses PCIe for host-decoded pictures since round 3; csrc/entropy_decode.h):
 * per-block offset tables (uint32 per block of each component's MCU-padded raster grid, the tables back to back; table_offsets[c] = index of
 * component c's first entry; 0 = block never coded) followed by the records [n : u8][DC : i16 LE] n x {position : u8, value : i16 LE}, position =
 * index into the device-layout block (column-major).  UNSUPPORTED for frames the format does not cover (progressive, several scans): those
 * stay dense.  capacity_bytes >= 200 bytes per block is always enough. */
HIPJPEG_API hipjpegStatus_t hipjpegEntropyDecodeHostSparse(const uint8_t* data, size_t length, uint8_t* stream, size_t capacity_bytes, size_t* stream_bytes,
                                                           uint64_t table_offsets[4]);

/* The GPU entropy decoder's algorithm (self-synchronizing subsequence decoding, csrc/huffman_gpu_core.h) executed on the
 * host, lane by lane, with the very code the kernels run: lets the algorithm be verified without a GPU.  Same output
 * layout as hipjpegEntropyDecodeHost; returns HIPJPEG_STATUS_UNSUPPORTED for streams the GPU entropy path does not take
 * (sequential streams in several scans, arithmetic coding, progressive scripts beyond the walker's limits). */
HIPJPEG_API hipjpegStatus_t hipjpegEntropyDecodeGpuAlgorithmHost(const uint8_t* data, size_t length, int16_t* coef,
                                                                 size_t coef_capacity_bytes, uint64_t comp_offsets[4],
                                                                 int32_t* sync_passes);

/* ---- device pipeline ---- */
/* num_host_threads: CPU threads for the entropy stage (0 = hardware concurrency). */
HIPJPEG_API hipjpegStatus_t hipjpegCreate(hipjpegHandle_t* handle, int device_id, int num_host_threads);
HIPJPEG_API hipjpegStatus_t hipjpegDestroy(hipjpegHandle_t handle);

/* One call = host entropy stage (thread pool) + H2D staging + batched device stage, asynchronous on `stream`
 * (the call returns after the last kernel is enqueued; per-image host failures are reported in `statuses`). */
HIPJPEG_API hipjpegStatus_t hipjpegDecodeBatch(hipjpegHandle_t handle, const uint8_t* const* data, const size_t* lengths, int batch_size,
                                               const hipjpegOutput_t* outputs, hipjpegOutputFormat_t format, unsigned flags,
                                               hipjpegStatus_t* statuses, void* stream);

/* Geometry for the NEXT batch handed to hipjpegDecodeBatch / Host / Submit: `transforms` = batch_size entries (copied), or
 * NULL to go back to plain decoding.  Consumed by that one batch. */
HIPJPEG_API hipjpegStatus_t hipjpegDecodeBatchSetTransforms(hipjpegHandle_t handle, const hipjpegTransform_t* transforms, int batch_size);

/* The three phases separately (what hipjpegDecodeBatch does internally); used by bench.py to time the device
 * stage with the coefficient blocks already resident in HBM. */
HIPJPEG_API hipjpegStatus_t hipjpegDecodeBatchHost(hipjpegHandle_t handle, const uint8_t* const* data, const size_t* lengths, int batch_size,
                                                   const hipjpegOutput_t* outputs, hipjpegOutputFormat_t format, unsigned flags,
                                                   hipjpegStatus_t* statuses);
HIPJPEG_API hipjpegStatus_t hipjpegDecodeBatchTransfer(hipjpegHandle_t handle, void* stream);
HIPJPEG_API hipjpegStatus_t hipjpegDecodeBatchDevice(hipjpegHandle_t handle, void* stream);
/* One kernel family of the device stage at a time (0 = idct_plane, 1 = luma_color, 2 = generic_color, 3 = GPU entropy stage
 * followed by the blocking read-back of its verdicts, 4 = geometry pass, 6 = GPU entropy stage enqueued only: its verdicts
 * are settled by the next hipjpegDecodeBatchGetStatuses, 7 = idct_plane and luma_color alternating over slices of
 * HIPJPEG_PIXEL_CHUNK images -- a measurement aid), so a caller can bracket each with events.
 * hipjpegDecodeBatchDevice == entropy (if any image uses it), then 0, 1, 2, 4. */
HIPJPEG_API hipjpegStatus_t hipjpegDecodeBatchDeviceKernel(hipjpegHandle_t handle, int which, void* stream);
/* Pipelined submission.  Submit = host stage + H2D copy (on an internal copy stream) + every kernel on `stream`, without
 * waiting for anything on the device; at most three batches (hipjpegSetPipelineDepth: up to eight) may be in flight (the handle's staging pages).  Wait = block
 * until the OLDEST submitted batch has finished and return its final per-image statuses.  The host stage of batch n+1 and
 * its H2D copy overlap the kernels of batch n.  Outputs and `data` of a submitted batch must stay valid until its Wait. */
HIPJPEG_API hipjpegStatus_t hipjpegDecodeBatchSubmit(hipjpegHandle_t handle, const uint8_t* const* data, const size_t* lengths, int batch_size,
                                                     const hipjpegOutput_t* outputs, hipjpegOutputFormat_t format, unsigned flags, void* stream);
HIPJPEG_API hipjpegStatus_t hipjpegDecodeBatchWait(hipjpegHandle_t handle, hipjpegStatus_t* statuses, int batch_size);
/* Zero-copy input.  When an image's bitstream lies in page-locked host memory (hipHostMalloc / hipHostRegister, a pinned torch tensor) and
 * takes the GPU entropy stage, the copy engine reads the scan's bytes from THAT memory -- no staging copy on the host (one pass over host
 * DRAM per byte instead of three; what matters when eight ranks share a host).  Nothing to call: the library asks the runtime about every
 * input pointer; HIPJPEG_NO_ZERO_COPY=1 in the environment switches it off.  The caller keeps the memory valid until the batch has been
 * waited for (as for every Submit).  Returns how many images of the handle's current batch went that way (after Transfer / Submit). */
HIPJPEG_API int32_t hipjpegDecodeBatchZeroCopyImages(hipjpegHandle_t handle);
/* What the current batch's transfer puts on PCIe (descriptors, tables, bitstreams of GPU-decoded pictures, coefficients of host-decoded ones)
 * and how many host-decoded pictures went as zero-run-compressed streams rather than dense blocks (HIPJPEG_DENSE_STAGING=1 switches that off). */
HIPJPEG_API hipjpegStatus_t hipjpegDecodeBatchTransferStats(hipjpegHandle_t handle, uint64_t* h2d_bytes, int32_t* sparse_images);

/* With HIPJPEG_FLAG_GPU_HUFFMAN: only images of MORE than `pixels` pixels (width x height) take the GPU entropy stage, smaller ones the
 * host Huffman decoder -- nvJPEG's switch between its HYBRID and GPU_HYBRID backends (plugin option hybrid_huffman_threshold,
 * reference extensions/nvjpeg/cuda_decoder.cpp:188-209, 512-521).  0 (default) = every eligible stream on the GPU. */
HIPJPEG_API hipjpegStatus_t hipjpegSetHybridHuffmanThreshold(hipjpegHandle_t handle, uint64_t pixels);

/* How many batches Submit may have in flight (staging pages in use): 1..8, default 3; only while nothing is in flight.
 * Batches of progressive images keep a small part of the chip busy for a long time (one wave per scan), so their throughput
 * grows with the depth; each page in flight runs its entropy stage on a stream of its own, and the HIP runtime must be
 * allowed as many hardware queues (environment GPU_MAX_HW_QUEUES, default 4, read when the runtime starts). */
HIPJPEG_API hipjpegStatus_t hipjpegSetPipelineDepth(hipjpegHandle_t handle, int depth);
/* Final per-image statuses of the current batch (after the device stage they include what the GPU entropy stage found;
 * blocks until that stage has reported). */
HIPJPEG_API hipjpegStatus_t hipjpegDecodeBatchGetStatuses(hipjpegHandle_t handle, hipjpegStatus_t* statuses, int batch_size);
/* GPU entropy stage statistics: images that used it, kernel launches the last synchronisation needed, destuffed bytes. */
HIPJPEG_API hipjpegStatus_t hipjpegDecodeBatchEntropyStats(hipjpegHandle_t handle, int32_t* gpu_images, int32_t* sync_launches,
                                                           uint64_t* stream_bytes);
/* Launch statistics of the prepared batch: workgroups per kernel (idct_plane, luma_color, generic). */
HIPJPEG_API hipjpegStatus_t hipjpegDecodeBatchStats(hipjpegHandle_t handle, int32_t num_units[3], uint64_t* coef_bytes,
                                                    uint64_t* output_bytes);

/* ---- encode: RGB -> baseline JPEG (replaces nvjpegEncodeImage + nvjpegEncodeRetrieveBitstream,
 *      reference extensions/nvjpeg/cuda_encoder.cpp:336-388) ---- */
typedef struct {
    const void* plane[3]; /* device pointers; interleaved / gray input uses plane[0] only */
    uint32_t pitch[3];
    int32_t width, height;
} hipjpegEncodeInput_t;

typedef struct {
    int32_t quality;           /* 1..100, libjpeg quality scaling of the Annex-K tables */
    int32_t subsampling;       /* hipjpegChromaSubsampling_t of the OUTPUT stream (GRAY = single component) */
    int32_t input_format;      /* HIPJPEG_OUTPUT_RGBI / BGRI / RGB_PLANAR / BGR_PLANAR / Y (gray plane) / YUV_PLANAR (Y, Cb, Cr planes
                                  already in the stream's sampling: plane c holds ceil(width / hs_c) x ceil(height / vs_c) samples) */
    int32_t restart_interval;  /* MCUs per restart interval, 0 = none */
    int32_t optimized_huffman; /* 0 = Annex-K tables, 1 = per-image optimal tables (two-pass) */
    int32_t progressive;       /* 0 = baseline sequential (SOF0); 1 = progressive (SOF2): libjpeg's jpeg_simple_progression scan script
                                  with per-scan optimal tables, what nvimgcodecJpegImageInfo_t::encoding = PROGRESSIVE_DCT_HUFFMAN asks
                                  for (reference extensions/nvjpeg/cuda_encoder.cpp:339-346) */
} hipjpegEncodeParams_t;

/* Device stage only: colour conversion + downsampling + FDCT + quantization for the whole batch (asynchronous). */
HIPJPEG_API hipjpegStatus_t hipjpegEncodeBatchDevice(hipjpegHandle_t handle, const hipjpegEncodeInput_t* inputs,
                                                     const hipjpegEncodeParams_t* params, int batch_size, hipjpegStatus_t* statuses, void* stream);
/* Re-launch the kernel of the prepared batch (bench.py times this). */
HIPJPEG_API hipjpegStatus_t hipjpegEncodeBatchRelaunch(hipjpegHandle_t handle, void* stream);
/* D2H of the quantized coefficients, then Huffman coding + marker writing on the host thread pool (blocking). */
HIPJPEG_API hipjpegStatus_t hipjpegEncodeBatchHost(hipjpegHandle_t handle, hipjpegStatus_t* statuses);
/* Entropy stage with a choice: flags = HIPJPEG_FLAG_GPU_HUFFMAN codes every baseline image without restart markers on the GPU
 * (Annex-K tables, or optimized ones: histograms on the device, jpeg_gen_optimal_table on the host, second pass on the device; then
 * lengths, prefix sums, bit packing, byte stuffing, file assembly -- only finished JPEG files cross PCIe); progressive output and restart
 * intervals (which the reference's encode parameters do not have), and everything when flags = 0, go through the host coder as in
 * hipjpegEncodeBatchHost.  Blocking. */
HIPJPEG_API hipjpegStatus_t hipjpegEncodeBatchEntropy(hipjpegHandle_t handle, unsigned flags, hipjpegStatus_t* statuses);
/* Both of the above. */
HIPJPEG_API hipjpegStatus_t hipjpegEncodeBatch(hipjpegHandle_t handle, const hipjpegEncodeInput_t* inputs, const hipjpegEncodeParams_t* params,
                                               int batch_size, hipjpegStatus_t* statuses, void* stream);
/* Pipelined encoding: Submit queues a whole batch (forward kernel + entropy stage per `flags` as in
 * hipjpegEncodeBatchEntropy + copy of the files to host memory) and returns at once; it runs on an internal stream, ordered
 * behind the work already queued on `stream` (the producer of the pixels).  Wait blocks until the OLDEST submitted batch is
 * complete, returns its statuses and makes it the batch hipjpegEncodeGetBitstream talks about.  At most three batches in
 * flight (three pages, used round robin); the bitstreams of a waited batch stay valid until a Submit takes its page again:
 * the third Submit after the one that queued it.  The PCIe-bound file output of one batch overlaps the kernels of the
 * others. */
HIPJPEG_API hipjpegStatus_t hipjpegEncodeBatchSubmit(hipjpegHandle_t handle, const hipjpegEncodeInput_t* inputs, const hipjpegEncodeParams_t* params,
                                                     int batch_size, unsigned flags, void* stream);
HIPJPEG_API hipjpegStatus_t hipjpegEncodeBatchWait(hipjpegHandle_t handle, hipjpegStatus_t* statuses, int batch_size);
/* Bitstream of image i of the last encoded (or waited-for) batch; valid until the next encode call on this handle. */
HIPJPEG_API hipjpegStatus_t hipjpegEncodeGetBitstream(hipjpegHandle_t handle, int index, const uint8_t** data, size_t* length);
/* Quantized coefficients of (image, component) after hipjpegEncodeBatchHost: zigzag-ordered int16[64] blocks over the
 * MCU-padded grid (only the real_w x real_h area is defined).  For tests and for callers with their own entropy coder. */
HIPJPEG_API hipjpegStatus_t hipjpegEncodeGetCoefficients(hipjpegHandle_t handle, int index, int component, const int16_t** coef,
                                                         int32_t grid[4] /* blocks_w, blocks_h, real_w, real_h */);
HIPJPEG_API hipjpegStatus_t hipjpegEncodeBatchStats(hipjpegHandle_t handle, int32_t* num_units, uint64_t* pixel_bytes, uint64_t* coef_bytes);
/* How many images of the handle's last entropy stage the GPU entropy coder took (Annex-K or optimized tables, no restart intervals, baseline);
 * the others were coded by the host coder. */
HIPJPEG_API int32_t hipjpegEncodeBatchGpuEntropyImages(hipjpegHandle_t handle);
/* Host-only: entropy-code given coefficient grids (zigzag order, MCU-padded grids as above) into a JFIF file.
 * Returns HIPJPEG_STATUS_BUFFER_TOO_SMALL with *length = needed size if capacity is insufficient. */
HIPJPEG_API hipjpegStatus_t hipjpegEncodeFromCoefficientsHost(int32_t width, int32_t height, const hipjpegEncodeParams_t* params,
                                                              const int16_t* const coef[3], uint8_t* out, size_t capacity, size_t* length);

#ifdef __cplusplus
}
#endif
#endif /* HIPJPEG_H_ */
