/*
 * nvimgcodec_abi.h -- clean-room, layout-compatible restatement of the nvImageCodec
 * extension/plugin C-ABI (reference: include/nvimgcodec.h, v0.2.0 snapshot) for a
 * ROCm/HIP build.  No CUDA headers are needed: the one CUDA type that crosses the
 * boundary (cudaStream_t, reference include/nvimgcodec.h:29,496) is a pointer, and on
 * this platform it is a HIP stream handle.
 *
 * Every type, enumerator, field order and function-table slot below mirrors the
 * reference so that a plugin built against this header loads into a genuine
 * nvImageCodec build unchanged (and vice versa).  Layout is pinned by the
 * static_asserts at the end of the file (offsets measured in SURVEY.md Appendix B).
 *
 * Reference line numbers are given as "ref:NNN" (= /root/reference/include/nvimgcodec.h:NNN).
 */
#ifndef NVIMGCODEC_ABI_H_
#define NVIMGCODEC_ABI_H_

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#ifndef NVIMGCODECAPI
#define NVIMGCODECAPI __attribute__((visibility("default")))
#endif

/* ext API version 0.2.0 encoded as major*1000 + minor*100 + patch (ref nvimgcodec_version.h.in:35-39) */
#define NVIMGCODEC_VER_MAJOR 0
#define NVIMGCODEC_VER_MINOR 2
#define NVIMGCODEC_VER_PATCH 0
#define NVIMGCODEC_MAKE_VERSION(major, minor, patch) ((major)*1000 + (minor)*100 + (patch))
#define NVIMGCODEC_VER NVIMGCODEC_MAKE_VERSION(NVIMGCODEC_VER_MAJOR, NVIMGCODEC_VER_MINOR, NVIMGCODEC_VER_PATCH)
#define NVIMGCODEC_EXT_API_VER_MAJOR 0
#define NVIMGCODEC_EXT_API_VER_MINOR 2
#define NVIMGCODEC_EXT_API_VER_PATCH 0
#define NVIMGCODEC_EXT_API_VER \
    NVIMGCODEC_MAKE_VERSION(NVIMGCODEC_EXT_API_VER_MAJOR, NVIMGCODEC_EXT_API_VER_MINOR, NVIMGCODEC_EXT_API_VER_PATCH)

#ifdef __cplusplus
extern "C" {
#endif

/* The stream handle that rides in nvimgcodecImageInfo_t::cuda_stream. Pointer-sized on both platforms. */
struct ihipStream_t;
#ifndef NVIMGCODEC_ABI_NO_STREAM_TYPEDEF
typedef struct ihipStream_t* cudaStream_t;
#endif

/* ref:48-53 */
#define NVIMGCODEC_MAX_CODEC_NAME_SIZE 256
#define NVIMGCODEC_DEVICE_CURRENT -1
#define NVIMGCODEC_DEVICE_CPU_ONLY -99999
#define NVIMGCODEC_MAX_NUM_DIM 5
#define NVIMGCODEC_MAX_NUM_PLANES 32
#define NVIMGCODEC_JPEG2K_MAXRES 33

/* Opaque handles (ref:58-110) */
typedef struct nvimgcodecInstance* nvimgcodecInstance_t;
typedef struct nvimgcodecImage* nvimgcodecImage_t;
typedef struct nvimgcodecCodeStream* nvimgcodecCodeStream_t;
typedef struct nvimgcodecParser* nvimgcodecParser_t;
typedef struct nvimgcodecEncoder* nvimgcodecEncoder_t;
typedef struct nvimgcodecDecoder* nvimgcodecDecoder_t;
typedef struct nvimgcodecDebugMessenger* nvimgcodecDebugMessenger_t;
typedef struct nvimgcodecExtension* nvimgcodecExtension_t;
typedef struct nvimgcodecFuture* nvimgcodecFuture_t;

/* ref:121-152 -- the values are positional (0..26) */
typedef enum {
    NVIMGCODEC_STRUCTURE_TYPE_PROPERTIES = 0,
    NVIMGCODEC_STRUCTURE_TYPE_INSTANCE_CREATE_INFO = 1,
    NVIMGCODEC_STRUCTURE_TYPE_DEVICE_ALLOCATOR = 2,
    NVIMGCODEC_STRUCTURE_TYPE_PINNED_ALLOCATOR = 3,
    NVIMGCODEC_STRUCTURE_TYPE_DECODE_PARAMS = 4,
    NVIMGCODEC_STRUCTURE_TYPE_ENCODE_PARAMS = 5,
    NVIMGCODEC_STRUCTURE_TYPE_ORIENTATION = 6,
    NVIMGCODEC_STRUCTURE_TYPE_REGION = 7,
    NVIMGCODEC_STRUCTURE_TYPE_IMAGE_INFO = 8,
    NVIMGCODEC_STRUCTURE_TYPE_IMAGE_PLANE_INFO = 9,
    NVIMGCODEC_STRUCTURE_TYPE_JPEG_IMAGE_INFO = 10,
    NVIMGCODEC_STRUCTURE_TYPE_JPEG_ENCODE_PARAMS = 11,
    NVIMGCODEC_STRUCTURE_TYPE_JPEG2K_ENCODE_PARAMS = 12,
    NVIMGCODEC_STRUCTURE_TYPE_BACKEND = 13,
    NVIMGCODEC_STRUCTURE_TYPE_IO_STREAM_DESC = 14,
    NVIMGCODEC_STRUCTURE_TYPE_FRAMEWORK_DESC = 15,
    NVIMGCODEC_STRUCTURE_TYPE_DECODER_DESC = 16,
    NVIMGCODEC_STRUCTURE_TYPE_ENCODER_DESC = 17,
    NVIMGCODEC_STRUCTURE_TYPE_PARSER_DESC = 18,
    NVIMGCODEC_STRUCTURE_TYPE_IMAGE_DESC = 19,
    NVIMGCODEC_STRUCTURE_TYPE_CODE_STREAM_DESC = 20,
    NVIMGCODEC_STRUCTURE_TYPE_DEBUG_MESSENGER_DESC = 21,
    NVIMGCODEC_STRUCTURE_TYPE_DEBUG_MESSAGE_DATA = 22,
    NVIMGCODEC_STRUCTURE_TYPE_EXTENSION_DESC = 23,
    NVIMGCODEC_STRUCTURE_TYPE_EXECUTOR_DESC = 24,
    NVIMGCODEC_STRUCTURE_TYPE_BACKEND_PARAMS = 25,
    NVIMGCODEC_STRUCTURE_TYPE_EXECUTION_PARAMS = 26,
    NVIMGCODEC_STRUCTURE_TYPE_ENUM_FORCE_INT = INT32_MAX
} nvimgcodecStructureType_t;

/* Every ABI struct opens with these three members (24 bytes on LP64). */
#define NVIMGCODEC_STRUCT_HEAD          \
    nvimgcodecStructureType_t struct_type; \
    size_t struct_size;                    \
    void* struct_next

/* ref:158-167 */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    uint32_t version;
    uint32_t ext_api_version;
    uint32_t cudart_version; /* here: HIP runtime version */
} nvimgcodecProperties_t;

/* Allocator hooks, ref:182-302.  Return 0 on success. */
typedef int (*nvimgcodecDeviceMalloc_t)(void* ctx, void** ptr, size_t size, cudaStream_t stream);
typedef int (*nvimgcodecDeviceFree_t)(void* ctx, void* ptr, size_t size, cudaStream_t stream);
typedef int (*nvimgcodecPinnedMalloc_t)(void* ctx, void** ptr, size_t size, cudaStream_t stream);
typedef int (*nvimgcodecPinnedFree_t)(void* ctx, void* ptr, size_t size, cudaStream_t stream);

typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    nvimgcodecDeviceMalloc_t device_malloc;
    nvimgcodecDeviceFree_t device_free;
    void* device_ctx;
    size_t device_mem_padding;
} nvimgcodecDeviceAllocator_t;

typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    nvimgcodecPinnedMalloc_t pinned_malloc;
    nvimgcodecPinnedFree_t pinned_free;
    void* pinned_ctx;
    size_t pinned_mem_padding;
} nvimgcodecPinnedAllocator_t;

/* ref:307-332 */
typedef enum {
    NVIMGCODEC_STATUS_SUCCESS = 0,
    NVIMGCODEC_STATUS_NOT_INITIALIZED = 1,
    NVIMGCODEC_STATUS_INVALID_PARAMETER = 2,
    NVIMGCODEC_STATUS_BAD_CODESTREAM = 3,
    NVIMGCODEC_STATUS_CODESTREAM_UNSUPPORTED = 4,
    NVIMGCODEC_STATUS_ALLOCATOR_FAILURE = 5,
    NVIMGCODEC_STATUS_EXECUTION_FAILED = 6,
    NVIMGCODEC_STATUS_ARCH_MISMATCH = 7,
    NVIMGCODEC_STATUS_INTERNAL_ERROR = 8,
    NVIMGCODEC_STATUS_IMPLEMENTATION_UNSUPPORTED = 9,
    NVIMGCODEC_STATUS_MISSED_DEPENDENCIES = 10,
    NVIMGCODEC_STATUS_EXTENSION_NOT_INITIALIZED = 11,
    NVIMGCODEC_STATUS_EXTENSION_INVALID_PARAMETER = 12,
    NVIMGCODEC_STATUS_EXTENSION_BAD_CODE_STREAM = 13,
    NVIMGCODEC_STATUS_EXTENSION_CODESTREAM_UNSUPPORTED = 14,
    NVIMGCODEC_STATUS_EXTENSION_ALLOCATOR_FAILURE = 15,
    NVIMGCODEC_STATUS_EXTENSION_ARCH_MISMATCH = 16,
    NVIMGCODEC_STATUS_EXTENSION_INTERNAL_ERROR = 17,
    NVIMGCODEC_STATUS_EXTENSION_IMPLEMENTATION_NOT_SUPPORTED = 18,
    NVIMGCODEC_STATUS_EXTENSION_INCOMPLETE_BITSTREAM = 19,
    NVIMGCODEC_STATUS_EXTENSION_EXECUTION_FAILED = 20,
    NVIMGCODEC_STATUS_EXTENSION_CUDA_CALL_ERROR = 21, /* used for HIP runtime failures on this platform */
    NVIMGCODEC_STATUS_ENUM_FORCE_INT = INT32_MAX
} nvimgcodecStatus_t;

/* ref:340-356 -- (bitdepth << 8) | ordinal */
typedef enum {
    NVIMGCODEC_SAMPLE_DATA_TYPE_UNKNOWN = 0,
    NVIMGCODEC_SAMPLE_DATA_TYPE_INT8 = 0x0801,
    NVIMGCODEC_SAMPLE_DATA_TYPE_UINT8 = 0x0802,
    NVIMGCODEC_SAMPLE_DATA_TYPE_INT16 = 0x1003,
    NVIMGCODEC_SAMPLE_DATA_TYPE_UINT16 = 0x1004,
    NVIMGCODEC_SAMPLE_DATA_TYPE_INT32 = 0x2005,
    NVIMGCODEC_SAMPLE_DATA_TYPE_UINT32 = 0x2006,
    NVIMGCODEC_SAMPLE_DATA_TYPE_INT64 = 0x4007,
    NVIMGCODEC_SAMPLE_DATA_TYPE_UINT64 = 0x4008,
    NVIMGCODEC_SAMPLE_DATA_TYPE_FLOAT16 = 0x1009,
    NVIMGCODEC_SAMPLE_DATA_TYPE_FLOAT32 = 0x200B,
    NVIMGCODEC_SAMPLE_DATA_TYPE_FLOAT64 = 0x400D,
    NVIMGCODEC_SAMPLE_DATA_TYPE_UNSUPPORTED = -1,
    NVIMGCODEC_SAMPLE_ENUM_FORCE_INT = INT32_MAX
} nvimgcodecSampleDataType_t;

/* ref:361-374 */
typedef enum {
    NVIMGCODEC_SAMPLING_NONE = 0,
    NVIMGCODEC_SAMPLING_444 = NVIMGCODEC_SAMPLING_NONE,
    NVIMGCODEC_SAMPLING_422 = 2,
    NVIMGCODEC_SAMPLING_420 = 3,
    NVIMGCODEC_SAMPLING_440 = 4,
    NVIMGCODEC_SAMPLING_411 = 5,
    NVIMGCODEC_SAMPLING_410 = 6,
    NVIMGCODEC_SAMPLING_GRAY = 7,
    NVIMGCODEC_SAMPLING_410V = 8,
    NVIMGCODEC_SAMPLING_UNSUPPORTED = -1,
    NVIMGCODEC_SAMPLING_ENUM_FORCE_INT = INT32_MAX
} nvimgcodecChromaSubsampling_t;

/* ref:379-393.  P_ = one plane per channel, I_ = channels interleaved in one plane. */
typedef enum {
    NVIMGCODEC_SAMPLEFORMAT_UNKNOWN = 0,
    NVIMGCODEC_SAMPLEFORMAT_P_UNCHANGED = 1,
    NVIMGCODEC_SAMPLEFORMAT_I_UNCHANGED = 2,
    NVIMGCODEC_SAMPLEFORMAT_P_RGB = 3,
    NVIMGCODEC_SAMPLEFORMAT_I_RGB = 4,
    NVIMGCODEC_SAMPLEFORMAT_P_BGR = 5,
    NVIMGCODEC_SAMPLEFORMAT_I_BGR = 6,
    NVIMGCODEC_SAMPLEFORMAT_P_Y = 7,
    NVIMGCODEC_SAMPLEFORMAT_P_YUV = 9,
    NVIMGCODEC_SAMPLEFORMAT_UNSUPPORTED = -1,
    NVIMGCODEC_SAMPLEFORMAT_ENUM_FORCE_INT = INT32_MAX
} nvimgcodecSampleFormat_t;

/* ref:398-409 */
typedef enum {
    NVIMGCODEC_COLORSPEC_UNKNOWN = 0,
    NVIMGCODEC_COLORSPEC_UNCHANGED = NVIMGCODEC_COLORSPEC_UNKNOWN,
    NVIMGCODEC_COLORSPEC_SRGB = 1,
    NVIMGCODEC_COLORSPEC_GRAY = 2,
    NVIMGCODEC_COLORSPEC_SYCC = 3,
    NVIMGCODEC_COLORSPEC_CMYK = 4,
    NVIMGCODEC_COLORSPEC_YCCK = 5,
    NVIMGCODEC_COLORSPEC_UNSUPPORTED = -1,
    NVIMGCODEC_COLORSPEC_ENUM_FORCE_INT = INT32_MAX
} nvimgcodecColorSpec_t;

/* ref:414-423: clockwise rotation in degrees (multiples of 90) + flips */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    int rotated;
    int flip_x;
    int flip_y;
} nvimgcodecOrientation_t;

/* ref:428-441 */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    uint32_t width;
    uint32_t height;
    size_t row_stride; /* bytes between rows of this plane */
    uint32_t num_channels;
    nvimgcodecSampleDataType_t sample_type;
    uint8_t precision; /* 0 == full bit depth of sample_type */
} nvimgcodecImagePlaneInfo_t;

/* ref:446-455: start/end are per dimension, [0]=y, [1]=x (see libjpeg_turbo_decoder.cpp:361-365) */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    int ndim;
    int start[NVIMGCODEC_MAX_NUM_DIM];
    int end[NVIMGCODEC_MAX_NUM_DIM];
} nvimgcodecRegion_t;

/* ref:460-467 */
typedef enum {
    NVIMGCODEC_IMAGE_BUFFER_KIND_UNKNOWN = 0,
    NVIMGCODEC_IMAGE_BUFFER_KIND_STRIDED_DEVICE = 1,
    NVIMGCODEC_IMAGE_BUFFER_KIND_STRIDED_HOST = 2,
    NVIMGCODEC_IMAGE_BUFFER_KIND_UNSUPPORTED = -1,
    NVIMGCODEC_IMAGE_BUFFER_KIND_ENUM_FORCE_INT = INT32_MAX
} nvimgcodecImageBufferKind_t;

/* ref:474-497.  2240 bytes; planes are laid out back to back inside `buffer`
 * (plane p starts at sum_{q<p} row_stride[q]*height[q]; nvjpeg/cuda_decoder.cpp:532-538). */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    char codec_name[NVIMGCODEC_MAX_CODEC_NAME_SIZE];
    nvimgcodecColorSpec_t color_spec;
    nvimgcodecChromaSubsampling_t chroma_subsampling;
    nvimgcodecSampleFormat_t sample_format;
    nvimgcodecOrientation_t orientation;
    nvimgcodecRegion_t region;
    uint32_t num_planes;
    nvimgcodecImagePlaneInfo_t plane_info[NVIMGCODEC_MAX_NUM_PLANES];
    void* buffer;
    size_t buffer_size;
    nvimgcodecImageBufferKind_t buffer_kind;
    cudaStream_t cuda_stream; /* stream the consumer will use; the plugin orders it after its own work */
} nvimgcodecImageInfo_t;

/* ref:502-521: value == SOFn marker byte */
typedef enum {
    NVIMGCODEC_JPEG_ENCODING_UNKNOWN = 0x0,
    NVIMGCODEC_JPEG_ENCODING_BASELINE_DCT = 0xc0,
    NVIMGCODEC_JPEG_ENCODING_EXTENDED_SEQUENTIAL_DCT_HUFFMAN = 0xc1,
    NVIMGCODEC_JPEG_ENCODING_PROGRESSIVE_DCT_HUFFMAN = 0xc2,
    NVIMGCODEC_JPEG_ENCODING_LOSSLESS_HUFFMAN = 0xc3,
    NVIMGCODEC_JPEG_ENCODING_DIFFERENTIAL_SEQUENTIAL_DCT_HUFFMAN = 0xc5,
    NVIMGCODEC_JPEG_ENCODING_DIFFERENTIAL_PROGRESSIVE_DCT_HUFFMAN = 0xc6,
    NVIMGCODEC_JPEG_ENCODING_DIFFERENTIAL_LOSSLESS_HUFFMAN = 0xc7,
    NVIMGCODEC_JPEG_ENCODING_RESERVED_FOR_JPEG_EXTENSIONS = 0xc8,
    NVIMGCODEC_JPEG_ENCODING_EXTENDED_SEQUENTIAL_DCT_ARITHMETIC = 0xc9,
    NVIMGCODEC_JPEG_ENCODING_PROGRESSIVE_DCT_ARITHMETIC = 0xca,
    NVIMGCODEC_JPEG_ENCODING_LOSSLESS_ARITHMETIC = 0xcb,
    NVIMGCODEC_JPEG_ENCODING_DIFFERENTIAL_SEQUENTIAL_DCT_ARITHMETIC = 0xcd,
    NVIMGCODEC_JPEG_ENCODING_DIFFERENTIAL_PROGRESSIVE_DCT_ARITHMETIC = 0xce,
    NVIMGCODEC_JPEG_ENCODING_DIFFERENTIAL_LOSSLESS_ARITHMETIC = 0xcf,
    NVIMGCODEC_JPEG_ENCODING_ENUM_FORCE_INT = INT32_MAX
} nvimgcodecJpegEncoding_t;

/* ref:526-533: chained through nvimgcodecImageInfo_t::struct_next */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    nvimgcodecJpegEncoding_t encoding;
} nvimgcodecJpegImageInfo_t;

/* ref:538-544 */
typedef enum {
    NVIMGCODEC_BACKEND_KIND_CPU_ONLY = 1,
    NVIMGCODEC_BACKEND_KIND_GPU_ONLY = 2,
    NVIMGCODEC_BACKEND_KIND_HYBRID_CPU_GPU = 3,
    NVIMGCODEC_BACKEND_KIND_HW_GPU_ONLY = 4
} nvimgcodecBackendKind_t;

/* ref:549-563: fraction of the batch a backend may take before reporting SATURATED */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    float load_hint;
} nvimgcodecBackendParams_t;

/* ref:568-576 */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    nvimgcodecBackendKind_t kind;
    nvimgcodecBackendParams_t params;
} nvimgcodecBackend_t;

/* ref:583-617.  Bit 0 set == "could be processed" (possibly with other params);
 * low bits 0b11 == hard failure.  Carried as uint32_t. */
typedef enum {
    NVIMGCODEC_PROCESSING_STATUS_UNKNOWN = 0x0,
    NVIMGCODEC_PROCESSING_STATUS_SUCCESS = 0x1,
    NVIMGCODEC_PROCESSING_STATUS_SATURATED = 0x2,
    NVIMGCODEC_PROCESSING_STATUS_FAIL = 0x3,
    NVIMGCODEC_PROCESSING_STATUS_IMAGE_CORRUPTED = 0x7,
    NVIMGCODEC_PROCESSING_STATUS_CODEC_UNSUPPORTED = 0xb,
    NVIMGCODEC_PROCESSING_STATUS_BACKEND_UNSUPPORTED = 0x13,
    NVIMGCODEC_PROCESSING_STATUS_ENCODING_UNSUPPORTED = 0x23,
    NVIMGCODEC_PROCESSING_STATUS_RESOLUTION_UNSUPPORTED = 0x43,
    NVIMGCODEC_PROCESSING_STATUS_CODESTREAM_UNSUPPORTED = 0x83,
    NVIMGCODEC_PROCESSING_STATUS_COLOR_SPEC_UNSUPPORTED = 0x5,
    NVIMGCODEC_PROCESSING_STATUS_ORIENTATION_UNSUPPORTED = 0x9,
    NVIMGCODEC_PROCESSING_STATUS_ROI_UNSUPPORTED = 0x11,
    NVIMGCODEC_PROCESSING_STATUS_SAMPLING_UNSUPPORTED = 0x21,
    NVIMGCODEC_PROCESSING_STATUS_SAMPLE_TYPE_UNSUPPORTED = 0x41,
    NVIMGCODEC_PROCESSING_STATUS_SAMPLE_FORMAT_UNSUPPORTED = 0x81,
    NVIMGCODEC_PROCESSING_STATUS_NUM_PLANES_UNSUPPORTED = 0x101,
    NVIMGCODEC_PROCESSING_STATUS_NUM_CHANNELS_UNSUPPORTED = 0x201,
    NVIMGCODEC_PROCESSING_STATUS_ENUM_FORCE_INT = INT32_MAX
} nvimgcodecProcessingStatus;
typedef uint32_t nvimgcodecProcessingStatus_t;

/* ref:628-636 */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    int apply_exif_orientation;
    int enable_roi;
} nvimgcodecDecodeParams_t;

/* ref:641-660 */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    float quality;
    float target_psnr;
} nvimgcodecEncodeParams_t;

/* ref:665-702 (JPEG2000 pieces: declared for layout completeness only) */
typedef enum {
    NVIMGCODEC_JPEG2K_PROG_ORDER_LRCP = 0,
    NVIMGCODEC_JPEG2K_PROG_ORDER_RLCP = 1,
    NVIMGCODEC_JPEG2K_PROG_ORDER_RPCL = 2,
    NVIMGCODEC_JPEG2K_PROG_ORDER_PCRL = 3,
    NVIMGCODEC_JPEG2K_PROG_ORDER_CPRL = 4,
    NVIMGCODEC_JPEG2K_PROG_ORDER_ENUM_FORCE_INT = INT32_MAX
} nvimgcodecJpeg2kProgOrder_t;

typedef enum {
    NVIMGCODEC_JPEG2K_STREAM_J2K = 0,
    NVIMGCODEC_JPEG2K_STREAM_JP2 = 1,
    NVIMGCODEC_JPEG2K_STREAM_ENUM_FORCE_INT = INT32_MAX
} nvimgcodecJpeg2kBitstreamType_t;

typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    nvimgcodecJpeg2kBitstreamType_t stream_type;
    nvimgcodecJpeg2kProgOrder_t prog_order;
    uint32_t num_resolutions;
    uint32_t code_block_w;
    uint32_t code_block_h;
    int irreversible;
} nvimgcodecJpeg2kEncodeParams_t;

/* ref:707-717: chained through nvimgcodecEncodeParams_t::struct_next */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    int optimized_huffman;
} nvimgcodecJpegEncodeParams_t;

/* ref:722-753 */
typedef enum {
    NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_NONE = 0x00000000,
    NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_TRACE = 0x00000001,
    NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_DEBUG = 0x00000010,
    NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_INFO = 0x00000100,
    NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_WARNING = 0x00001000,
    NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_ERROR = 0x00010000,
    NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_FATAL = 0x00100000,
    NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_ALL = 0x0FFFFFFF,
    NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_DEFAULT =
        NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_WARNING | NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_ERROR | NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_FATAL,
    NVIMGCODEC_DEBUG_MESSAGE_SEVERITY_ENUM_FORCE_INT = INT32_MAX
} nvimgcodecDebugMessageSeverity_t;

typedef enum {
    NVIMGCODEC_DEBUG_MESSAGE_CATEGORY_NONE = 0x00000000,
    NVIMGCODEC_DEBUG_MESSAGE_CATEGORY_GENERAL = 0x00000001,
    NVIMGCODEC_DEBUG_MESSAGE_CATEGORY_VALIDATION = 0x00000010,
    NVIMGCODEC_DEBUG_MESSAGE_CATEGORY_PERFORMANCE = 0x00000100,
    NVIMGCODEC_DEBUG_MESSAGE_CATEGORY_ALL = 0x0FFFFFFF,
    NVIMGCODEC_DEBUG_MESSAGE_CATEGORY_ENUM_FORCE_INT = INT32_MAX
} nvimgcodecDebugMessageCategory_t;

/* ref:758-769 */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    const char* message;
    uint32_t internal_status_id;
    const char* codec;
    const char* codec_id;
    uint32_t codec_version;
} nvimgcodecDebugMessageData_t;

/* ref:780-793 */
typedef int (*nvimgcodecDebugCallback_t)(const nvimgcodecDebugMessageSeverity_t message_severity,
                                         const nvimgcodecDebugMessageCategory_t message_category,
                                         const nvimgcodecDebugMessageData_t* callback_data, void* user_data);

typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    uint32_t message_severity;
    uint32_t message_category;
    nvimgcodecDebugCallback_t user_callback;
    void* user_data;
} nvimgcodecDebugMessengerDesc_t;

/* ref:800-828.  The only threading facility a plugin gets.  task() runs on a pool
 * thread with thread_id in [0, getNumThreads()). */
typedef struct {
    nvimgcodecStructureType_t struct_type;
    size_t struct_size;
    const void* struct_next;
    void* instance;
    nvimgcodecStatus_t (*launch)(void* instance, int device_id, int sample_idx, void* task_context,
                                 void (*task)(int thread_id, int sample_idx, void* task_context));
    int (*getNumThreads)(void* instance);
} nvimgcodecExecutorDesc_t;

/* ref:833-852 */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    nvimgcodecDeviceAllocator_t* device_allocator;
    nvimgcodecPinnedAllocator_t* pinned_allocator;
    int max_num_cpu_threads;
    nvimgcodecExecutorDesc_t* executor; /* never NULL at plugin level */
    int device_id;
    int pre_init;
    int num_backends;
    const nvimgcodecBackend_t* backends;
} nvimgcodecExecutionParams_t;

/* ref:860-976: byte source/sink of a code stream.  map() may hand back NULL -> use read(). */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    void* instance;
    nvimgcodecStatus_t (*read)(void* instance, size_t* output_size, void* buf, size_t bytes);
    nvimgcodecStatus_t (*write)(void* instance, size_t* output_size, void* buf, size_t bytes);
    nvimgcodecStatus_t (*putc)(void* instance, size_t* output_size, unsigned char ch);
    nvimgcodecStatus_t (*skip)(void* instance, size_t count);
    nvimgcodecStatus_t (*seek)(void* instance, ptrdiff_t offset, int whence);
    nvimgcodecStatus_t (*tell)(void* instance, ptrdiff_t* offset);
    nvimgcodecStatus_t (*size)(void* instance, size_t* size);
    nvimgcodecStatus_t (*reserve)(void* instance, size_t bytes);
    nvimgcodecStatus_t (*flush)(void* instance);
    nvimgcodecStatus_t (*map)(void* instance, void** buffer, size_t offset, size_t size);
    nvimgcodecStatus_t (*unmap)(void* instance, void* buffer, size_t size);
} nvimgcodecIoStreamDesc_t;

/* ref:981-1001 */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    void* instance;
    nvimgcodecIoStreamDesc_t* io_stream;
    nvimgcodecStatus_t (*getImageInfo)(void* instance, nvimgcodecImageInfo_t* image_info);
} nvimgcodecCodeStreamDesc_t;

/* ref:1006-1029.  imageReady must be called exactly once per sample. */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    void* instance;
    nvimgcodecStatus_t (*getImageInfo)(void* instance, nvimgcodecImageInfo_t* image_info);
    nvimgcodecStatus_t (*imageReady)(void* instance, nvimgcodecProcessingStatus_t processing_status);
} nvimgcodecImageDesc_t;

/* ref:1034-1082 */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    void* instance;
    const char* id;
    const char* codec;
    nvimgcodecStatus_t (*canParse)(void* instance, int* result, nvimgcodecCodeStreamDesc_t* code_stream);
    nvimgcodecStatus_t (*create)(void* instance, nvimgcodecParser_t* parser);
    nvimgcodecStatus_t (*destroy)(nvimgcodecParser_t parser);
    nvimgcodecStatus_t (*getImageInfo)(nvimgcodecParser_t parser, nvimgcodecImageInfo_t* image_info,
                                       nvimgcodecCodeStreamDesc_t* code_stream);
} nvimgcodecParserDesc_t;

/* ref:1087-1145 */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    void* instance;
    const char* id;
    const char* codec;
    nvimgcodecBackendKind_t backend_kind;
    nvimgcodecStatus_t (*create)(void* instance, nvimgcodecEncoder_t* encoder, const nvimgcodecExecutionParams_t* exec_params,
                                 const char* options);
    nvimgcodecStatus_t (*destroy)(nvimgcodecEncoder_t encoder);
    nvimgcodecStatus_t (*canEncode)(nvimgcodecEncoder_t encoder, nvimgcodecProcessingStatus_t* status, nvimgcodecImageDesc_t** images,
                                    nvimgcodecCodeStreamDesc_t** code_streams, int batch_size, const nvimgcodecEncodeParams_t* params);
    nvimgcodecStatus_t (*encode)(nvimgcodecEncoder_t encoder, nvimgcodecImageDesc_t** images, nvimgcodecCodeStreamDesc_t** code_streams,
                                 int batch_size, const nvimgcodecEncodeParams_t* params);
} nvimgcodecEncoderDesc_t;

/* ref:1150-1209 */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    void* instance;
    const char* id;
    const char* codec;
    nvimgcodecBackendKind_t backend_kind;
    nvimgcodecStatus_t (*create)(void* instance, nvimgcodecDecoder_t* decoder, const nvimgcodecExecutionParams_t* exec_params,
                                 const char* options);
    nvimgcodecStatus_t (*destroy)(nvimgcodecDecoder_t decoder);
    nvimgcodecStatus_t (*canDecode)(nvimgcodecDecoder_t decoder, nvimgcodecProcessingStatus_t* status,
                                    nvimgcodecCodeStreamDesc_t** code_streams, nvimgcodecImageDesc_t** images, int batch_size,
                                    const nvimgcodecDecodeParams_t* params);
    nvimgcodecStatus_t (*decode)(nvimgcodecDecoder_t decoder, nvimgcodecCodeStreamDesc_t** code_streams, nvimgcodecImageDesc_t** images,
                                 int batch_size, const nvimgcodecDecodeParams_t* params);
} nvimgcodecDecoderDesc_t;

/* ref:1219-1229: lower value == tried earlier */
typedef enum {
    NVIMGCODEC_PRIORITY_HIGHEST = 0,
    NVIMGCODEC_PRIORITY_VERY_HIGH = 100,
    NVIMGCODEC_PRIORITY_HIGH = 200,
    NVIMGCODEC_PRIORITY_NORMAL = 300,
    NVIMGCODEC_PRIORITY_LOW = 400,
    NVIMGCODEC_PRIORITY_VERY_LOW = 500,
    NVIMGCODEC_PRIORITY_LOWEST = 1000,
    NVIMGCODEC_PRIORITY_ENUM_FORCE_INT = INT32_MAX
} nvimgcodecPriority_t;

/* ref:1239-1240 */
typedef nvimgcodecStatus_t (*nvimgcodecLogFunc_t)(void* instance, const nvimgcodecDebugMessageSeverity_t message_severity,
                                                  const nvimgcodecDebugMessageCategory_t message_category,
                                                  const nvimgcodecDebugMessageData_t* data);

/* ref:1245-1315: what the framework hands to an extension's create() */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    void* instance;
    const char* id;
    uint32_t version;
    uint32_t ext_api_version;
    uint32_t cudart_version;
    nvimgcodecLogFunc_t log;
    nvimgcodecStatus_t (*registerEncoder)(void* instance, const nvimgcodecEncoderDesc_t* desc, float priority);
    nvimgcodecStatus_t (*unregisterEncoder)(void* instance, const nvimgcodecEncoderDesc_t* desc);
    nvimgcodecStatus_t (*registerDecoder)(void* instance, const nvimgcodecDecoderDesc_t* desc, float priority);
    nvimgcodecStatus_t (*unregisterDecoder)(void* instance, const nvimgcodecDecoderDesc_t* desc);
    nvimgcodecStatus_t (*registerParser)(void* instance, const nvimgcodecParserDesc_t* desc, float priority);
    nvimgcodecStatus_t (*unregisterParser)(void* instance, const nvimgcodecParserDesc_t* desc);
} nvimgcodecFrameworkDesc_t;

/* ref:1320-1348 */
typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    void* instance;
    const char* id;
    uint32_t version;
    uint32_t ext_api_version;
    nvimgcodecStatus_t (*create)(void* instance, nvimgcodecExtension_t* extension, const nvimgcodecFrameworkDesc_t* framework);
    nvimgcodecStatus_t (*destroy)(nvimgcodecExtension_t extension);
} nvimgcodecExtensionDesc_t;

/* ref:1356-1364: the one symbol an extension module exports */
typedef nvimgcodecStatus_t (*nvimgcodecExtensionModuleEntryFunc_t)(nvimgcodecExtensionDesc_t* ext_desc);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecExtensionModuleEntry(nvimgcodecExtensionDesc_t* ext_desc);

/* ---- public (application-side) API, ref:1372-1706.  Implemented by the host harness
 * in this repository for the subset the JPEG path needs. ---- */
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecGetProperties(nvimgcodecProperties_t* properties);

typedef struct {
    NVIMGCODEC_STRUCT_HEAD;
    int load_builtin_modules;
    int load_extension_modules;
    const char* extension_modules_path;
    int create_debug_messenger;
    const nvimgcodecDebugMessengerDesc_t* debug_messenger_desc;
    uint32_t message_severity;
    uint32_t message_category;
} nvimgcodecInstanceCreateInfo_t;

NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecInstanceCreate(nvimgcodecInstance_t* instance, const nvimgcodecInstanceCreateInfo_t* create_info);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecInstanceDestroy(nvimgcodecInstance_t instance);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecExtensionCreate(nvimgcodecInstance_t instance, nvimgcodecExtension_t* extension,
                                                           nvimgcodecExtensionDesc_t* extension_desc);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecExtensionDestroy(nvimgcodecExtension_t extension);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecDebugMessengerCreate(nvimgcodecInstance_t instance, nvimgcodecDebugMessenger_t* dbg_messenger,
                                                                const nvimgcodecDebugMessengerDesc_t* messenger_desc);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecDebugMessengerDestroy(nvimgcodecDebugMessenger_t dbg_messenger);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecFutureWaitForAll(nvimgcodecFuture_t future);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecFutureDestroy(nvimgcodecFuture_t future);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecFutureGetProcessingStatus(nvimgcodecFuture_t future,
                                                                     nvimgcodecProcessingStatus_t* processing_status, size_t* size);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecImageCreate(nvimgcodecInstance_t instance, nvimgcodecImage_t* image,
                                                       const nvimgcodecImageInfo_t* image_info);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecImageDestroy(nvimgcodecImage_t image);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecImageGetImageInfo(nvimgcodecImage_t image, nvimgcodecImageInfo_t* image_info);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecCodeStreamCreateFromFile(nvimgcodecInstance_t instance, nvimgcodecCodeStream_t* code_stream,
                                                                    const char* file_name);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecCodeStreamCreateFromHostMem(nvimgcodecInstance_t instance, nvimgcodecCodeStream_t* code_stream,
                                                                       const unsigned char* data, size_t length);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecCodeStreamCreateToFile(nvimgcodecInstance_t instance, nvimgcodecCodeStream_t* code_stream,
                                                                  const char* file_name, const nvimgcodecImageInfo_t* image_info);
typedef unsigned char* (*nvimgcodecResizeBufferFunc_t)(void* ctx, size_t req_size);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecCodeStreamCreateToHostMem(nvimgcodecInstance_t instance, nvimgcodecCodeStream_t* code_stream,
                                                                     void* ctx, nvimgcodecResizeBufferFunc_t resize_buffer_func,
                                                                     const nvimgcodecImageInfo_t* image_info);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecCodeStreamDestroy(nvimgcodecCodeStream_t code_stream);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecCodeStreamGetImageInfo(nvimgcodecCodeStream_t code_stream, nvimgcodecImageInfo_t* image_info);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecDecoderCreate(nvimgcodecInstance_t instance, nvimgcodecDecoder_t* decoder,
                                                         const nvimgcodecExecutionParams_t* exec_params, const char* options);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecDecoderDestroy(nvimgcodecDecoder_t decoder);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecDecoderCanDecode(nvimgcodecDecoder_t decoder, const nvimgcodecCodeStream_t* streams,
                                                            const nvimgcodecImage_t* images, int batch_size,
                                                            const nvimgcodecDecodeParams_t* params,
                                                            nvimgcodecProcessingStatus_t* processing_status, int force_format);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecDecoderDecode(nvimgcodecDecoder_t decoder, const nvimgcodecCodeStream_t* streams,
                                                         const nvimgcodecImage_t* images, int batch_size,
                                                         const nvimgcodecDecodeParams_t* params, nvimgcodecFuture_t* future);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecEncoderCreate(nvimgcodecInstance_t instance, nvimgcodecEncoder_t* encoder,
                                                         const nvimgcodecExecutionParams_t* exec_params, const char* options);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecEncoderDestroy(nvimgcodecEncoder_t encoder);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecEncoderCanEncode(nvimgcodecEncoder_t encoder, const nvimgcodecImage_t* images,
                                                            const nvimgcodecCodeStream_t* streams, int batch_size,
                                                            const nvimgcodecEncodeParams_t* params,
                                                            nvimgcodecProcessingStatus_t* processing_status, int force_format);
NVIMGCODECAPI nvimgcodecStatus_t nvimgcodecEncoderEncode(nvimgcodecEncoder_t encoder, const nvimgcodecImage_t* images,
                                                         const nvimgcodecCodeStream_t* streams, int batch_size,
                                                         const nvimgcodecEncodeParams_t* params, nvimgcodecFuture_t* future);

#ifdef __cplusplus
} /* extern "C" */
#endif

/* ---- layout pins (LP64), SURVEY.md Appendix B ---- */
#if defined(__cplusplus) && defined(__LP64__)
#define NVIMGCODEC_ABI_PIN(T, size) static_assert(sizeof(T) == (size), "ABI size mismatch: " #T)
#define NVIMGCODEC_ABI_OFF(T, f, off) static_assert(offsetof(T, f) == (off), "ABI offset mismatch: " #T "::" #f)
NVIMGCODEC_ABI_PIN(nvimgcodecImagePlaneInfo_t, 56);
NVIMGCODEC_ABI_OFF(nvimgcodecImagePlaneInfo_t, width, 24);
NVIMGCODEC_ABI_OFF(nvimgcodecImagePlaneInfo_t, row_stride, 32);
NVIMGCODEC_ABI_OFF(nvimgcodecImagePlaneInfo_t, sample_type, 44);
NVIMGCODEC_ABI_PIN(nvimgcodecOrientation_t, 40);
NVIMGCODEC_ABI_PIN(nvimgcodecRegion_t, 72);
NVIMGCODEC_ABI_PIN(nvimgcodecImageInfo_t, 2240);
NVIMGCODEC_ABI_OFF(nvimgcodecImageInfo_t, codec_name, 24);
NVIMGCODEC_ABI_OFF(nvimgcodecImageInfo_t, color_spec, 280);
NVIMGCODEC_ABI_OFF(nvimgcodecImageInfo_t, sample_format, 288);
NVIMGCODEC_ABI_OFF(nvimgcodecImageInfo_t, orientation, 296);
NVIMGCODEC_ABI_OFF(nvimgcodecImageInfo_t, region, 336);
NVIMGCODEC_ABI_OFF(nvimgcodecImageInfo_t, num_planes, 408);
NVIMGCODEC_ABI_OFF(nvimgcodecImageInfo_t, plane_info, 416);
NVIMGCODEC_ABI_OFF(nvimgcodecImageInfo_t, buffer, 2208);
NVIMGCODEC_ABI_OFF(nvimgcodecImageInfo_t, buffer_size, 2216);
NVIMGCODEC_ABI_OFF(nvimgcodecImageInfo_t, buffer_kind, 2224);
NVIMGCODEC_ABI_OFF(nvimgcodecImageInfo_t, cuda_stream, 2232);
NVIMGCODEC_ABI_PIN(nvimgcodecJpegImageInfo_t, 32);
NVIMGCODEC_ABI_PIN(nvimgcodecBackendParams_t, 32);
NVIMGCODEC_ABI_PIN(nvimgcodecDecodeParams_t, 32);
NVIMGCODEC_ABI_PIN(nvimgcodecEncodeParams_t, 32);
NVIMGCODEC_ABI_PIN(nvimgcodecJpegEncodeParams_t, 32);
NVIMGCODEC_ABI_PIN(nvimgcodecBackend_t, 64);
NVIMGCODEC_ABI_PIN(nvimgcodecExecutorDesc_t, 48);
NVIMGCODEC_ABI_PIN(nvimgcodecExecutionParams_t, 80);
NVIMGCODEC_ABI_OFF(nvimgcodecExecutionParams_t, executor, 48);
NVIMGCODEC_ABI_OFF(nvimgcodecExecutionParams_t, device_id, 56);
NVIMGCODEC_ABI_OFF(nvimgcodecExecutionParams_t, backends, 72);
NVIMGCODEC_ABI_PIN(nvimgcodecDeviceAllocator_t, 56);
NVIMGCODEC_ABI_PIN(nvimgcodecPinnedAllocator_t, 56);
NVIMGCODEC_ABI_PIN(nvimgcodecIoStreamDesc_t, 120);
NVIMGCODEC_ABI_PIN(nvimgcodecCodeStreamDesc_t, 48);
NVIMGCODEC_ABI_PIN(nvimgcodecImageDesc_t, 48);
NVIMGCODEC_ABI_OFF(nvimgcodecImageDesc_t, imageReady, 40);
NVIMGCODEC_ABI_PIN(nvimgcodecParserDesc_t, 80);
NVIMGCODEC_ABI_PIN(nvimgcodecDecoderDesc_t, 88);
NVIMGCODEC_ABI_OFF(nvimgcodecDecoderDesc_t, backend_kind, 48);
NVIMGCODEC_ABI_OFF(nvimgcodecDecoderDesc_t, canDecode, 72);
NVIMGCODEC_ABI_OFF(nvimgcodecDecoderDesc_t, decode, 80);
NVIMGCODEC_ABI_PIN(nvimgcodecEncoderDesc_t, 88);
NVIMGCODEC_ABI_PIN(nvimgcodecFrameworkDesc_t, 112);
NVIMGCODEC_ABI_OFF(nvimgcodecFrameworkDesc_t, log, 56);
NVIMGCODEC_ABI_OFF(nvimgcodecFrameworkDesc_t, registerDecoder, 80);
NVIMGCODEC_ABI_PIN(nvimgcodecExtensionDesc_t, 64);
NVIMGCODEC_ABI_OFF(nvimgcodecExtensionDesc_t, create, 48);
NVIMGCODEC_ABI_PIN(nvimgcodecDebugMessageData_t, 64);
NVIMGCODEC_ABI_PIN(nvimgcodecInstanceCreateInfo_t, 64);
NVIMGCODEC_ABI_PIN(nvimgcodecProperties_t, 40);
#undef NVIMGCODEC_ABI_PIN
#undef NVIMGCODEC_ABI_OFF
#endif

#endif /* NVIMGCODEC_ABI_H_ */
