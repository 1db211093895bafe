"""ctypes mirror of include/nvimgcodec_abi.h (the nvImageCodec C-ABI) -- used by the Python front-end and by tests that
act as a fake plugin or a fake framework across the real function tables."""
import ctypes as C

MAX_CODEC_NAME_SIZE = 256
MAX_NUM_DIM = 5
MAX_NUM_PLANES = 32
DEVICE_CURRENT = -1
DEVICE_CPU_ONLY = -99999

# nvimgcodecStructureType_t
(ST_PROPERTIES, ST_INSTANCE_CREATE_INFO, ST_DEVICE_ALLOCATOR, ST_PINNED_ALLOCATOR, ST_DECODE_PARAMS, ST_ENCODE_PARAMS, ST_ORIENTATION,
 ST_REGION, ST_IMAGE_INFO, ST_IMAGE_PLANE_INFO, ST_JPEG_IMAGE_INFO, ST_JPEG_ENCODE_PARAMS, ST_JPEG2K_ENCODE_PARAMS, ST_BACKEND,
 ST_IO_STREAM_DESC, ST_FRAMEWORK_DESC, ST_DECODER_DESC, ST_ENCODER_DESC, ST_PARSER_DESC, ST_IMAGE_DESC, ST_CODE_STREAM_DESC,
 ST_DEBUG_MESSENGER_DESC, ST_DEBUG_MESSAGE_DATA, ST_EXTENSION_DESC, ST_EXECUTOR_DESC, ST_BACKEND_PARAMS, ST_EXECUTION_PARAMS) = range(27)

STATUS_SUCCESS = 0
STATUS_INVALID_PARAMETER = 2
STATUS_BAD_CODESTREAM = 3
STATUS_CODESTREAM_UNSUPPORTED = 4

SAMPLE_DATA_TYPE_UINT8 = 0x0802
SAMPLE_DATA_TYPE_UINT16 = 0x1004

SAMPLING_444, SAMPLING_422, SAMPLING_420, SAMPLING_440, SAMPLING_411, SAMPLING_410, SAMPLING_GRAY, SAMPLING_410V = 0, 2, 3, 4, 5, 6, 7, 8
SAMPLING_UNSUPPORTED = -1

SAMPLEFORMAT_P_UNCHANGED, SAMPLEFORMAT_I_UNCHANGED, SAMPLEFORMAT_P_RGB, SAMPLEFORMAT_I_RGB = 1, 2, 3, 4
SAMPLEFORMAT_P_BGR, SAMPLEFORMAT_I_BGR, SAMPLEFORMAT_P_Y, SAMPLEFORMAT_P_YUV = 5, 6, 7, 9

COLORSPEC_UNCHANGED, COLORSPEC_SRGB, COLORSPEC_GRAY, COLORSPEC_SYCC, COLORSPEC_CMYK, COLORSPEC_YCCK = 0, 1, 2, 3, 4, 5

BUFFER_KIND_STRIDED_DEVICE, BUFFER_KIND_STRIDED_HOST = 1, 2

JPEG_ENCODING_BASELINE_DCT, JPEG_ENCODING_EXTENDED_SEQUENTIAL_DCT_HUFFMAN, JPEG_ENCODING_PROGRESSIVE_DCT_HUFFMAN = 0xC0, 0xC1, 0xC2

BACKEND_KIND_CPU_ONLY, BACKEND_KIND_GPU_ONLY, BACKEND_KIND_HYBRID_CPU_GPU, BACKEND_KIND_HW_GPU_ONLY = 1, 2, 3, 4

PS_UNKNOWN, PS_SUCCESS, PS_SATURATED, PS_FAIL = 0x0, 0x1, 0x2, 0x3
PS_IMAGE_CORRUPTED, PS_CODEC_UNSUPPORTED, PS_BACKEND_UNSUPPORTED, PS_ENCODING_UNSUPPORTED = 0x7, 0xB, 0x13, 0x23
PS_RESOLUTION_UNSUPPORTED, PS_CODESTREAM_UNSUPPORTED = 0x43, 0x83
PS_COLOR_SPEC_UNSUPPORTED, PS_ORIENTATION_UNSUPPORTED, PS_ROI_UNSUPPORTED, PS_SAMPLING_UNSUPPORTED = 0x5, 0x9, 0x11, 0x21
PS_SAMPLE_TYPE_UNSUPPORTED, PS_SAMPLE_FORMAT_UNSUPPORTED, PS_NUM_PLANES_UNSUPPORTED, PS_NUM_CHANNELS_UNSUPPORTED = 0x41, 0x81, 0x101, 0x201

PRIORITY_HIGHEST, PRIORITY_VERY_HIGH, PRIORITY_HIGH, PRIORITY_NORMAL, PRIORITY_LOW, PRIORITY_VERY_LOW, PRIORITY_LOWEST = 0, 100, 200, 300, 400, 500, 1000

SEVERITY_TRACE, SEVERITY_DEBUG, SEVERITY_INFO, SEVERITY_WARNING, SEVERITY_ERROR, SEVERITY_FATAL = 0x1, 0x10, 0x100, 0x1000, 0x10000, 0x100000
SEVERITY_DEFAULT = SEVERITY_WARNING | SEVERITY_ERROR | SEVERITY_FATAL
CATEGORY_ALL = 0x0FFFFFFF

_HEAD = [("struct_type", C.c_int), ("struct_size", C.c_size_t), ("struct_next", C.c_void_p)]


def _struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": _HEAD + fields})


Properties = _struct("Properties", [("version", C.c_uint32), ("ext_api_version", C.c_uint32), ("cudart_version", C.c_uint32)])
Orientation = _struct("Orientation", [("rotated", C.c_int), ("flip_x", C.c_int), ("flip_y", C.c_int)])
ImagePlaneInfo = _struct("ImagePlaneInfo", [("width", C.c_uint32), ("height", C.c_uint32), ("row_stride", C.c_size_t),
                                            ("num_channels", C.c_uint32), ("sample_type", C.c_int), ("precision", C.c_uint8)])
Region = _struct("Region", [("ndim", C.c_int), ("start", C.c_int * MAX_NUM_DIM), ("end", C.c_int * MAX_NUM_DIM)])
ImageInfo = _struct("ImageInfo", [("codec_name", C.c_char * MAX_CODEC_NAME_SIZE), ("color_spec", C.c_int), ("chroma_subsampling", C.c_int),
                                  ("sample_format", C.c_int), ("orientation", Orientation), ("region", Region), ("num_planes", C.c_uint32),
                                  ("plane_info", ImagePlaneInfo * MAX_NUM_PLANES), ("buffer", C.c_void_p), ("buffer_size", C.c_size_t),
                                  ("buffer_kind", C.c_int), ("cuda_stream", C.c_void_p)])
JpegImageInfo = _struct("JpegImageInfo", [("encoding", C.c_int)])
BackendParams = _struct("BackendParams", [("load_hint", C.c_float)])
Backend = _struct("Backend", [("kind", C.c_int), ("params", BackendParams)])
DecodeParams = _struct("DecodeParams", [("apply_exif_orientation", C.c_int), ("enable_roi", C.c_int)])
EncodeParams = _struct("EncodeParams", [("quality", C.c_float), ("target_psnr", C.c_float)])
JpegEncodeParams = _struct("JpegEncodeParams", [("optimized_huffman", C.c_int)])
DebugMessageData = _struct("DebugMessageData", [("message", C.c_char_p), ("internal_status_id", C.c_uint32), ("codec", C.c_char_p),
                                                ("codec_id", C.c_char_p), ("codec_version", C.c_uint32)])
DebugCallback = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.POINTER(DebugMessageData), C.c_void_p)
DebugMessengerDesc = _struct("DebugMessengerDesc", [("message_severity", C.c_uint32), ("message_category", C.c_uint32),
                                                    ("user_callback", DebugCallback), ("user_data", C.c_void_p)])

TaskFunc = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_void_p)
ExecutorLaunch = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, TaskFunc)
ExecutorGetNumThreads = C.CFUNCTYPE(C.c_int, C.c_void_p)
ExecutorDesc = _struct("ExecutorDesc", [("instance", C.c_void_p), ("launch", ExecutorLaunch), ("getNumThreads", ExecutorGetNumThreads)])

DeviceMalloc = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p)
DeviceFree = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
DeviceAllocator = _struct("DeviceAllocator", [("device_malloc", DeviceMalloc), ("device_free", DeviceFree), ("device_ctx", C.c_void_p),
                                              ("device_mem_padding", C.c_size_t)])
PinnedAllocator = _struct("PinnedAllocator", [("pinned_malloc", DeviceMalloc), ("pinned_free", DeviceFree), ("pinned_ctx", C.c_void_p),
                                              ("pinned_mem_padding", C.c_size_t)])
ExecutionParams = _struct("ExecutionParams", [("device_allocator", C.POINTER(DeviceAllocator)), ("pinned_allocator", C.POINTER(PinnedAllocator)),
                                              ("max_num_cpu_threads", C.c_int), ("executor", C.POINTER(ExecutorDesc)), ("device_id", C.c_int),
                                              ("pre_init", C.c_int), ("num_backends", C.c_int), ("backends", C.POINTER(Backend))])

_io_rw = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t)
IoStreamDesc = _struct("IoStreamDesc", [
    ("instance", C.c_void_p), ("read", _io_rw), ("write", _io_rw),
    ("putc", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_size_t), C.c_ubyte)),
    ("skip", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t)),
    ("seek", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_ssize_t, C.c_int)),
    ("tell", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_ssize_t))),
    ("size", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_size_t))),
    ("reserve", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t)),
    ("flush", C.CFUNCTYPE(C.c_int, C.c_void_p)),
    ("map", C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t)),
    ("unmap", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t))])
GetImageInfoFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(ImageInfo))
ImageReadyFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32)
CodeStreamDesc = _struct("CodeStreamDesc", [("instance", C.c_void_p), ("io_stream", C.POINTER(IoStreamDesc)), ("getImageInfo", GetImageInfoFn)])
ImageDesc = _struct("ImageDesc", [("instance", C.c_void_p), ("getImageInfo", GetImageInfoFn), ("imageReady", ImageReadyFn)])

_PP_CS = C.POINTER(C.POINTER(CodeStreamDesc))
_PP_IM = C.POINTER(C.POINTER(ImageDesc))
DecoderCreateFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(ExecutionParams), C.c_char_p)
DecoderDestroyFn = C.CFUNCTYPE(C.c_int, C.c_void_p)
CanDecodeFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), _PP_CS, _PP_IM, C.c_int, C.POINTER(DecodeParams))
DecodeFn = C.CFUNCTYPE(C.c_int, C.c_void_p, _PP_CS, _PP_IM, C.c_int, C.POINTER(DecodeParams))
DecoderDesc = _struct("DecoderDesc", [("instance", C.c_void_p), ("id", C.c_char_p), ("codec", C.c_char_p), ("backend_kind", C.c_int),
                                      ("create", DecoderCreateFn), ("destroy", DecoderDestroyFn), ("canDecode", CanDecodeFn), ("decode", DecodeFn)])
CanEncodeFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), _PP_IM, _PP_CS, C.c_int, C.POINTER(EncodeParams))
EncodeFn = C.CFUNCTYPE(C.c_int, C.c_void_p, _PP_IM, _PP_CS, C.c_int, C.POINTER(EncodeParams))
EncoderDesc = _struct("EncoderDesc", [("instance", C.c_void_p), ("id", C.c_char_p), ("codec", C.c_char_p), ("backend_kind", C.c_int),
                                      ("create", DecoderCreateFn), ("destroy", DecoderDestroyFn), ("canEncode", CanEncodeFn), ("encode", EncodeFn)])
ParserDesc = _struct("ParserDesc", [("instance", C.c_void_p), ("id", C.c_char_p), ("codec", C.c_char_p), ("canParse", C.c_void_p),
                                    ("create", C.c_void_p), ("destroy", C.c_void_p), ("getImageInfo", C.c_void_p)])

LogFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(DebugMessageData))
RegisterDecoderFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(DecoderDesc), C.c_float)
UnregisterDecoderFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(DecoderDesc))
RegisterEncoderFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(EncoderDesc), C.c_float)
UnregisterEncoderFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(EncoderDesc))
RegisterParserFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(ParserDesc), C.c_float)
UnregisterParserFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(ParserDesc))
FrameworkDesc = _struct("FrameworkDesc", [("instance", C.c_void_p), ("id", C.c_char_p), ("version", C.c_uint32), ("ext_api_version", C.c_uint32),
                                          ("cudart_version", C.c_uint32), ("log", LogFn), ("registerEncoder", RegisterEncoderFn),
                                          ("unregisterEncoder", UnregisterEncoderFn), ("registerDecoder", RegisterDecoderFn),
                                          ("unregisterDecoder", UnregisterDecoderFn), ("registerParser", RegisterParserFn),
                                          ("unregisterParser", UnregisterParserFn)])
ExtensionCreateFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(FrameworkDesc))
ExtensionDestroyFn = C.CFUNCTYPE(C.c_int, C.c_void_p)
ExtensionDesc = _struct("ExtensionDesc", [("instance", C.c_void_p), ("id", C.c_char_p), ("version", C.c_uint32), ("ext_api_version", C.c_uint32),
                                          ("create", ExtensionCreateFn), ("destroy", ExtensionDestroyFn)])
InstanceCreateInfo = _struct("InstanceCreateInfo", [("load_builtin_modules", C.c_int), ("load_extension_modules", C.c_int),
                                                    ("extension_modules_path", C.c_char_p), ("create_debug_messenger", C.c_int),
                                                    ("debug_messenger_desc", C.POINTER(DebugMessengerDesc)), ("message_severity", C.c_uint32),
                                                    ("message_category", C.c_uint32)])
ResizeBufferFn = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

# SURVEY.md Appendix B / static_asserts of include/nvimgcodec_abi.h
EXPECTED_SIZES = {ImagePlaneInfo: 56, Orientation: 40, Region: 72, ImageInfo: 2240, JpegImageInfo: 32, BackendParams: 32, DecodeParams: 32,
                  EncodeParams: 32, JpegEncodeParams: 32, Backend: 64, ExecutorDesc: 48, ExecutionParams: 80, DeviceAllocator: 56,
                  PinnedAllocator: 56, IoStreamDesc: 120, CodeStreamDesc: 48, ImageDesc: 48, ParserDesc: 80, DecoderDesc: 88, EncoderDesc: 88,
                  FrameworkDesc: 112, ExtensionDesc: 64, DebugMessageData: 64, InstanceCreateInfo: 64, Properties: 40}


def init(struct_cls, struct_type, **kw):
    s = struct_cls()
    s.struct_type = struct_type
    s.struct_size = C.sizeof(struct_cls)
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def bind(lib):
    """Declare argtypes/restype of the public API functions `lib` exports (libhipjpeg_host.so: the nvimgcodec* API; libhipjpeg_ext.so:
    nvimgcodecExtensionModuleEntry)."""
    vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
    P = C.POINTER
    sig = {
        "nvimgcodecGetProperties": [P(Properties)],
        "nvimgcodecInstanceCreate": [P(vp), P(InstanceCreateInfo)],
        "nvimgcodecInstanceDestroy": [vp],
        "nvimgcodecExtensionCreate": [vp, P(vp), P(ExtensionDesc)],
        "nvimgcodecExtensionDestroy": [vp],
        "nvimgcodecExtensionModuleEntry": [P(ExtensionDesc)],
        "nvimgcodecDebugMessengerCreate": [vp, P(vp), P(DebugMessengerDesc)],
        "nvimgcodecDebugMessengerDestroy": [vp],
        "nvimgcodecFutureWaitForAll": [vp],
        "nvimgcodecFutureDestroy": [vp],
        "nvimgcodecFutureGetProcessingStatus": [vp, P(C.c_uint32), P(sz)],
        "nvimgcodecImageCreate": [vp, P(vp), P(ImageInfo)],
        "nvimgcodecImageDestroy": [vp],
        "nvimgcodecImageGetImageInfo": [vp, P(ImageInfo)],
        "nvimgcodecCodeStreamCreateFromFile": [vp, P(vp), C.c_char_p],
        "nvimgcodecCodeStreamCreateFromHostMem": [vp, P(vp), vp, sz],
        "nvimgcodecCodeStreamCreateToFile": [vp, P(vp), C.c_char_p, P(ImageInfo)],
        "nvimgcodecCodeStreamCreateToHostMem": [vp, P(vp), vp, ResizeBufferFn, P(ImageInfo)],
        "nvimgcodecCodeStreamDestroy": [vp],
        "nvimgcodecCodeStreamGetImageInfo": [vp, P(ImageInfo)],
        "nvimgcodecDecoderCreate": [vp, P(vp), P(ExecutionParams), C.c_char_p],
        "nvimgcodecDecoderDestroy": [vp],
        "nvimgcodecDecoderCanDecode": [vp, P(vp), P(vp), i, P(DecodeParams), P(C.c_uint32), i],
        "nvimgcodecDecoderDecode": [vp, P(vp), P(vp), i, P(DecodeParams), P(vp)],
        "nvimgcodecEncoderCreate": [vp, P(vp), P(ExecutionParams), C.c_char_p],
        "nvimgcodecEncoderDestroy": [vp],
        "nvimgcodecEncoderCanEncode": [vp, P(vp), P(vp), i, P(EncodeParams), P(C.c_uint32), i],
        "nvimgcodecEncoderEncode": [vp, P(vp), P(vp), i, P(EncodeParams), P(vp)],
    }
    for name, args in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue
        fn.argtypes = args
        fn.restype = C.c_int
    return lib
