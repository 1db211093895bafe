"""ctypes loader for libhipjpeg_ext.so (the HIP extension).  No fallback: if the library is missing the
import error says how to build it, and every device entry point reports HIPJPEG_STATUS_NO_DEVICE without a GPU."""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HIPJPEG_LIB_PATH") or os.path.join(_HERE, "libhipjpeg_ext.so")  # the override is a development aid (A/B builds)

STATUS_NAMES = {0: "SUCCESS", 1: "INVALID_ARGUMENT", 2: "BAD_JPEG", 3: "UNSUPPORTED", 4: "TRUNCATED", 5: "CORRUPT", 6: "ALLOC_FAILED",
                7: "HIP_ERROR", 8: "NO_DEVICE", 9: "BUFFER_TOO_SMALL", 10: "INTERNAL_ERROR"}

OUTPUT_RGBI, OUTPUT_BGRI, OUTPUT_RGB_PLANAR, OUTPUT_BGR_PLANAR, OUTPUT_Y, OUTPUT_YUV_PLANAR = range(6)
FLAG_FANCY_UPSAMPLING = 1
FLAG_GPU_HUFFMAN = 2


class HipJpegError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = int(status)
        super().__init__(f"{what}: HIPJPEG_STATUS_{STATUS_NAMES.get(self.status, self.status)}")


class ImageInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("width", "height", "num_components", "sof_marker", "color_model", "subsampling",
                                              "restart_interval", "num_scans")] + [
        ("h", ctypes.c_int32 * 4), ("v", ctypes.c_int32 * 4), ("blocks_w", ctypes.c_int32 * 4), ("blocks_h", ctypes.c_int32 * 4),
        ("samp_w", ctypes.c_int32 * 4), ("samp_h", ctypes.c_int32 * 4), ("coef_bytes", ctypes.c_uint64)]


class EncodeInput(ctypes.Structure):
    _fields_ = [("plane", ctypes.c_void_p * 3), ("pitch", ctypes.c_uint32 * 3), ("width", ctypes.c_int32), ("height", ctypes.c_int32)]


class EncodeParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("quality", "subsampling", "input_format", "restart_interval", "optimized_huffman", "progressive")]


CSS = {"444": 0, "422": 1, "420": 2, "440": 3, "411": 4, "410": 5, "gray": 6}


class Transform(ctypes.Structure):
    """hipjpegTransform_t: region of interest (stored-image coordinates, end exclusive; all zero = whole image) + EXIF orientation"""
    _fields_ = [("x0", ctypes.c_int32), ("y0", ctypes.c_int32), ("x1", ctypes.c_int32), ("y1", ctypes.c_int32), ("orientation", ctypes.c_int32)]


class Output(ctypes.Structure):
    _fields_ = [("plane", ctypes.c_void_p * 3), ("pitch", ctypes.c_uint32 * 3)]


HOST_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libhipjpeg_host.so")
_lib = None
_host = None


def load_host():
    """libhipjpeg_host.so: the host harness with the application-side nvimgcodec* API (it loads libhipjpeg_ext.so itself, as an extension
    module).  Opened RTLD_LOCAL -- a process that also holds a real libnvimgcodec must not see these names in its global scope."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError(f"{HOST_LIB_PATH} not found: build first (make -C nvimagecodec_amd/csrc)")
        _host = ctypes.CDLL(HOST_LIB_PATH)
    return _host


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
                          "or make -C nvimagecodec_amd/csrc)")
    # torch first: it ships its own libamdhip64; were this library opened before it, the process would hold two HIP runtimes (the system's,
    # through this library, and torch's) and the second to initialise finds no device
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.hipjpegStatusString.restype = ctypes.c_char_p
    L.hipjpegStatusString.argtypes = [i32]
    L.hipjpegVersion.restype = i32
    L.hipjpegTestSetFault.argtypes = [ctypes.c_char_p, i32]
    L.hipjpegGetImageInfo.argtypes = [vp, sz, ctypes.POINTER(ImageInfo)]
    L.hipjpegEntropyDecodeHost.argtypes = [vp, sz, vp, sz, vp, vp]
    L.hipjpegEntropyDecodeGpuAlgorithmHost.argtypes = [vp, sz, vp, sz, vp, ctypes.POINTER(ctypes.c_int32)]
    L.hipjpegCreate.argtypes = [ctypes.POINTER(vp), i32, i32]
    L.hipjpegDestroy.argtypes = [vp]
    L.hipjpegDecodeBatch.argtypes = [vp, vp, vp, i32, vp, i32, ctypes.c_uint, vp, vp]
    L.hipjpegDecodeBatchHost.argtypes = [vp, vp, vp, i32, vp, i32, ctypes.c_uint, vp]
    L.hipjpegDecodeBatchTransfer.argtypes = [vp, vp]
    L.hipjpegDecodeBatchDevice.argtypes = [vp, vp]
    L.hipjpegDecodeBatchDeviceKernel.argtypes = [vp, i32, vp]
    L.hipjpegDecodeBatchStats.argtypes = [vp, vp, vp, vp]
    L.hipjpegDecodeBatchGetStatuses.argtypes = [vp, vp, i32]
    L.hipjpegDecodeBatchSetTransforms.argtypes = [vp, vp, i32]
    L.hipjpegDecodeBatchSubmit.argtypes = [vp, vp, vp, i32, vp, i32, ctypes.c_uint, vp]
    L.hipjpegDecodeBatchWait.argtypes = [vp, vp, i32]
    L.hipjpegSetPipelineDepth.argtypes = [vp, i32]
    L.hipjpegSetHybridHuffmanThreshold.argtypes = [vp, ctypes.c_uint64]
    L.hipjpegDecodeBatchTransferStats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int32)]
    L.hipjpegEntropyDecodeHostSparse.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz), vp]
    L.hipjpegDecodeBatchZeroCopyImages.argtypes = [vp]
    L.hipjpegDecodeBatchZeroCopyImages.restype = i32
    L.hipjpegTestHostFallbacks.argtypes = [vp]
    L.hipjpegTestHostFallbacks.restype = i32
    L.hipjpegTestScanChunkDrops.argtypes = [vp, ctypes.c_size_t, i32, ctypes.POINTER(ctypes.c_uint32), i32]
    L.hipjpegTestScanChunkDrops.restype = i32
    L.hipjpegTestKernelFlavours.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    L.hipjpegDecodeBatchEntropyStats.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint64)]
    L.hipjpegEncodeBatchDevice.argtypes = [vp, vp, vp, i32, vp, vp]
    L.hipjpegEncodeBatchRelaunch.argtypes = [vp, vp]
    L.hipjpegEncodeBatchHost.argtypes = [vp, vp]
    L.hipjpegEncodeBatchEntropy.argtypes = [vp, ctypes.c_uint, vp]
    L.hipjpegEncodeBatchSubmit.argtypes = [vp, vp, vp, i32, ctypes.c_uint, vp]
    L.hipjpegEncodeBatchWait.argtypes = [vp, vp, i32]
    L.hipjpegEncodeBatch.argtypes = [vp, vp, vp, i32, vp, vp]
    L.hipjpegEncodeGetBitstream.argtypes = [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(sz)]
    L.hipjpegEncodeGetCoefficients.argtypes = [vp, i32, i32, ctypes.POINTER(vp), vp]
    L.hipjpegEncodeBatchStats.argtypes = [vp, vp, vp, vp]
    L.hipjpegEncodeBatchGpuEntropyImages.argtypes = [vp]
    L.hipjpegEncodeBatchGpuEntropyImages.restype = i32
    L.hipjpegEncodeFromCoefficientsHost.argtypes = [i32, i32, ctypes.POINTER(EncodeParams), vp, vp, sz, ctypes.POINTER(sz)]
    _lib = L
    return L
