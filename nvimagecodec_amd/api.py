"""Python surface mirroring the reference's `nvimgcodec` module for the JPEG path (python/decoder.cpp:262-403,
python/image.cpp:430-480, python/decode_params.cpp, python/backend.cpp), implemented with ctypes on top of the C API that
libhipjpeg_ext.so exports.  Device memory and streams are torch-ROCm objects; every call goes
   Python -> nvimgcodecDecoderDecode -> priority chain -> hipjpeg_decoder plugin -> HIP kernels.
"""
import ctypes as C
import os
import sys
import enum

import numpy as np

from . import _native
from . import abi as A


class BackendKind(enum.IntEnum):
    CPU_ONLY = A.BACKEND_KIND_CPU_ONLY
    GPU_ONLY = A.BACKEND_KIND_GPU_ONLY
    HYBRID_CPU_GPU = A.BACKEND_KIND_HYBRID_CPU_GPU
    HW_GPU_ONLY = A.BACKEND_KIND_HW_GPU_ONLY


class ColorSpec(enum.IntEnum):
    UNCHANGED = A.COLORSPEC_UNCHANGED
    RGB = A.COLORSPEC_SRGB
    GRAY = A.COLORSPEC_GRAY
    YCC = A.COLORSPEC_SYCC


class ChromaSubsampling(enum.IntEnum):
    CSS_444 = A.SAMPLING_444
    CSS_422 = A.SAMPLING_422
    CSS_420 = A.SAMPLING_420
    CSS_440 = A.SAMPLING_440
    CSS_411 = A.SAMPLING_411
    CSS_410 = A.SAMPLING_410
    CSS_GRAY = A.SAMPLING_GRAY
    CSS_410V = A.SAMPLING_410V


class ImageBufferKind(enum.IntEnum):
    STRIDED_DEVICE = A.BUFFER_KIND_STRIDED_DEVICE
    STRIDED_HOST = A.BUFFER_KIND_STRIDED_HOST


class Backend:
    def __init__(self, backend_kind=BackendKind.HYBRID_CPU_GPU, load_hint=1.0):
        self.backend_kind = BackendKind(backend_kind)
        self.load_hint = float(load_hint)


class DecodeParams:
    def __init__(self, apply_exif_orientation=True, color_spec=ColorSpec.RGB, allow_any_depth=False):
        self.apply_exif_orientation = bool(apply_exif_orientation)
        self.color_spec = ColorSpec(color_spec)
        self.allow_any_depth = bool(allow_any_depth)


class JpegEncodeParams:
    def __init__(self, progressive=False, optimized_huffman=False):
        self.progressive = bool(progressive)
        self.optimized_huffman = bool(optimized_huffman)


class EncodeParams:
    def __init__(self, quality=95, target_psnr=50, color_spec=ColorSpec.UNCHANGED, chroma_subsampling=ChromaSubsampling.CSS_444,
                 jpeg_encode_params=None):
        self.quality = float(quality)
        self.target_psnr = float(target_psnr)
        self.color_spec = ColorSpec(color_spec)
        self.chroma_subsampling = ChromaSubsampling(chroma_subsampling)
        self.jpeg_params = jpeg_encode_params or JpegEncodeParams()


class NvImgCodecError(RuntimeError):
    pass


def _check(st, what):
    if st != A.STATUS_SUCCESS:
        raise NvImgCodecError(f"{what} failed with nvimgcodecStatus_t {st}")


_lib = None
_instance = None
_messenger_cb = None


def _api():
    """Process-wide library + instance (the reference module also owns one instance, python/main.cpp:48-85)."""
    global _lib, _instance
    if _lib is None:
        _lib = A.bind(_native.load_host())
    if _instance is None:
        ci = A.init(A.InstanceCreateInfo, A.ST_INSTANCE_CREATE_INFO, load_builtin_modules=1, load_extension_modules=1,
                    create_debug_messenger=1, message_severity=A.SEVERITY_ERROR | A.SEVERITY_FATAL, message_category=A.CATEGORY_ALL)
        inst = C.c_void_p()
        _check(_lib.nvimgcodecInstanceCreate(C.byref(inst), C.byref(ci)), "nvimgcodecInstanceCreate")
        _instance = inst
    return _lib, _instance


class Image:
    """Decoded (or to-be-encoded) image: a uint8 HxWxC tensor on the GPU (torch) or on the host (numpy)."""

    def __init__(self, array, buffer_kind):
        self._array = array
        self.buffer_kind = ImageBufferKind(buffer_kind)

    @property
    def shape(self):
        return tuple(self._array.shape)

    @property
    def height(self):
        return self.shape[0]

    @property
    def width(self):
        return self.shape[1]

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def dtype(self):
        d = self._array.dtype
        return np.dtype(d) if isinstance(d, np.dtype) else np.dtype(str(d).replace("torch.", ""))

    @property
    def precision(self):
        return 0

    @property
    def __cuda_array_interface__(self):
        if self.buffer_kind != ImageBufferKind.STRIDED_DEVICE:
            raise AttributeError("host image has no __cuda_array_interface__")
        return self._array.__cuda_array_interface__

    @property
    def __array_interface__(self):
        if self.buffer_kind != ImageBufferKind.STRIDED_HOST:
            raise AttributeError("device image has no __array_interface__")
        return self._array.__array_interface__

    def __dlpack__(self, stream=None):
        return self._array.__dlpack__() if stream is None else self._array.__dlpack__(stream=stream)

    def __dlpack_device__(self):
        return self._array.__dlpack_device__()

    def to_dlpack(self, cuda_stream=None):
        return self.__dlpack__()

    def as_tensor(self):
        """The underlying torch tensor (device images) -- zero copy."""
        return self._array

    def cpu(self):
        if self.buffer_kind == ImageBufferKind.STRIDED_HOST:
            return self
        return Image(self._array.cpu().numpy(), ImageBufferKind.STRIDED_HOST)

    def cuda(self, synchronize=True):
        if self.buffer_kind == ImageBufferKind.STRIDED_DEVICE:
            return self
        import torch
        return Image(torch.from_numpy(np.ascontiguousarray(self._array)).cuda(), ImageBufferKind.STRIDED_DEVICE)


def _is_dlpack_capsule(obj):
    return type(obj).__name__ == "PyCapsule"


def _from_dlpack_object(source):
    """DLPack import (python/dlpack_utils.cpp): a PyCapsule named "dltensor", or any object with __dlpack__ /
    __dlpack_device__.  Zero copy: the tensor torch builds on top of the capsule keeps the producer's memory alive."""
    import torch
    t = torch.utils.dlpack.from_dlpack(source) if _is_dlpack_capsule(source) else torch.from_dlpack(source)
    if t.is_cuda:
        return Image(t, ImageBufferKind.STRIDED_DEVICE)
    return Image(t.numpy(), ImageBufferKind.STRIDED_HOST)


def as_image(source, cuda_stream=0):
    """Wrap an external buffer as an Image and tie its lifetime to the Image (python/module.cpp:91-105): a DLPack capsule, or an
    object with __cuda_array_interface__ / __array_interface__ / __dlpack__ + __dlpack_device__ (torch tensors, numpy arrays)."""
    import torch
    if isinstance(source, Image):
        return source
    if isinstance(source, torch.Tensor):
        return Image(source, ImageBufferKind.STRIDED_DEVICE) if source.is_cuda else Image(source.numpy(), ImageBufferKind.STRIDED_HOST)
    if isinstance(source, np.ndarray):
        return Image(source, ImageBufferKind.STRIDED_HOST)
    if hasattr(source, "__cuda_array_interface__"):
        return Image(torch.as_tensor(source, device="cuda"), ImageBufferKind.STRIDED_DEVICE)
    if _is_dlpack_capsule(source) or (hasattr(source, "__dlpack__") and hasattr(source, "__dlpack_device__")):
        return _from_dlpack_object(source)
    if hasattr(source, "__array_interface__"):
        return Image(np.asarray(source), ImageBufferKind.STRIDED_HOST)
    raise TypeError("unsupported image source")


def from_dlpack(source, cuda_stream=0):
    """Zero-copy conversion from a DLPack tensor to an Image (python/module.cpp:134-150): `source` is a PyCapsule holding a
    DLPack tensor, or an (array) object with __dlpack__ and __dlpack_device__."""
    if not (_is_dlpack_capsule(source) or (hasattr(source, "__dlpack__") and hasattr(source, "__dlpack_device__"))):
        raise TypeError("from_dlpack needs a DLPack capsule or an object with __dlpack__ and __dlpack_device__")
    return _from_dlpack_object(source)


def as_images(sources, cuda_stream=0):
    return [as_image(s, cuda_stream) for s in sources]


def _fill_image_info(info, h, w, channels, sample_format, color_spec, buffer, pitch, buffer_kind, stream, planar=False, subsampling=A.SAMPLING_444):
    info.struct_type = A.ST_IMAGE_INFO
    info.struct_size = C.sizeof(A.ImageInfo)
    info.sample_format = sample_format
    info.color_spec = color_spec
    info.chroma_subsampling = subsampling
    planes = channels if planar else 1
    info.num_planes = planes
    for p in range(planes):
        pi = info.plane_info[p]
        pi.struct_type = A.ST_IMAGE_PLANE_INFO
        pi.struct_size = C.sizeof(A.ImagePlaneInfo)
        pi.width, pi.height = w, h
        pi.row_stride = pitch
        pi.num_channels = 1 if planar else channels
        pi.sample_type = A.SAMPLE_DATA_TYPE_UINT8
        pi.precision = 0
    info.buffer = buffer
    info.buffer_size = pitch * h * planes
    info.buffer_kind = buffer_kind
    info.cuda_stream = stream
    info.orientation.struct_type = A.ST_ORIENTATION
    info.orientation.struct_size = C.sizeof(A.Orientation)
    info.region.struct_type = A.ST_REGION
    info.region.struct_size = C.sizeof(A.Region)


class _ExecMixin:
    def _make_exec_params(self, device_id, max_num_cpu_threads, backends):
        self._backends_arr = None
        ep = A.init(A.ExecutionParams, A.ST_EXECUTION_PARAMS, max_num_cpu_threads=int(max_num_cpu_threads), device_id=int(device_id))
        if backends:
            arr = (A.Backend * len(backends))()
            for i, b in enumerate(backends):
                arr[i].struct_type = A.ST_BACKEND
                arr[i].struct_size = C.sizeof(A.Backend)
                arr[i].kind = int(b.backend_kind)
                arr[i].params.struct_type = A.ST_BACKEND_PARAMS
                arr[i].params.struct_size = C.sizeof(A.BackendParams)
                arr[i].params.load_hint = b.load_hint
            self._backends_arr = arr
            ep.num_backends = len(backends)
            ep.backends = C.cast(arr, C.POINTER(A.Backend))
        return ep


class Decoder(_ExecMixin):
    """nvimgcodec.Decoder (python/decoder.cpp:262-300).  Note one deliberate difference in defaults: the reference's Python
    layer passes options=":fancy_upsampling=0" (decoder.cpp:283); here the default is the plugins' own default (fancy on),
    which is the setting under which pixels are bit-exact against the libjpeg-turbo CPU path."""

    def __init__(self, device_id=A.DEVICE_CURRENT, max_num_cpu_threads=0, backends=None, options=""):
        import torch
        self._torch = torch
        lib, inst = _api()
        if device_id == A.DEVICE_CURRENT:
            device_id = torch.cuda.current_device() if torch.cuda.is_available() else A.DEVICE_CPU_ONLY
        self.device_id = device_id
        ep = self._make_exec_params(device_id, max_num_cpu_threads, backends)
        self._h = C.c_void_p()
        _check(lib.nvimgcodecDecoderCreate(inst, C.byref(self._h), C.byref(ep), options.encode()), "nvimgcodecDecoderCreate")

    def close(self):
        if getattr(self, "_h", None):
            _lib.nvimgcodecDecoderDestroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def read(self, paths, params=None, cuda_stream=0):
        single = isinstance(paths, (str, bytes))
        plist = [paths] if single else list(paths)
        res = self._decode_sources([("file", p) for p in plist], params, cuda_stream)
        return res[0] if single else res

    def decode(self, data, params=None, cuda_stream=0):
        single = isinstance(data, (bytes, bytearray, memoryview, np.ndarray))
        dlist = [data] if single else list(data)
        res = self._decode_sources([("mem", d) for d in dlist], params, cuda_stream)
        return res[0] if single else res

    def _decode_sources(self, sources, params, cuda_stream):
        import time as _t
        _dbg = os.environ.get("HIPJPEG_DEBUG_TIMING")
        _t0 = _t.perf_counter()
        torch = self._torch
        lib, inst = _api()
        params = params or DecodeParams()
        n = len(sources)
        keep, streams, images, outs = [], [], [], []
        dev = torch.device("cuda", self.device_id)
        for kind, src in sources:
            cs = C.c_void_p()
            if kind == "file":
                st = lib.nvimgcodecCodeStreamCreateFromFile(inst, C.byref(cs), str(src).encode())
            else:
                arr = np.frombuffer(bytes(src), dtype=np.uint8) if not isinstance(src, np.ndarray) else np.ascontiguousarray(src, dtype=np.uint8)
                keep.append(arr)
                st = lib.nvimgcodecCodeStreamCreateFromHostMem(inst, C.byref(cs), arr.ctypes.data, arr.size)
            if st != A.STATUS_SUCCESS:
                streams.append(None)
                images.append(None)
                outs.append(None)
                continue
            info = A.init(A.ImageInfo, A.ST_IMAGE_INFO)
            _check(lib.nvimgcodecCodeStreamGetImageInfo(cs, C.byref(info)), "nvimgcodecCodeStreamGetImageInfo")
            h, w = info.plane_info[0].height, info.plane_info[0].width
            # python/decoder.cpp:202-205: a quarter-turn orientation swaps the output's width and height
            if params.apply_exif_orientation and (info.orientation.rotated // 90) % 2:
                h, w = w, h
            # python/decoder.cpp:179-225: interleaved RGB u8 (or gray), row_stride = w * channels, device buffer
            gray = params.color_spec == ColorSpec.GRAY or (params.color_spec == ColorSpec.UNCHANGED and info.num_planes == 1)
            ch = 1 if gray else 3
            t = torch.empty((h, w, ch), dtype=torch.uint8, device=dev)
            out_info = A.ImageInfo()
            _fill_image_info(out_info, h, w, ch, A.SAMPLEFORMAT_P_Y if gray else A.SAMPLEFORMAT_I_RGB,
                             A.COLORSPEC_GRAY if gray else A.COLORSPEC_SRGB, t.data_ptr(), t.stride(0), A.BUFFER_KIND_STRIDED_DEVICE,
                             cuda_stream or torch.cuda.current_stream(self.device_id).cuda_stream)
            out_info.orientation = info.orientation
            im = C.c_void_p()
            _check(lib.nvimgcodecImageCreate(inst, C.byref(im), C.byref(out_info)), "nvimgcodecImageCreate")
            streams.append(cs)
            images.append(im)
            outs.append(t)
        valid = [i for i in range(n) if streams[i] is not None]
        _t1 = _t.perf_counter()
        results = [None] * n
        if valid:
            cs_arr = (C.c_void_p * len(valid))(*[streams[i] for i in valid])
            im_arr = (C.c_void_p * len(valid))(*[images[i] for i in valid])
            dp = A.init(A.DecodeParams, A.ST_DECODE_PARAMS, apply_exif_orientation=int(params.apply_exif_orientation), enable_roi=0)
            fut = C.c_void_p()
            _check(lib.nvimgcodecDecoderDecode(self._h, cs_arr, im_arr, len(valid), C.byref(dp), C.byref(fut)), "nvimgcodecDecoderDecode")
            _t2 = _t.perf_counter()
            _check(lib.nvimgcodecFutureWaitForAll(fut), "nvimgcodecFutureWaitForAll")
            _t3 = _t.perf_counter()
            st = (C.c_uint32 * len(valid))()
            size = C.c_size_t()
            lib.nvimgcodecFutureGetProcessingStatus(fut, st, C.byref(size))
            lib.nvimgcodecFutureDestroy(fut)
            # failed samples are dropped from the result list like the reference does (python/decoder.cpp:230-242): None here
            for k, i in enumerate(valid):
                if st[k] == A.PS_SUCCESS:
                    results[i] = Image(outs[i], ImageBufferKind.STRIDED_DEVICE)
        for i in valid:
            lib.nvimgcodecImageDestroy(images[i])
            lib.nvimgcodecCodeStreamDestroy(streams[i])
        if _dbg and valid:
            _t4 = _t.perf_counter()
            print("[api] setup %.2f ms, DecoderDecode %.2f ms, wait %.2f ms, teardown %.2f ms" % ((_t1 - _t0) * 1e3, (_t2 - _t1) * 1e3, (_t3 - _t2) * 1e3, (_t4 - _t3) * 1e3), file=sys.stderr)
        return results


class MultiDeviceDecoder:
    """One process, several devices: a `Decoder` per entry of `device_ids` -- the reference keys its worker
    pools by device the same way (src/default_executor.cpp:45-58) -- and a host thread per entry.  `decode(batch)` partitions the batch
    over the devices' queues by size (sharding.shard_indices: greedy longest-processing-time over coefficient + bitstream bytes, visited
    in decreasing size like src/image_generic_decoder.cpp:134-178 sorts a batch), every queue decodes on its own device, and the results
    come back in input order, each Image on the device that decoded it.  No device talks to another (no collective).  An id may appear
    more than once: several queues on one card."""

    def __init__(self, device_ids, max_num_cpu_threads=0, backends=None, options=""):
        self.device_ids = [int(d) for d in device_ids]
        if not self.device_ids:
            raise ValueError("device_ids must name at least one device")
        self._decoders = [Decoder(d, max_num_cpu_threads, backends, options) for d in self.device_ids]

    def close(self):
        for d in self._decoders:
            d.close()
        self._decoders = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def shard(self, data):
        """Index lists, one per device queue (every index exactly once)."""
        from . import sharding
        return sharding.shard_indices([sharding.image_cost(d) for d in data], len(self._decoders))

    def decode(self, data, params=None):
        import threading
        import torch
        dlist = list(data)
        queues = self.shard(dlist)
        results = [None] * len(dlist)
        errors = []

        def run(q):
            try:
                dec = self._decoders[q]
                with torch.cuda.device(dec.device_id):
                    if queues[q]:
                        for i, im in zip(queues[q], dec.decode([dlist[i] for i in queues[q]], params)):
                            results[i] = im
                        torch.cuda.synchronize(dec.device_id)
            except Exception as e:  # noqa: BLE001 -- handed to the caller below
                errors.append(e)

        threads = [threading.Thread(target=run, args=(q,)) for q in range(len(self._decoders))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return results


class Encoder(_ExecMixin):
    """nvimgcodec.Encoder (python/encoder.cpp:136-179, 292-390) for JPEG output."""

    def __init__(self, device_id=A.DEVICE_CURRENT, max_num_cpu_threads=0, backends=None, options=""):
        import torch
        self._torch = torch
        lib, inst = _api()
        if device_id == A.DEVICE_CURRENT:
            device_id = torch.cuda.current_device() if torch.cuda.is_available() else A.DEVICE_CPU_ONLY
        self.device_id = device_id
        ep = self._make_exec_params(device_id, max_num_cpu_threads, backends)
        self._h = C.c_void_p()
        _check(lib.nvimgcodecEncoderCreate(inst, C.byref(self._h), C.byref(ep), options.encode()), "nvimgcodecEncoderCreate")

    def close(self):
        if getattr(self, "_h", None):
            _lib.nvimgcodecEncoderDestroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def write(self, file_names, images, codec="jpeg", params=None, cuda_stream=0):
        single = isinstance(file_names, (str, bytes))
        names = [file_names] if single else list(file_names)
        imgs = [images] if single else list(images)
        data = self.encode(imgs, codec, params, cuda_stream)
        out = []
        for n, d in zip(names, data):
            if d is not None:
                with open(n, "wb") as f:
                    f.write(d)
                out.append(n)
            else:
                out.append(None)
        return out[0] if single else out

    def encode(self, images, codec="jpeg", params=None, cuda_stream=0):
        if codec.lstrip(".").lower() not in ("jpeg", "jpg"):
            raise NvImgCodecError(f"codec {codec!r} is outside this extension (JPEG only)")
        torch = self._torch
        lib, inst = _api()
        params = params or EncodeParams()
        single = not isinstance(images, (list, tuple))
        ilist = [as_image(i) for i in ([images] if single else images)]
        n = len(ilist)
        sinks, streams, handles, keep = [], [], [], []
        stream_ptr = cuda_stream or torch.cuda.current_stream(self.device_id).cuda_stream
        import time as _t
        _dbg = os.environ.get("HIPJPEG_DEBUG_TIMING")
        _t0 = _t.perf_counter()

        for im in ilist:
            arr = im._array
            if im.buffer_kind == ImageBufferKind.STRIDED_HOST:
                arr = np.ascontiguousarray(arr)
            h, w = arr.shape[0], arr.shape[1]
            ch = arr.shape[2] if arr.ndim == 3 else 1
            gray = ch == 1
            if im.buffer_kind == ImageBufferKind.STRIDED_DEVICE:
                ptr, pitch, kind = arr.data_ptr(), arr.stride(0), A.BUFFER_KIND_STRIDED_DEVICE
            else:
                ptr, pitch, kind = arr.ctypes.data, arr.strides[0], A.BUFFER_KIND_STRIDED_HOST
            keep.append(arr)
            css = A.SAMPLING_GRAY if gray else int(params.chroma_subsampling)
            info = A.ImageInfo()
            _fill_image_info(info, h, w, ch, A.SAMPLEFORMAT_P_Y if gray else A.SAMPLEFORMAT_I_RGB,
                             A.COLORSPEC_GRAY if gray else A.COLORSPEC_SRGB, ptr, pitch, kind, stream_ptr, subsampling=css)
            ih = C.c_void_p()
            _check(lib.nvimgcodecImageCreate(inst, C.byref(ih), C.byref(info)), "nvimgcodecImageCreate")
            # output stream description: codec name, target subsampling, JPEG encoding (baseline / progressive)
            out_info = A.ImageInfo()
            _fill_image_info(out_info, h, w, ch, info.sample_format, info.color_spec, None, 0, A.BUFFER_KIND_STRIDED_HOST, None, subsampling=css)
            out_info.codec_name = b"jpeg"
            ji = A.init(A.JpegImageInfo, A.ST_JPEG_IMAGE_INFO, encoding=A.JPEG_ENCODING_PROGRESSIVE_DCT_HUFFMAN if params.jpeg_params.progressive
                        else A.JPEG_ENCODING_BASELINE_DCT)
            out_info.struct_next = C.addressof(ji)
            sink = {"buf": None}

            def resize(ctx, size, sink=sink):
                if sink["buf"] is None or size > len(sink["buf"]):
                    nb = C.create_string_buffer(size)
                    if sink["buf"] is not None:
                        C.memmove(nb, sink["buf"], len(sink["buf"]))
                    sink["buf"] = nb
                sink["size"] = size
                return C.addressof(sink["buf"])

            cb = A.ResizeBufferFn(resize)
            cs = C.c_void_p()
            _check(lib.nvimgcodecCodeStreamCreateToHostMem(inst, C.byref(cs), None, cb, C.byref(out_info)), "nvimgcodecCodeStreamCreateToHostMem")
            keep.append((cb, ji))
            sinks.append(sink)
            streams.append(cs)
            handles.append(ih)

        ep = A.init(A.EncodeParams, A.ST_ENCODE_PARAMS, quality=params.quality, target_psnr=params.target_psnr)
        jp = A.init(A.JpegEncodeParams, A.ST_JPEG_ENCODE_PARAMS, optimized_huffman=int(params.jpeg_params.optimized_huffman))
        ep.struct_next = C.addressof(jp)
        fut = C.c_void_p()
        _t1 = _t.perf_counter()
        _check(lib.nvimgcodecEncoderEncode(self._h, (C.c_void_p * n)(*handles), (C.c_void_p * n)(*streams), n, C.byref(ep), C.byref(fut)),
               "nvimgcodecEncoderEncode")
        _t2 = _t.perf_counter()
        _check(lib.nvimgcodecFutureWaitForAll(fut), "nvimgcodecFutureWaitForAll")
        _t3 = _t.perf_counter()
        st = (C.c_uint32 * n)()
        size = C.c_size_t()
        lib.nvimgcodecFutureGetProcessingStatus(fut, st, C.byref(size))
        lib.nvimgcodecFutureDestroy(fut)
        res = []
        for i in range(n):
            ok = st[i] == A.PS_SUCCESS and sinks[i]["buf"] is not None
            res.append(C.string_at(sinks[i]["buf"], sinks[i]["size"]) if ok else None)
            lib.nvimgcodecImageDestroy(handles[i])
            lib.nvimgcodecCodeStreamDestroy(streams[i])
        if _dbg:
            _t4 = _t.perf_counter()
            print("[api] encode setup %.2f ms, EncoderEncode %.2f ms, wait %.2f ms, collect %.2f ms" % ((_t1 - _t0) * 1e3, (_t2 - _t1) * 1e3, (_t3 - _t2) * 1e3, (_t4 - _t3) * 1e3), file=sys.stderr)
        return res[0] if single else res
