"""Thin Python front-end of the hipjpeg C-ABI (include/hipjpeg.h).  torch is used only as the owner of device
memory and streams; every pixel is produced by the HIP kernels in libhipjpeg_ext.so."""
import ctypes

import numpy as np

from . import _native as N

_FORMATS = {"rgb": N.OUTPUT_RGBI, "bgr": N.OUTPUT_BGRI, "rgb_planar": N.OUTPUT_RGB_PLANAR, "bgr_planar": N.OUTPUT_BGR_PLANAR,
            "y": N.OUTPUT_Y, "yuv_planar": N.OUTPUT_YUV_PLANAR}


def _as_u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    if hasattr(data, "data_ptr") and hasattr(data, "numpy"):  # a torch CPU tensor: its memory, no copy
        return data.numpy()
    return np.frombuffer(bytes(data), dtype=np.uint8)


def entropy_decode_host_sparse(data):
    """The host entropy stage's zero-run-compressed stream of a picture, expanded again: per component int16 [blocks_h, blocks_w, 64] in
    NATURAL (row-major) order like entropy_decode_host + the stream's size in bytes.  Raises HipJpegError(UNSUPPORTED) for frames the format
    does not cover (progressive, several scans)."""
    a = _as_u8(data)
    info = get_image_info(data)
    nblocks = [bh * bw for bh, bw in zip(info["blocks_h"], info["blocks_w"])]
    cap = sum(nblocks) * 200 + 64
    buf = np.zeros(cap, dtype=np.uint8)
    n = ctypes.c_size_t()
    toff = (ctypes.c_uint64 * 4)()
    st = N.load().hipjpegEntropyDecodeHostSparse(a.ctypes.data, a.size, buf.ctypes.data, cap, ctypes.byref(n), toff)
    if st:
        raise N.HipJpegError(st, "hipjpegEntropyDecodeHostSparse")
    tables = buf[: sum(nblocks) * 4].view(np.uint32)
    out = []
    for c, nb in enumerate(nblocks):
        blocks = np.zeros((nb, 64), dtype=np.int16)
        for b in range(nb):
            off = int(tables[int(toff[c]) + b])
            if off == 0:
                continue
            k = int(buf[off])
            blocks[b, 0] = np.frombuffer(buf[off + 1:off + 3].tobytes(), dtype="<i2")[0]
            for e in range(k):
                r = off + 3 + 3 * e
                blocks[b, int(buf[r])] = np.frombuffer(buf[r + 1:r + 3].tobytes(), dtype="<i2")[0]
        blk = blocks.reshape(info["blocks_h"][c], info["blocks_w"][c], 8, 8)  # device layout: [column][row]
        out.append(np.ascontiguousarray(blk.transpose(0, 1, 3, 2)).reshape(info["blocks_h"][c], info["blocks_w"][c], 64))
    return out, int(n.value)


def get_image_info(data):
    a = _as_u8(data)
    info = N.ImageInfo()
    st = N.load().hipjpegGetImageInfo(a.ctypes.data, a.size, ctypes.byref(info))
    if st:
        raise N.HipJpegError(st, "hipjpegGetImageInfo")
    nc = info.num_components
    d = {k: getattr(info, k) for k in ("width", "height", "num_components", "sof_marker", "color_model", "subsampling",
                                        "restart_interval", "num_scans", "coef_bytes")}
    for k in ("h", "v", "blocks_w", "blocks_h", "samp_w", "samp_h"):
        d[k] = list(getattr(info, k))[:nc]
    return d


def entropy_decode_host(data):
    """Host stage only (no GPU).  Returns (coefs, qtables): per component int16 [blocks_h, blocks_w, 64] and uint16[64],
    both converted back to NATURAL (row-major) order for easy comparison with the oracle."""
    a = _as_u8(data)
    info = get_image_info(a)
    buf = np.zeros(info["coef_bytes"] // 2, dtype=np.int16)
    offs = (ctypes.c_uint64 * 4)()
    qt = np.zeros(256, dtype=np.uint16)
    st = N.load().hipjpegEntropyDecodeHost(a.ctypes.data, a.size, buf.ctypes.data, buf.nbytes, ctypes.addressof(offs), qt.ctypes.data)
    if st:
        raise N.HipJpegError(st, "hipjpegEntropyDecodeHost")
    coefs, qts = [], []
    for c in range(info["num_components"]):
        n = info["blocks_w"][c] * info["blocks_h"][c]
        blk = buf[offs[c]: offs[c] + n * 64].reshape(info["blocks_h"][c], info["blocks_w"][c], 8, 8)
        coefs.append(np.ascontiguousarray(blk.transpose(0, 1, 3, 2)).reshape(info["blocks_h"][c], info["blocks_w"][c], 64))
        qts.append(np.ascontiguousarray(qt[c * 64:(c + 1) * 64].reshape(8, 8).T).reshape(64))
    return coefs, qts


def entropy_decode_gpu_algorithm_host(data):
    """The GPU entropy decoder's multi-pass algorithm emulated on the host (no GPU).  Returns (coefs natural order, sync passes)."""
    a = _as_u8(data)
    info = get_image_info(a)
    buf = np.zeros(info["coef_bytes"] // 2, dtype=np.int16)
    offs = (ctypes.c_uint64 * 4)()
    passes = ctypes.c_int32()
    st = N.load().hipjpegEntropyDecodeGpuAlgorithmHost(a.ctypes.data, a.size, buf.ctypes.data, buf.nbytes, ctypes.addressof(offs), ctypes.byref(passes))
    if st:
        raise N.HipJpegError(st, "hipjpegEntropyDecodeGpuAlgorithmHost")
    coefs = []
    for c in range(info["num_components"]):
        n = info["blocks_w"][c] * info["blocks_h"][c]
        blk = buf[offs[c]: offs[c] + n * 64].reshape(info["blocks_h"][c], info["blocks_w"][c], 8, 8)
        coefs.append(np.ascontiguousarray(blk.transpose(0, 1, 3, 2)).reshape(info["blocks_h"][c], info["blocks_w"][c], 64))
    return coefs, passes.value


class BatchDecoder:
    """hipjpegCreate / hipjpegDecodeBatch* on one device."""

    def __init__(self, device=0, num_threads=0):
        import torch
        self._torch = torch
        self.device = int(device)
        self._h = ctypes.c_void_p()
        st = N.load().hipjpegCreate(ctypes.byref(self._h), self.device, int(num_threads))
        if st:
            raise N.HipJpegError(st, "hipjpegCreate")
        self._keep = None
        self._inflight = []

    def close(self):
        if self._h:
            N.load().hipjpegDestroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- output allocation mirrors python/decoder.cpp:179-225 of the reference: I_RGB u8, row_stride = w*3
    def allocate_outputs(self, jpegs, fmt="rgb", transforms=None):
        """transforms: optional list with one entry per image, None or (roi, orientation) with roi = (x0, y0, x1, y1) or
        None and orientation = EXIF 1..8; the output then has the size of the region, turned upright."""
        torch = self._torch
        dev = torch.device("cuda", self.device)
        outs = []
        for i, j in enumerate(jpegs):
            try:
                info = get_image_info(j)
            except N.HipJpegError:
                outs.append(None)
                continue
            h, w = info["height"], info["width"]
            if transforms is not None and transforms[i] is not None:
                roi, orientation = transforms[i]
                if roi is not None:
                    w, h = roi[2] - roi[0], roi[3] - roi[1]
                if orientation >= 5:
                    w, h = h, w
            if fmt in ("rgb", "bgr"):
                outs.append(torch.empty((h, w, 3), dtype=torch.uint8, device=dev))
            elif fmt in ("rgb_planar", "bgr_planar"):
                outs.append(torch.empty((3, h, w), dtype=torch.uint8, device=dev))
            elif fmt == "y":
                outs.append(torch.empty((h, w), dtype=torch.uint8, device=dev))
            else:
                outs.append([torch.empty((info["samp_h"][c], info["samp_w"][c]), dtype=torch.uint8, device=dev)
                             for c in range(info["num_components"])])
        return outs

    def _marshal(self, jpegs, outs, fmt):
        n = len(jpegs)
        # a torch CPU tensor (uint8, contiguous) is taken where it lies -- in pinned memory (tensor.pin_memory()) the library then sends its
        # bitstream to the device without a staging copy (zero-copy input); everything else goes through numpy
        arrs = [j if (hasattr(j, "data_ptr") and hasattr(j, "numel")) else _as_u8(j) for j in jpegs]
        ptrs = (ctypes.c_void_p * n)(*[(a.data_ptr() if hasattr(a, "data_ptr") else a.ctypes.data) for a in arrs])
        lens = (ctypes.c_size_t * n)(*[(a.numel() if hasattr(a, "numel") else a.size) for a in arrs])
        O = (N.Output * n)()
        for i, o in enumerate(outs):
            if o is None:
                continue
            if fmt in ("rgb", "bgr"):
                O[i].plane[0] = o.data_ptr()
                O[i].pitch[0] = o.stride(0)
            elif fmt in ("rgb_planar", "bgr_planar"):
                for p in range(3):
                    O[i].plane[p] = o[p].data_ptr()
                    O[i].pitch[p] = o.stride(1)
            elif fmt == "y":
                O[i].plane[0] = o.data_ptr()
                O[i].pitch[0] = o.stride(0)
            else:
                for p, t in enumerate(o):
                    O[i].plane[p] = t.data_ptr()
                    O[i].pitch[p] = t.stride(0)
        statuses = (ctypes.c_int * n)()
        self._keep = (arrs, ptrs, lens, O, outs)  # keep host inputs alive until the next call
        return ptrs, lens, O, statuses

    def _stream_ptr(self, stream):
        torch = self._torch
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        return ctypes.c_void_p(s.cuda_stream)

    def set_transforms(self, transforms, n):
        """Geometry for the next batch (see allocate_outputs); None clears it."""
        if transforms is None:
            st = N.load().hipjpegDecodeBatchSetTransforms(self._h, None, 0)
        else:
            T = (N.Transform * n)()
            for i, t in enumerate(transforms):
                if t is None:
                    T[i].orientation = 1
                    continue
                roi, orientation = t
                if roi is not None:
                    T[i].x0, T[i].y0, T[i].x1, T[i].y1 = [int(v) for v in roi]
                T[i].orientation = int(orientation)
            st = N.load().hipjpegDecodeBatchSetTransforms(self._h, T, n)
        if st:
            raise N.HipJpegError(st, "hipjpegDecodeBatchSetTransforms")

    def decode(self, jpegs, fmt="rgb", fancy=True, outs=None, stream=None, check=True, gpu_huffman=False, transforms=None):
        """Full pipeline.  Returns (outputs, statuses).  gpu_huffman=True: entropy-decode eligible streams on the GPU.
        transforms: per-image (roi, orientation) or None -- region of interest and EXIF orientation applied on the device."""
        if outs is None:
            outs = self.allocate_outputs(jpegs, fmt, transforms)
        if transforms is not None:
            self.set_transforms(transforms, len(jpegs))
        ptrs, lens, O, statuses = self._marshal(jpegs, outs, fmt)
        flags = (N.FLAG_FANCY_UPSAMPLING if fancy else 0) | (N.FLAG_GPU_HUFFMAN if gpu_huffman else 0)
        st = N.load().hipjpegDecodeBatch(self._h, ptrs, lens, len(jpegs), O, _FORMATS[fmt], flags, statuses, self._stream_ptr(stream))
        if st:
            raise N.HipJpegError(st, "hipjpegDecodeBatch")
        statuses = list(statuses)
        if check:
            for i, s in enumerate(statuses):
                if s:
                    raise N.HipJpegError(s, f"image {i}")
        return outs, statuses

    # -- pipelined: submit() returns once everything is queued; wait() returns the statuses of the oldest submitted batch.
    #    At most three batches in flight (set_pipeline_depth: up to eight).  The caller keeps jpegs and outs alive until the
    #    matching wait().
    def host_fallbacks(self):
        """Images of the last settled batch that the GPU entropy stage handed back to the host entropy decoder."""
        return int(N.load().hipjpegTestHostFallbacks(self._h))

    def kernel_flavours(self):
        """(plane_units[1], luma_units[3]) of the current batch: work units of K1 and of K2 per layout (hipjpegTestKernelFlavours)."""
        import ctypes
        a, b = (ctypes.c_int32 * 1)(), (ctypes.c_int32 * 3)()
        st = N.load().hipjpegTestKernelFlavours(self._h, a, b)
        if st:
            raise N.HipJpegError(st, "hipjpegTestKernelFlavours")
        return list(a), list(b)

    def set_hybrid_huffman_threshold(self, pixels):
        """gpu_huffman=True then applies to images of more than `pixels` pixels only (nvJPEG's hybrid_huffman_threshold); 0 = all."""
        st = N.load().hipjpegSetHybridHuffmanThreshold(self._h, ctypes.c_uint64(int(pixels)))
        if st:
            raise N.HipJpegError(st, "hipjpegSetHybridHuffmanThreshold")

    def fused_units(self):
        """Work units of the current batch that went to the FUSED kernel builds (HIPJPEG_FUSED_DECODE=1)."""
        return int(N.load().hipjpegTestFusedUnits(self._h))

    def set_pipeline_depth(self, depth):
        st = N.load().hipjpegSetPipelineDepth(self._h, int(depth))
        if st:
            raise N.HipJpegError(st, "hipjpegSetPipelineDepth")

    def submit(self, jpegs, outs, fmt="rgb", fancy=True, stream=None, gpu_huffman=True):
        ptrs, lens, O, statuses = self._marshal(jpegs, outs, fmt)
        flags = (N.FLAG_FANCY_UPSAMPLING if fancy else 0) | (N.FLAG_GPU_HUFFMAN if gpu_huffman else 0)
        st = N.load().hipjpegDecodeBatchSubmit(self._h, ptrs, lens, len(jpegs), O, _FORMATS[fmt], flags, self._stream_ptr(stream))
        if st:
            raise N.HipJpegError(st, "hipjpegDecodeBatchSubmit")
        self._inflight.append((len(jpegs), (ptrs, lens, O, jpegs, outs)))

    def wait(self, check=True):
        n, _keep = self._inflight.pop(0)
        statuses = (ctypes.c_int32 * n)()
        st = N.load().hipjpegDecodeBatchWait(self._h, statuses, n)
        if st:
            raise N.HipJpegError(st, "hipjpegDecodeBatchWait")
        statuses = list(statuses)
        if check:
            for i, s in enumerate(statuses):
                if s:
                    raise N.HipJpegError(s, f"image {i}")
        return statuses

    # -- the three phases separately (bench.py times device_stage with coefficients resident in HBM)
    def host_stage(self, jpegs, outs, fmt="rgb", fancy=True, gpu_huffman=False):
        ptrs, lens, O, statuses = self._marshal(jpegs, outs, fmt)
        flags = (N.FLAG_FANCY_UPSAMPLING if fancy else 0) | (N.FLAG_GPU_HUFFMAN if gpu_huffman else 0)
        st = N.load().hipjpegDecodeBatchHost(self._h, ptrs, lens, len(jpegs), O, _FORMATS[fmt], flags, statuses)
        if st:
            raise N.HipJpegError(st, "hipjpegDecodeBatchHost")
        return list(statuses)

    def transfer(self, stream=None):
        st = N.load().hipjpegDecodeBatchTransfer(self._h, self._stream_ptr(stream))
        if st:
            raise N.HipJpegError(st, "hipjpegDecodeBatchTransfer")

    def device_stage(self, stream=None, which=None):
        """which=None: all kernels; 0 idct_plane, 1 luma_color, 2 generic_color."""
        if which is None:
            st = N.load().hipjpegDecodeBatchDevice(self._h, self._stream_ptr(stream))
        else:
            st = N.load().hipjpegDecodeBatchDeviceKernel(self._h, int(which), self._stream_ptr(stream))
        if st:
            raise N.HipJpegError(st, "hipjpegDecodeBatchDevice")

    def stats(self):
        units = (ctypes.c_int32 * 3)()
        cb, ob = ctypes.c_uint64(), ctypes.c_uint64()
        N.load().hipjpegDecodeBatchStats(self._h, ctypes.addressof(units), ctypes.byref(cb), ctypes.byref(ob))
        gi, sl, sb = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_uint64()
        N.load().hipjpegDecodeBatchEntropyStats(self._h, ctypes.byref(gi), ctypes.byref(sl), ctypes.byref(sb))
        h2d, sp = ctypes.c_uint64(), ctypes.c_int32()
        N.load().hipjpegDecodeBatchTransferStats(self._h, ctypes.byref(h2d), ctypes.byref(sp))
        return dict(units=list(units), coef_bytes=cb.value, output_bytes=ob.value, gpu_entropy_images=gi.value, sync_launches=sl.value,
                    stream_bytes=sb.value, zero_copy_images=int(N.load().hipjpegDecodeBatchZeroCopyImages(self._h)), h2d_bytes=h2d.value,
                    sparse_images=sp.value)

    def statuses(self, n):
        st = (ctypes.c_int * n)()
        N.load().hipjpegDecodeBatchGetStatuses(self._h, st, n)
        return list(st)


# ---------------------------------------------------------------------------------------------- encode
_ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42,
                    49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
_IN_FORMATS = {"rgb": N.OUTPUT_RGBI, "bgr": N.OUTPUT_BGRI, "rgb_planar": N.OUTPUT_RGB_PLANAR, "bgr_planar": N.OUTPUT_BGR_PLANAR, "gray": N.OUTPUT_Y,
               "yuv_planar": N.OUTPUT_YUV_PLANAR}


def _enc_params(subsampling, quality, input_format="rgb", restart_interval=0, optimized_huffman=False, progressive=False):
    return N.EncodeParams(int(quality), N.CSS[subsampling], _IN_FORMATS[input_format], int(restart_interval), int(bool(optimized_huffman)),
                          int(bool(progressive)))


def encode_from_coefficients_host(width, height, coefs_natural, subsampling="420", quality=90, restart_interval=0, optimized_huffman=False,
                                  progressive=False):
    """Host-only entropy coding (no GPU).  coefs_natural: per component int16 [blocks_h, blocks_w, 64] in natural order over the
    MCU-padded grid (what oracle.forward returns); converted to the zigzag layout the C-ABI takes."""
    zz = [np.ascontiguousarray(c[:, :, _ZIGZAG]) for c in coefs_natural]
    ptrs = (ctypes.c_void_p * 3)(*([z.ctypes.data for z in zz] + [None] * (3 - len(zz))))
    p = _enc_params(subsampling, quality, "rgb", restart_interval, optimized_huffman, progressive)
    cap = width * height * 3 + 65536
    out = np.zeros(cap, dtype=np.uint8)
    n = ctypes.c_size_t()
    st = N.load().hipjpegEncodeFromCoefficientsHost(width, height, ctypes.byref(p), ptrs, out.ctypes.data, cap, ctypes.byref(n))
    if st:
        raise N.HipJpegError(st, "hipjpegEncodeFromCoefficientsHost")
    return out[: n.value].tobytes()


class BatchEncoder:
    """hipjpegEncodeBatch* on one device; inputs are torch CUDA uint8 tensors ([H, W, 3] interleaved, [3, H, W] planar or [H, W] gray)."""

    def __init__(self, device=0, num_threads=0, gpu_huffman=False):
        import torch
        self._torch = torch
        self.device = int(device)
        self.gpu_huffman = bool(gpu_huffman)
        self._inflight = []
        self._h = ctypes.c_void_p()
        st = N.load().hipjpegCreate(ctypes.byref(self._h), self.device, int(num_threads))
        if st:
            raise N.HipJpegError(st, "hipjpegCreate")
        self._keep = None
        self._n = 0

    def close(self):
        if self._h:
            N.load().hipjpegDestroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream_ptr(self, stream):
        s = stream if stream is not None else self._torch.cuda.current_stream(self.device)
        return ctypes.c_void_p(s.cuda_stream)

    def _marshal(self, images, subsampling, quality, input_format, restart_interval, optimized_huffman, progressive=False):
        n = len(images)
        I = (N.EncodeInput * n)()
        P = (N.EncodeParams * n)()
        subs = subsampling if isinstance(subsampling, (list, tuple)) else [subsampling] * n
        quals = quality if isinstance(quality, (list, tuple)) else [quality] * n
        for i, t in enumerate(images):
            fmt = input_format
            if fmt in ("rgb", "bgr"):
                h, w = t.shape[0], t.shape[1]
                I[i].plane[0] = t.data_ptr()
                I[i].pitch[0] = t.stride(0)
            elif fmt == "gray":
                h, w = t.shape
                I[i].plane[0] = t.data_ptr()
                I[i].pitch[0] = t.stride(0)
            elif fmt == "yuv_planar":  # t = [Y, Cb, Cr], the chroma planes already in the stream's sampling
                h, w = t[0].shape
                for p in range(3):
                    I[i].plane[p] = t[p].data_ptr()
                    I[i].pitch[p] = t[p].stride(0)
            else:
                h, w = t.shape[1], t.shape[2]
                for p in range(3):
                    I[i].plane[p] = t[p].data_ptr()
                    I[i].pitch[p] = t.stride(1)
            I[i].width, I[i].height = w, h
            P[i] = _enc_params(subs[i], quals[i], fmt, restart_interval, optimized_huffman, progressive)
        self._keep = (images, I, P)
        self._n = n
        return I, P

    def device_stage(self, images, subsampling="420", quality=90, input_format="rgb", restart_interval=0, optimized_huffman=False, stream=None,
                     progressive=False):
        I, P = self._marshal(images, subsampling, quality, input_format, restart_interval, optimized_huffman, progressive)
        st_arr = (ctypes.c_int * self._n)()
        st = N.load().hipjpegEncodeBatchDevice(self._h, I, P, self._n, st_arr, self._stream_ptr(stream))
        if st:
            raise N.HipJpegError(st, "hipjpegEncodeBatchDevice")
        return list(st_arr)

    # -- pipelined: submit() queues forward kernel + entropy stage + copy of the files and returns; wait() completes the
    #    oldest submitted batch and returns (statuses, bitstreams).  At most three batches in flight.
    def submit(self, images, subsampling="420", quality=90, input_format="rgb", restart_interval=0, optimized_huffman=False, stream=None,
               gpu_huffman=None, progressive=False):
        if gpu_huffman is None:
            gpu_huffman = self.gpu_huffman
        I, P = self._marshal(images, subsampling, quality, input_format, restart_interval, optimized_huffman, progressive)
        st = N.load().hipjpegEncodeBatchSubmit(self._h, I, P, self._n, N.FLAG_GPU_HUFFMAN if gpu_huffman else 0, self._stream_ptr(stream))
        if st:
            raise N.HipJpegError(st, "hipjpegEncodeBatchSubmit")
        self._inflight.append((self._n, self._keep))

    def wait(self, fetch=True):
        n, _keep = self._inflight.pop(0)
        st_arr = (ctypes.c_int * n)()
        st = N.load().hipjpegEncodeBatchWait(self._h, st_arr, n)
        if st:
            raise N.HipJpegError(st, "hipjpegEncodeBatchWait")
        self._n = n
        return list(st_arr), (self.bitstreams() if fetch else None)

    def relaunch(self, stream=None):
        st = N.load().hipjpegEncodeBatchRelaunch(self._h, self._stream_ptr(stream))
        if st:
            raise N.HipJpegError(st, "hipjpegEncodeBatchRelaunch")

    def host_stage(self, gpu_huffman=None):
        """Entropy stage of the prepared batch.  gpu_huffman (default: the encoder's setting): code on the GPU what it can
        take (Annex-K tables, no restart markers); the rest, or everything when False, on the host thread pool."""
        if gpu_huffman is None:
            gpu_huffman = self.gpu_huffman
        st_arr = (ctypes.c_int * self._n)()
        st = N.load().hipjpegEncodeBatchEntropy(self._h, N.FLAG_GPU_HUFFMAN if gpu_huffman else 0, st_arr)
        if st:
            raise N.HipJpegError(st, "hipjpegEncodeBatchEntropy")
        return list(st_arr)

    def bitstreams(self):
        out = []
        for i in range(self._n):
            p, n = ctypes.c_void_p(), ctypes.c_size_t()
            st = N.load().hipjpegEncodeGetBitstream(self._h, i, ctypes.byref(p), ctypes.byref(n))
            out.append(ctypes.string_at(p, n.value) if st == 0 else None)
        return out

    def coefficients(self, index):
        """Quantized coefficients of image `index`, per component int16 [real_h, real_w, 64] in NATURAL order."""
        res = []
        c = 0
        while True:
            p = ctypes.c_void_p()
            grid = (ctypes.c_int32 * 4)()
            st = N.load().hipjpegEncodeGetCoefficients(self._h, index, c, ctypes.byref(p), ctypes.addressof(grid))
            if st:
                break
            bw, bh, rw, rh = list(grid)
            a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_int16)), shape=(bh, bw, 64))
            nat = np.zeros((rh, rw, 64), dtype=np.int16)
            nat[:, :, _ZIGZAG] = a[:rh, :rw, :]
            res.append(nat)
            c += 1
        return res

    def encode(self, images, subsampling="420", quality=90, input_format="rgb", restart_interval=0, optimized_huffman=False, stream=None,
               progressive=False):
        st = self.device_stage(images, subsampling, quality, input_format, restart_interval, optimized_huffman, stream, progressive)
        st = self.host_stage()
        for i, s in enumerate(st):
            if s:
                raise N.HipJpegError(s, f"image {i}")
        return self.bitstreams()

    def stats(self):
        u = ctypes.c_int32()
        pb, cb = ctypes.c_uint64(), ctypes.c_uint64()
        N.load().hipjpegEncodeBatchStats(self._h, ctypes.byref(u), ctypes.byref(pb), ctypes.byref(cb))
        return dict(units=u.value, pixel_bytes=pb.value, coef_bytes=cb.value, gpu_entropy_images=int(N.load().hipjpegEncodeBatchGpuEntropyImages(self._h)))
