"""GPU tests of the drop-in boundary: the same pixels, but reached the way an nvImageCodec application reaches them --
nvimgcodecInstanceCreate -> code streams -> images -> nvimgcodecDecoderDecode -> priority chain -> hipjpeg_decoder plugin
(function table of include/nvimgcodec_abi.h).  Bit-exact against the golden vectors / oracle."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, load_decode_case
from fake_plugin import FakeDecoderPlugin
from nvimagecodec_amd import _native
from nvimagecodec_amd import abi as A

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    _M = json.load(_f)


def _case(name):
    return load_decode_case(next(e for e in _M["decode"] if e["name"] == name))


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_python_decoder_mirror_decodes_goldens(torch_mod):
    from nvimagecodec_amd import api
    names = ["s50x37_420_base_q90", "s33x65_422_prog_q50", "s64x48_444_base_q90", "r130x70_420_prog_rst7", "c1_640x480_444_base_q90",
             "c2_1920x1080_420_base_q90", "o64x48_411_base_q90"]
    cases = [_case(n) for n in names]
    with api.Decoder(max_num_cpu_threads=4) as dec:
        imgs = dec.decode([c[0] for c in cases])
        torch_mod.cuda.synchronize()
        for n, (jpeg, rgb), im in zip(names, cases, imgs):
            assert im is not None, n
            assert im.buffer_kind == api.ImageBufferKind.STRIDED_DEVICE and im.dtype == np.uint8
            got = im.cpu()
            ref = rgb if rgb is not None else oracle.decode(jpeg)
            assert got.shape == ref.shape
            assert np.array_equal(np.asarray(got._array), ref), n
            assert im.__cuda_array_interface__["shape"] == ref.shape
        # single bytes object -> single Image; gray request -> P_Y
        one = dec.decode(cases[0][0], params=api.DecodeParams(color_spec=api.ColorSpec.GRAY))
        torch_mod.cuda.synchronize()
        assert np.array_equal(one.cpu()._array[:, :, 0], oracle.decode(cases[0][0], oracle.FMT_GRAY))
        # garbage in a batch: that entry is None, neighbours fine
        res = dec.decode([cases[1][0], b"garbage", cases[2][0]])
        assert res[1] is None and res[0] is not None and res[2] is not None


def test_read_from_file_uses_read_fallback(torch_mod, tmp_path):
    """File streams return NULL from map() (reference src/io_stream.h:170) so the plugin must read() the bitstream."""
    from nvimagecodec_amd import api
    jpeg, rgb = _case("r130x70_444_base_rst1")
    p = tmp_path / "x.jpg"
    p.write_bytes(jpeg)
    with api.Decoder() as dec:
        im = dec.read(str(p))
        torch_mod.cuda.synchronize()
        assert np.array_equal(im.cpu()._array, rgb)


def _c_api_decode(lib, inst, dec, jpeg, h, w, fmt, planes, channels, buffer_ptr, pitch, kind, stream=0, params=None):
    arr = np.frombuffer(jpeg, dtype=np.uint8)
    cs = C.c_void_p()
    assert lib.nvimgcodecCodeStreamCreateFromHostMem(inst, C.byref(cs), arr.ctypes.data, arr.size) == 0
    info = A.init(A.ImageInfo, A.ST_IMAGE_INFO, sample_format=fmt, color_spec=A.COLORSPEC_SRGB, num_planes=planes, buffer=buffer_ptr,
                  buffer_size=pitch * h * planes, buffer_kind=kind, cuda_stream=stream)
    for p in range(planes):
        pi = info.plane_info[p]
        pi.width, pi.height, pi.row_stride, pi.num_channels, pi.sample_type = w, h, pitch, channels, A.SAMPLE_DATA_TYPE_UINT8
    im = C.c_void_p()
    assert lib.nvimgcodecImageCreate(inst, C.byref(im), C.byref(info)) == 0
    dp = params or A.init(A.DecodeParams, A.ST_DECODE_PARAMS)
    fut = C.c_void_p()
    assert lib.nvimgcodecDecoderDecode(dec, (C.c_void_p * 1)(cs), (C.c_void_p * 1)(im), 1, C.byref(dp), C.byref(fut)) == 0
    assert lib.nvimgcodecFutureWaitForAll(fut) == 0
    st = (C.c_uint32 * 1)()
    n = C.c_size_t()
    lib.nvimgcodecFutureGetProcessingStatus(fut, st, C.byref(n))
    lib.nvimgcodecFutureDestroy(fut)
    lib.nvimgcodecImageDestroy(im)
    lib.nvimgcodecCodeStreamDestroy(cs)
    return st[0]


def _setup(lib, extra_plugins=(), device_id=0, backends=None, allocators=None, options=b"", messenger=None):
    ci = A.init(A.InstanceCreateInfo, A.ST_INSTANCE_CREATE_INFO, load_builtin_modules=1, load_extension_modules=1)
    inst = C.c_void_p()
    assert lib.nvimgcodecInstanceCreate(C.byref(inst), C.byref(ci)) == 0
    if messenger is not None:
        dm = C.c_void_p()
        assert lib.nvimgcodecDebugMessengerCreate(inst, C.byref(dm), C.byref(messenger)) == 0
    for p in extra_plugins:
        assert lib.nvimgcodecExtensionCreate(inst, None, C.byref(p.ext_desc)) == 0
    ep = A.init(A.ExecutionParams, A.ST_EXECUTION_PARAMS, device_id=device_id, max_num_cpu_threads=2)
    if allocators:
        ep.device_allocator, ep.pinned_allocator = C.pointer(allocators[0]), C.pointer(allocators[1])
    dec = C.c_void_p()
    assert lib.nvimgcodecDecoderCreate(inst, C.byref(dec), C.byref(ep), options) == 0
    return inst, dec


def test_host_output_buffer_is_bounced(torch_mod):
    """A host output buffer with a GPU backend: the framework decodes into a device bounce buffer and copies back
    (reference src/work.h:144-190)."""
    lib = A.bind(_native.load_host())
    inst, dec = _setup(lib)
    jpeg, rgb = _case("s50x37_420_base_q90")
    buf = np.zeros((37, 50, 3), dtype=np.uint8)
    st = _c_api_decode(lib, inst, dec, jpeg, 37, 50, A.SAMPLEFORMAT_I_RGB, 1, 3, buf.ctypes.data, 150, A.BUFFER_KIND_STRIDED_HOST)
    assert st == A.PS_SUCCESS
    assert np.array_equal(buf, rgb)
    lib.nvimgcodecDecoderDestroy(dec)
    lib.nvimgcodecInstanceDestroy(inst)


def test_planar_bgr_and_user_stream(torch_mod):
    torch = torch_mod
    lib = A.bind(_native.load_host())
    inst, dec = _setup(lib)
    jpeg, rgb = _case("r130x70_420_base_rst7")
    side = torch.cuda.Stream()
    out = torch.zeros((3, 70, 130), dtype=torch.uint8, device="cuda")
    st = _c_api_decode(lib, inst, dec, jpeg, 70, 130, A.SAMPLEFORMAT_P_BGR, 3, 1, out.data_ptr(), 130, A.BUFFER_KIND_STRIDED_DEVICE,
                       stream=side.cuda_stream)
    assert st == A.PS_SUCCESS
    # the plugin made `side` wait for the decode (cuda_decoder.cpp:552-556): work queued on it afterwards sees the pixels
    with torch.cuda.stream(side):
        copy = out.clone()
    side.synchronize()
    assert np.array_equal(copy.cpu().numpy(), rgb[:, :, ::-1].transpose(2, 0, 1))
    lib.nvimgcodecDecoderDestroy(dec)
    lib.nvimgcodecInstanceDestroy(inst)


def _decode_region(lib, inst, dec, jpeg, region, shape):
    """nvimgcodecDecoderDecode of one stream into a host buffer of `shape` with image_info.region = (y0, x0, y1, x1)"""
    arr = np.frombuffer(jpeg, dtype=np.uint8)
    cs = C.c_void_p()
    assert lib.nvimgcodecCodeStreamCreateFromHostMem(inst, C.byref(cs), arr.ctypes.data, arr.size) == 0
    buf = np.zeros(shape, dtype=np.uint8)
    info = A.init(A.ImageInfo, A.ST_IMAGE_INFO, sample_format=A.SAMPLEFORMAT_I_RGB, color_spec=A.COLORSPEC_SRGB, num_planes=1,
                  buffer=buf.ctypes.data, buffer_size=buf.nbytes, buffer_kind=A.BUFFER_KIND_STRIDED_HOST)
    pi = info.plane_info[0]
    pi.width, pi.height, pi.row_stride, pi.num_channels, pi.sample_type = shape[1], shape[0], shape[1] * 3, 3, A.SAMPLE_DATA_TYPE_UINT8
    info.region.ndim = 2
    info.region.start[0], info.region.start[1], info.region.end[0], info.region.end[1] = region
    im = C.c_void_p()
    assert lib.nvimgcodecImageCreate(inst, C.byref(im), C.byref(info)) == 0
    dp = A.init(A.DecodeParams, A.ST_DECODE_PARAMS, enable_roi=1)
    fut = C.c_void_p()
    assert lib.nvimgcodecDecoderDecode(dec, (C.c_void_p * 1)(cs), (C.c_void_p * 1)(im), 1, C.byref(dp), C.byref(fut)) == 0
    lib.nvimgcodecFutureWaitForAll(fut)
    st = (C.c_uint32 * 1)()
    n = C.c_size_t()
    lib.nvimgcodecFutureGetProcessingStatus(fut, st, C.byref(n))
    lib.nvimgcodecFutureDestroy(fut)
    lib.nvimgcodecImageDestroy(im)
    lib.nvimgcodecCodeStreamDestroy(cs)
    return st[0], buf


def test_region_of_interest_on_the_device_and_fallback_for_what_is_not_supported(torch_mod):
    """A region inside the image is decoded by the HIP decoder (the pixels of the full decode, like the reference CPU path's
    crop, extensions/libjpeg_turbo/jpeg_mem.cpp:206-240).  A region that leaves the image is outside it: canDecode says so
    and the chain moves on, exactly like nvjpeg -> libjpeg_turbo in the reference."""
    lib = A.bind(_native.load_host())
    cpu = FakeDecoderPlugin("cpu_fallback", priority=A.PRIORITY_NORMAL, fill=0x42)
    inst, dec = _setup(lib, extra_plugins=[cpu])
    jpeg, rgb = _case("s64x48_420_base_q90")
    st, buf = _decode_region(lib, inst, dec, jpeg, (8, 16, 32, 48), (24, 32, 3))
    assert st == A.PS_SUCCESS and cpu.count("decode") == 0
    assert np.array_equal(buf, rgb[8:32, 16:48])
    st, buf = _decode_region(lib, inst, dec, jpeg, (8, 16, 32, 80), (24, 64, 3))  # x1 = 80 > width 64
    assert st == A.PS_SUCCESS and buf.flat[0] == 0x42 and cpu.count("decode") == 1
    lib.nvimgcodecDecoderDestroy(dec)
    lib.nvimgcodecInstanceDestroy(inst)


def _with_exif_orientation(jpeg, orientation):
    # APP1 Exif, little-endian TIFF, IFD0 with one entry: Orientation (0x0112) SHORT
    tiff = b"II*\x00\x08\x00\x00\x00" + b"\x01\x00" + b"\x12\x01\x03\x00\x01\x00\x00\x00" + bytes([orientation]) + b"\x00\x00\x00" + b"\x00\x00\x00\x00"
    payload = b"Exif\x00\x00" + tiff
    return jpeg[:2] + b"\xff\xe1" + (len(payload) + 2).to_bytes(2, "big") + payload + jpeg[2:]


def test_exif_orientation_is_applied_like_the_reference_python_decoder(torch_mod):
    """python/decoder.cpp:202-205 sizes the output for the upright picture when apply_exif_orientation is set; the pixels
    are the stored picture turned according to the EXIF tag (all eight values)."""
    from nvimagecodec_amd import api
    from test_gpu_geometry import upright
    jpeg, rgb = _case("s50x37_420_base_q90")
    with api.Decoder(max_num_cpu_threads=2) as dec:
        imgs = dec.decode([_with_exif_orientation(jpeg, o) for o in range(1, 9)])
        torch_mod.cuda.synchronize()
        for o, im in zip(range(1, 9), imgs):
            assert im is not None, o
            assert np.array_equal(np.asarray(im.cpu()._array), upright(rgb, o)), o
        # apply_exif_orientation=False: as stored
        im = dec.decode(_with_exif_orientation(jpeg, 6), params=api.DecodeParams(apply_exif_orientation=False))
        torch_mod.cuda.synchronize()
        assert np.array_equal(np.asarray(im.cpu()._array), rgb)


def test_cpu_only_device_is_refused_and_custom_allocators_are_used(torch_mod):
    torch = torch_mod
    lib = A.bind(_native.load_host())
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipHostFree.argtypes = [C.c_void_p]
    calls = {"dev": 0, "pin": 0, "devfree": 0, "pinfree": 0}

    def dmalloc(ctx, pp, size, stream):
        calls["dev"] += 1
        return hip.hipMalloc(pp, size)

    def dfree(ctx, p, size, stream):
        calls["devfree"] += 1
        return hip.hipFree(p)

    def pmalloc(ctx, pp, size, stream):
        calls["pin"] += 1
        return hip.hipHostMalloc(pp, size, 0)

    def pfree(ctx, p, size, stream):
        calls["pinfree"] += 1
        return hip.hipHostFree(p)

    cbs = [A.DeviceMalloc(dmalloc), A.DeviceFree(dfree), A.DeviceMalloc(pmalloc), A.DeviceFree(pfree)]
    da = A.init(A.DeviceAllocator, A.ST_DEVICE_ALLOCATOR, device_malloc=cbs[0], device_free=cbs[1])
    pa = A.init(A.PinnedAllocator, A.ST_PINNED_ALLOCATOR, pinned_malloc=cbs[2], pinned_free=cbs[3])
    inst, dec = _setup(lib, allocators=(da, pa))
    jpeg, rgb = _case("s64x48_444_base_q90")
    out = torch.zeros((48, 64, 3), dtype=torch.uint8, device="cuda")
    st = _c_api_decode(lib, inst, dec, jpeg, 48, 64, A.SAMPLEFORMAT_I_RGB, 1, 3, out.data_ptr(), 192, A.BUFFER_KIND_STRIDED_DEVICE)
    torch.cuda.synchronize()
    assert st == A.PS_SUCCESS and np.array_equal(out.cpu().numpy(), rgb)
    assert calls["dev"] >= 1 and calls["pin"] >= 1
    lib.nvimgcodecDecoderDestroy(dec)
    assert calls["devfree"] == calls["dev"] and calls["pinfree"] == calls["pin"]
    lib.nvimgcodecInstanceDestroy(inst)
    # device_id = CPU_ONLY: a GPU plugin must refuse to be created (cuda_decoder.cpp:272-273); nothing else registered -> no decoder
    inst, dec = _setup(lib, device_id=A.DEVICE_CPU_ONLY)
    buf = np.zeros((48, 64, 3), dtype=np.uint8)
    st = _c_api_decode(lib, inst, dec, jpeg, 48, 64, A.SAMPLEFORMAT_I_RGB, 1, 3, buf.ctypes.data, 192, A.BUFFER_KIND_STRIDED_HOST)
    assert st != A.PS_SUCCESS
    lib.nvimgcodecDecoderDestroy(dec)
    lib.nvimgcodecInstanceDestroy(inst)


def test_many_batches_back_to_back_reuse_pages(torch_mod):
    """Alternating batch pages + executor threads: 6 consecutive batches of 24 mixed images stay bit-exact."""
    from nvimagecodec_amd import api
    entries = [e for e in _M["decode"] if e["pixels"]][:24]
    cases = [load_decode_case(e) for e in entries]
    with api.Decoder(max_num_cpu_threads=6) as dec:
        for rep in range(6):
            imgs = dec.decode([c[0] for c in cases])
            torch_mod.cuda.synchronize()
            for e, c, im in zip(entries, cases, imgs):
                assert np.array_equal(im.cpu()._array, c[1]), (rep, e["name"])


@pytest.mark.parametrize("options,count", [("hipjpeg_decoder:pipeline_chunks=3", 7), ("", 200), ("hipjpeg_decoder:pipeline_chunks=1", 100)])
def test_large_batches_are_cut_into_pipelined_pieces(torch_mod, options, count):
    """The plugin cuts a large batch into up to three jobs (own page, own stream) -- by size, or as the option says.  Whatever
    the cut: every sample reports once, results keep their order, a corrupt file in the middle fails alone."""
    from nvimagecodec_amd import api
    entries = [e for e in _M["decode"] if e["pixels"]][:20]
    cases = [load_decode_case(e) for e in entries]
    jpegs = [cases[i % len(cases)][0] for i in range(count)]
    bad = count // 2
    broken = bytearray(jpegs[bad])
    broken[len(broken) // 2:] = b""   # truncated: not even an EOI
    jpegs[bad] = bytes(broken[:200])
    with api.Decoder(max_num_cpu_threads=6, options=options) as dec:
        for rep in range(2):
            imgs = dec.decode(jpegs)
            torch_mod.cuda.synchronize()
            assert len(imgs) == count
            for i, im in enumerate(imgs):
                if i == bad:
                    assert im is None
                else:
                    assert im is not None and np.array_equal(im.cpu()._array, cases[i % len(cases)][1]), (rep, i)


def test_python_encoder_mirror_through_plugin(torch_mod):
    """Encoder.encode -> nvimgcodecEncoderEncode -> hipjpeg_encoder plugin -> HIP kernel + host Huffman -> host-memory code stream.
    Bitstream must equal the oracle's (== libjpeg-turbo's scan for these settings)."""
    from nvimagecodec_amd import api
    from nvimagecodec_amd.synth import synth_image
    torch = torch_mod
    imgs = [synth_image(w, h, seed=w) for (w, h) in ((64, 48), (131, 77), (640, 480))]
    dev = [torch.from_numpy(i).cuda() for i in imgs]
    with api.Encoder(max_num_cpu_threads=4) as enc:
        for css, sub in ((api.ChromaSubsampling.CSS_420, "420"), (api.ChromaSubsampling.CSS_444, "444"), (api.ChromaSubsampling.CSS_422, "422")):
            out = enc.encode([api.as_image(d) for d in dev], "jpeg", api.EncodeParams(quality=90, chroma_subsampling=css))
            for im, b in zip(imgs, out):
                assert b == oracle.encode(im, sub, 90), sub
        # host-memory input goes through the framework's H2D bounce (reference src/work.h:192-232)
        b = enc.encode(api.as_image(imgs[1]), ".jpg", api.EncodeParams(quality=75, chroma_subsampling=api.ChromaSubsampling.CSS_420))
        assert b == oracle.encode(imgs[1], "420", 75)
        # gray image -> single-component stream
        g = np.ascontiguousarray(imgs[0][:, :, :1])
        bg = enc.encode(api.as_image(torch.from_numpy(g).cuda()), "jpeg", api.EncodeParams(quality=90))
        assert bg == oracle.encode(np.repeat(g, 3, axis=2), "gray", 90)
        # optimized Huffman tables via the chained nvimgcodecJpegEncodeParams_t
        bo = enc.encode(api.as_image(dev[2]), "jpeg", api.EncodeParams(quality=90, chroma_subsampling=api.ChromaSubsampling.CSS_420,
                                                                       jpeg_encode_params=api.JpegEncodeParams(optimized_huffman=True)))
        ref = oracle.encode(imgs[2], "420", 90)
        assert len(bo) < len(ref)
        assert all(np.array_equal(a, c) for a, c in zip(oracle.decode_coefficients(bo)[0], oracle.decode_coefficients(ref)[0]))
        # progressive output (nvimgcodecJpegImageInfo_t::encoding = PROGRESSIVE_DCT_HUFFMAN, cuda_encoder.cpp:339-346): an SOF2 file with
        # the coefficients of the baseline one; byte parity with libjpeg-turbo is in test_gpu_encode.py
        bp = enc.encode(api.as_image(dev[0]), "jpeg", api.EncodeParams(quality=90, chroma_subsampling=api.ChromaSubsampling.CSS_420,
                                                                       jpeg_encode_params=api.JpegEncodeParams(progressive=True)))
        assert bp is not None and b"\xff\xc2" in bp[:700] and b"\xff\xc0" not in bp[:700]
        assert all(np.array_equal(a, c) for a, c in zip(oracle.decode_coefficients(bp)[0],
                                                        oracle.decode_coefficients(oracle.encode(imgs[0], "420", 90))[0]))


def test_transcode_roundtrip_through_both_plugins(torch_mod, tmp_path):
    """decode -> encode -> decode through the nvImageCodec API route (the nvimtrans use case, example/nvimtrans/main.cpp)."""
    from nvimagecodec_amd import api
    jpeg, rgb = _case("c1_640x480_444_base_q90")
    with api.Decoder() as dec, api.Encoder() as enc:
        img = dec.decode(jpeg)
        path = tmp_path / "out.jpg"
        assert enc.write(str(path), img, "jpeg", api.EncodeParams(quality=95, chroma_subsampling=api.ChromaSubsampling.CSS_420)) == str(path)
        again = dec.read(str(path))
        torch_mod.cuda.synchronize()
        ref_rgb = oracle.decode(jpeg)
        assert np.array_equal(again.cpu()._array, oracle.decode(oracle.encode(ref_rgb, "420", 95)))


def test_planar_ycbcr_image_through_the_encoder_plugin(torch_mod):
    """NVIMGCODEC_SAMPLEFORMAT_P_YUV + SYCC through nvimgcodecEncoderEncode (reference extensions/nvjpeg/cuda_encoder.cpp:104-115,
    362-368): three planes at component size in one strided device buffer.  Same subsampling in and out -> the planes go into the
    stream as they are and the file equals libjpeg-turbo's for the RGB picture the planes were made from; a different output
    subsampling, or planes with an RGB colour spec, is refused by canEncode (no other encoder registered -> no file)."""
    from nvimagecodec_amd import api
    from nvimagecodec_amd.api import _api, _check, _fill_image_info
    from nvimagecodec_amd.synth import synth_image
    from test_gpu_encode import _planes_like_libjpeg
    torch = torch_mod
    lib, inst = _api()
    rgb = synth_image(160, 96, seed=11)
    h, w = rgb.shape[:2]

    def encode(planes, in_css, out_css, color_spec, progressive=False):
        # one device buffer, planes back to back, each with its own pitch = its width
        flat = torch.cat([torch.from_numpy(np.ascontiguousarray(p)).reshape(-1) for p in planes]).cuda()
        info = A.ImageInfo()
        _fill_image_info(info, h, w, 3, A.SAMPLEFORMAT_P_YUV, color_spec, flat.data_ptr(), w, A.BUFFER_KIND_STRIDED_DEVICE,
                         torch.cuda.current_stream().cuda_stream, planar=True, subsampling=in_css)
        for p, plane in enumerate(planes):
            info.plane_info[p].height, info.plane_info[p].width = plane.shape
            info.plane_info[p].row_stride = plane.shape[1]
        info.buffer_size = flat.numel()
        image = C.c_void_p()
        _check(lib.nvimgcodecImageCreate(inst, C.byref(image), C.byref(info)), "nvimgcodecImageCreate")
        out_info = A.ImageInfo()
        _fill_image_info(out_info, h, w, 3, A.SAMPLEFORMAT_P_YUV, color_spec, None, 0, A.BUFFER_KIND_STRIDED_HOST, None, subsampling=out_css)
        out_info.codec_name = b"jpeg"
        ji = A.init(A.JpegImageInfo, A.ST_JPEG_IMAGE_INFO,
                    encoding=A.JPEG_ENCODING_PROGRESSIVE_DCT_HUFFMAN if progressive else A.JPEG_ENCODING_BASELINE_DCT)
        out_info.struct_next = C.addressof(ji)
        sink = {"buf": None, "size": 0}

        def resize(ctx, size):
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
        ep = A.init(A.EncodeParams, A.ST_ENCODE_PARAMS, quality=90.0, target_psnr=50.0)
        fut = C.c_void_p()
        with api.Encoder(max_num_cpu_threads=2) as enc:
            _check(lib.nvimgcodecEncoderEncode(enc._h, (C.c_void_p * 1)(image), (C.c_void_p * 1)(cs), 1, C.byref(ep), C.byref(fut)),
                   "nvimgcodecEncoderEncode")
            _check(lib.nvimgcodecFutureWaitForAll(fut), "nvimgcodecFutureWaitForAll")
            st = (C.c_uint32 * 1)()
            size = C.c_size_t()
            lib.nvimgcodecFutureGetProcessingStatus(fut, st, C.byref(size))
            lib.nvimgcodecFutureDestroy(fut)
        data = C.string_at(sink["buf"], sink["size"]) if st[0] == A.PS_SUCCESS and sink["buf"] is not None else None
        lib.nvimgcodecImageDestroy(image)
        lib.nvimgcodecCodeStreamDestroy(cs)
        return st[0], data

    for sub, hs, vs, css in (("420", 2, 2, A.SAMPLING_420), ("422", 2, 1, A.SAMPLING_422), ("444", 1, 1, A.SAMPLING_444)):
        planes = _planes_like_libjpeg(rgb, hs, vs)
        st, data = encode(planes, css, css, A.COLORSPEC_SYCC)
        assert st == A.PS_SUCCESS and data == oracle.encode(rgb, sub, 90), sub
    planes = _planes_like_libjpeg(rgb, 2, 2)
    st, data = encode(planes, A.SAMPLING_420, A.SAMPLING_420, A.COLORSPEC_SYCC, progressive=True)
    assert st == A.PS_SUCCESS and b"\xff\xc2" in data[:700]
    assert all(np.array_equal(a, b) for a, b in zip(oracle.decode_coefficients(data)[0], oracle.decode_coefficients(oracle.encode(rgb, "420", 90))[0]))
    # resampling planar YCbCr is not offered (nor by the reference's encoder): 4:2:0 planes cannot become a 4:4:4 stream
    st, data = encode(planes, A.SAMPLING_420, A.SAMPLING_444, A.COLORSPEC_SYCC)
    assert st != A.PS_SUCCESS and data is None
    st, data = encode(planes, A.SAMPLING_420, A.SAMPLING_420, A.COLORSPEC_SRGB)
    assert st != A.PS_SUCCESS and data is None


def test_several_decode_calls_outstanding(torch_mod):
    """What a throughput-minded caller does (hipimtrans -p N): nvimgcodecDecoderDecode returns a future, the next batch is submitted
    before the first is waited for.  With an earlier call in flight the plugin sends a batch out whole instead of in pieces (six
    pages); five calls of 120 mixed goldens each outstanding at once -- every sample reports exactly once, every picture bit-exact,
    a corrupt file in one batch fails alone."""
    lib = A.bind(_native.load_host())
    inst, dec = _setup(lib)
    before = lib.hipjpegTestDoubleReports()
    entries = [e for e in _M["decode"] if e["pixels"]][:30]
    cases = [load_decode_case(e) for e in entries]
    ncalls, per_call = 5, 120
    calls = []
    for c in range(ncalls):
        streams, images, outs, keep = [], [], [], []
        for i in range(per_call):
            jpeg, rgb = cases[(i + 7 * c) % len(cases)]
            if c == 2 and i == 50:
                jpeg = jpeg[:150]  # truncated inside the headers / first bytes of the scan
            arr = np.frombuffer(jpeg, dtype=np.uint8)
            keep.append(arr)
            cs = C.c_void_p()
            assert lib.nvimgcodecCodeStreamCreateFromHostMem(inst, C.byref(cs), arr.ctypes.data, arr.size) == 0
            h, w = rgb.shape[:2]
            t = torch_mod.zeros((h, w, 3), dtype=torch_mod.uint8, device="cuda")
            info = A.init(A.ImageInfo, A.ST_IMAGE_INFO, sample_format=A.SAMPLEFORMAT_I_RGB, color_spec=A.COLORSPEC_SRGB, num_planes=1,
                          buffer=t.data_ptr(), buffer_size=w * 3 * h, buffer_kind=A.BUFFER_KIND_STRIDED_DEVICE, cuda_stream=0)
            pi = info.plane_info[0]
            pi.width, pi.height, pi.row_stride, pi.num_channels, pi.sample_type = w, h, w * 3, 3, A.SAMPLE_DATA_TYPE_UINT8
            im = C.c_void_p()
            assert lib.nvimgcodecImageCreate(inst, C.byref(im), C.byref(info)) == 0
            streams.append(cs)
            images.append(im)
            outs.append(t)
        dp = A.init(A.DecodeParams, A.ST_DECODE_PARAMS)
        fut = C.c_void_p()
        cs_arr, im_arr = (C.c_void_p * per_call)(*streams), (C.c_void_p * per_call)(*images)
        assert lib.nvimgcodecDecoderDecode(dec, cs_arr, im_arr, per_call, C.byref(dp), C.byref(fut)) == 0
        calls.append((fut, streams, images, outs, keep, cs_arr, im_arr, dp))
    for c, (fut, streams, images, outs, keep, *_rest) in enumerate(calls):
        assert lib.nvimgcodecFutureWaitForAll(fut) == 0
        st = (C.c_uint32 * per_call)()
        n = C.c_size_t()
        lib.nvimgcodecFutureGetProcessingStatus(fut, st, C.byref(n))
        assert n.value == per_call
        torch_mod.cuda.synchronize()
        for i in range(per_call):
            if c == 2 and i == 50:
                assert st[i] != A.PS_SUCCESS
            else:
                assert st[i] == A.PS_SUCCESS, (c, i, st[i])
                assert np.array_equal(outs[i].cpu().numpy(), cases[(i + 7 * c) % len(cases)][1]), (c, i)
        lib.nvimgcodecFutureDestroy(fut)
        for i in range(per_call):
            lib.nvimgcodecImageDestroy(images[i])
            lib.nvimgcodecCodeStreamDestroy(streams[i])
    assert lib.hipjpegTestDoubleReports() == before
    lib.nvimgcodecDecoderDestroy(dec)
    lib.nvimgcodecInstanceDestroy(inst)


def test_decoder_options_like_the_reference_plugins(torch_mod):
    """Option strings use the reference's grammar (extensions/libjpeg_turbo/libjpeg_turbo_decoder.cpp:250-276).  A key addressed to this
    decoder BY NAME that it does not know is reported through the framework's log (it was ignored in silence); `fast_idct=1`
    -- JDCT_FASTEST, other pixels than ISLOW -- is not offered: canDecode passes every sample on to the next decoder of the chain, the way
    nvjpeg -> libjpeg_turbo hand over; `hybrid_huffman_threshold` is accepted (extensions/nvjpeg/cuda_decoder.cpp:188-209)."""
    lib = A.bind(_native.load_host())
    messages = []

    def on_message(severity, category, data, user):
        messages.append((severity, data.contents.message.decode(errors="replace") if data.contents.message else ""))
        return 0

    cb = A.DebugCallback(on_message)
    messenger = A.init(A.DebugMessengerDesc, A.ST_DEBUG_MESSENGER_DESC, message_severity=0xFFFF, message_category=0xFFFF, user_callback=cb)
    jpeg, rgb = _case("s50x37_420_base_q90")

    # unknown key with this decoder's name: a warning; without a name: silence; known keys: accepted
    inst, dec = _setup(lib, options=b"hipjpeg_decoder:bogus=1 :other_plugins_key=3 hipjpeg_decoder:hybrid_huffman_threshold=1000000 :fancy_upsampling=1",
                       messenger=messenger)
    buf = np.zeros((37, 50, 3), dtype=np.uint8)
    assert _c_api_decode(lib, inst, dec, jpeg, 37, 50, A.SAMPLEFORMAT_I_RGB, 1, 3, buf.ctypes.data, 150, A.BUFFER_KIND_STRIDED_HOST) == A.PS_SUCCESS
    assert np.array_equal(buf, rgb)
    assert any("unknown option 'bogus'" in m for _, m in messages), messages
    assert not any("other_plugins_key" in m for _, m in messages)
    lib.nvimgcodecDecoderDestroy(dec)
    lib.nvimgcodecInstanceDestroy(inst)

    # fast_idct: the HIP decoder steps aside, the next decoder of the chain gets the sample
    messages.clear()
    cpu = FakeDecoderPlugin("cpu_fallback", priority=A.PRIORITY_NORMAL, fill=0x42)
    inst, dec = _setup(lib, extra_plugins=[cpu], options=b":fast_idct=1", messenger=messenger)
    buf = np.zeros((37, 50, 3), dtype=np.uint8)
    assert _c_api_decode(lib, inst, dec, jpeg, 37, 50, A.SAMPLEFORMAT_I_RGB, 1, 3, buf.ctypes.data, 150, A.BUFFER_KIND_STRIDED_HOST) == A.PS_SUCCESS
    assert buf.flat[0] == 0x42 and cpu.count("decode") == 1
    assert any("fast_idct" in m for _, m in messages), messages
    lib.nvimgcodecDecoderDestroy(dec)
    lib.nvimgcodecInstanceDestroy(inst)
    del cb


def test_multi_device_decoder_partitions_a_batch_over_device_queues(torch_mod):
    """api.MultiDeviceDecoder: one process, a Decoder + host thread per device entry, the batch partitioned by size, results in input
    order.  Two queues on device 0 here (one-GPU box): every picture bit-exact, both queues used, nothing decoded twice."""
    from nvimagecodec_amd import api
    entries = [e for e in _M["decode"] if e["pixels"]][:40]
    cases = [load_decode_case(e) for e in entries]
    with api.MultiDeviceDecoder([0, 0], max_num_cpu_threads=2) as dec:
        queues = dec.shard([c[0] for c in cases])
        assert sorted(i for q in queues for i in q) == list(range(len(cases))) and all(queues)
        imgs = dec.decode([c[0] for c in cases])
        torch_mod.cuda.synchronize()
        for e, (jpeg, rgb), im in zip(entries, cases, imgs):
            assert im is not None, e["name"]
            assert np.array_equal(np.asarray(im.cpu()._array), rgb), e["name"]
