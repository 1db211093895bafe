"""Host logic of the dispatch harness (no GPU): registry priorities, canDecode -> fallback, runtime-failure fallback,
backend filtering, stream parsing, futures.  Modelled on the reference's test/api/can_decode_test.cpp,
test/decoder_worker_test.cpp and test/parsers/jpeg_test.cpp, with a fake plugin driven through the real C tables.

Without a GPU the hipjpeg_decoder's create() fails loudly (no CPU fallback inside it), so the chain skips it -- which
is itself one of the behaviours under test."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_decode_case
from fake_plugin import FakeDecoderPlugin
from nvimagecodec_amd import _native
from nvimagecodec_amd import abi as A

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    _M = json.load(_f)


@pytest.fixture()
def lib():
    return A.bind(_native.load_host())


def make_instance(lib, load_ext=0):
    ci = A.init(A.InstanceCreateInfo, A.ST_INSTANCE_CREATE_INFO, load_builtin_modules=1, load_extension_modules=load_ext)
    inst = C.c_void_p()
    assert lib.nvimgcodecInstanceCreate(C.byref(inst), C.byref(ci)) == 0
    return inst


def host_image(lib, inst, h, w, fmt=A.SAMPLEFORMAT_I_RGB):
    buf = np.zeros((h, w, 3), dtype=np.uint8)
    info = A.init(A.ImageInfo, A.ST_IMAGE_INFO, sample_format=fmt, color_spec=A.COLORSPEC_SRGB, num_planes=1, buffer=buf.ctypes.data,
                  buffer_size=buf.nbytes, buffer_kind=A.BUFFER_KIND_STRIDED_HOST)
    pi = info.plane_info[0]
    pi.width, pi.height, pi.row_stride, pi.num_channels, pi.sample_type = w, h, w * 3, 3, A.SAMPLE_DATA_TYPE_UINT8
    im = C.c_void_p()
    assert lib.nvimgcodecImageCreate(inst, C.byref(im), C.byref(info)) == 0
    return im, buf


def code_stream(lib, inst, jpeg):
    arr = np.frombuffer(jpeg, dtype=np.uint8)
    cs = C.c_void_p()
    st = lib.nvimgcodecCodeStreamCreateFromHostMem(inst, C.byref(cs), arr.ctypes.data, arr.size)
    return st, cs, arr


def make_decoder(lib, inst, device_id=A.DEVICE_CPU_ONLY, backends=None, options=b""):
    ep = A.init(A.ExecutionParams, A.ST_EXECUTION_PARAMS, device_id=device_id, max_num_cpu_threads=2)
    keep = None
    if backends:
        keep = (A.Backend * len(backends))()
        for i, k in enumerate(backends):
            keep[i].struct_type, keep[i].struct_size, keep[i].kind = A.ST_BACKEND, C.sizeof(A.Backend), k
        ep.num_backends, ep.backends = len(backends), C.cast(keep, C.POINTER(A.Backend))
    dec = C.c_void_p()
    assert lib.nvimgcodecDecoderCreate(inst, C.byref(dec), C.byref(ep), options) == 0
    return dec, keep


def decode(lib, dec, streams, images):
    n = len(streams)
    dp = A.init(A.DecodeParams, A.ST_DECODE_PARAMS)
    fut = C.c_void_p()
    assert lib.nvimgcodecDecoderDecode(dec, (C.c_void_p * n)(*streams), (C.c_void_p * n)(*images), n, C.byref(dp), C.byref(fut)) == 0
    assert lib.nvimgcodecFutureWaitForAll(fut) == 0
    st = (C.c_uint32 * n)()
    size = C.c_size_t()
    assert lib.nvimgcodecFutureGetProcessingStatus(fut, st, C.byref(size)) == 0
    assert size.value == n
    lib.nvimgcodecFutureDestroy(fut)
    return list(st)


def _jpeg(name):
    return load_decode_case(next(e for e in _M["decode"] if e["name"] == name))[0]


def test_properties(lib):
    p = A.init(A.Properties, A.ST_PROPERTIES)
    assert lib.nvimgcodecGetProperties(C.byref(p)) == 0
    assert p.ext_api_version == 200


def test_code_stream_info_matches_reference_parser_semantics(lib):
    """What src/parsers/jpeg.cpp:311-353 reports: planar RGB/Y sample format, SYCC/GRAY colour spec, subsampling enum,
    one plane per component with the full image size, UINT8, SOF marker as encoding."""
    inst = make_instance(lib)
    expect = {"444": A.SAMPLING_444, "422": A.SAMPLING_422, "420": A.SAMPLING_420, "440": A.SAMPLING_440, "411": A.SAMPLING_411,
              "410": A.SAMPLING_410, "gray": A.SAMPLING_GRAY}
    for e in _M["decode"][::7]:
        jpeg, _ = load_decode_case(e)
        st, cs, keep = code_stream(lib, inst, jpeg)
        assert st == 0
        ji = A.init(A.JpegImageInfo, A.ST_JPEG_IMAGE_INFO)
        info = A.init(A.ImageInfo, A.ST_IMAGE_INFO, struct_next=C.addressof(ji))
        assert lib.nvimgcodecCodeStreamGetImageInfo(cs, C.byref(info)) == 0
        gray = e["sub"] == "gray"
        assert info.codec_name == b"jpeg"
        assert info.num_planes == (1 if gray else 3)
        assert info.sample_format == (A.SAMPLEFORMAT_P_Y if gray else A.SAMPLEFORMAT_P_RGB)
        assert info.color_spec == (A.COLORSPEC_GRAY if gray else A.COLORSPEC_SYCC)
        assert info.chroma_subsampling == expect[e["sub"]]
        for p in range(info.num_planes):
            pi = info.plane_info[p]
            assert (pi.width, pi.height, pi.num_channels, pi.sample_type, pi.precision) == (e["width"], e["height"], 1, A.SAMPLE_DATA_TYPE_UINT8, 8)
        assert ji.encoding == (0xC2 if e["progressive"] else 0xC0)
        assert (info.orientation.rotated, info.orientation.flip_x, info.orientation.flip_y) == (0, 0, 0)
        lib.nvimgcodecCodeStreamDestroy(cs)
    lib.nvimgcodecInstanceDestroy(inst)


def test_exif_orientation_is_reported(lib):
    inst = make_instance(lib)
    jpeg = _jpeg("s64x48_420_base_q90")
    # APP1 Exif, little-endian TIFF, IFD0 with one entry: Orientation (0x0112) SHORT = 6  (rotate 90 CW -> rotated=270)
    tiff = b"II*\x00\x08\x00\x00\x00" + b"\x01\x00" + b"\x12\x01\x03\x00\x01\x00\x00\x00\x06\x00\x00\x00" + b"\x00\x00\x00\x00"
    payload = b"Exif\x00\x00" + tiff
    app1 = b"\xff\xe1" + (len(payload) + 2).to_bytes(2, "big") + payload
    st, cs, keep = code_stream(lib, inst, jpeg[:2] + app1 + jpeg[2:])
    assert st == 0
    info = A.init(A.ImageInfo, A.ST_IMAGE_INFO)
    assert lib.nvimgcodecCodeStreamGetImageInfo(cs, C.byref(info)) == 0
    assert (info.orientation.rotated, info.orientation.flip_x, info.orientation.flip_y) == (270, 0, 0)
    lib.nvimgcodecCodeStreamDestroy(cs)
    lib.nvimgcodecInstanceDestroy(inst)


def test_non_jpeg_stream_is_rejected(lib):
    inst = make_instance(lib)
    st, cs, keep = code_stream(lib, inst, b"\x89PNG\r\n\x1a\n" + bytes(64))
    assert st == A.STATUS_CODESTREAM_UNSUPPORTED
    lib.nvimgcodecInstanceDestroy(inst)


def test_priority_order_and_can_decode_fallback(lib):
    """Lower priority value is asked first; what it rejects in canDecode goes to the next decoder
    (reference src/decoder_worker.cpp:258-296)."""
    inst = make_instance(lib)
    picky = FakeDecoderPlugin("picky", priority=A.PRIORITY_HIGH, fill=0x11,
                              can_status=lambda i, ci, ii: A.PS_SUCCESS if ci.plane_info[0].width == 64 else A.PS_SAMPLE_FORMAT_UNSUPPORTED)
    backup = FakeDecoderPlugin("backup", priority=A.PRIORITY_LOW, fill=0x22)
    for p in (backup, picky):  # registration order must not matter
        assert lib.nvimgcodecExtensionCreate(inst, None, C.byref(p.ext_desc)) == 0
    dec, _ = make_decoder(lib, inst)
    jpegs = [_jpeg("s64x48_420_base_q90"), _jpeg("s50x37_444_base_q90"), _jpeg("s64x48_gray_base_q50")]
    cs = [code_stream(lib, inst, j) for j in jpegs]
    ims = [host_image(lib, inst, h, w) for (w, h) in ((64, 48), (50, 37), (64, 48))]
    st = decode(lib, dec, [c[1] for c in cs], [i[0] for i in ims])
    assert st == [A.PS_SUCCESS] * 3
    assert ims[0][1].flat[0] == 0x11 and ims[2][1].flat[0] == 0x11 and ims[1][1].flat[0] == 0x22
    assert picky.log[:2] == [("create", 0), ("canDecode", 3)] and ("decode", 2) in picky.log
    assert ("canDecode", 1) in backup.log and ("decode", 1) in backup.log
    lib.nvimgcodecDecoderDestroy(dec)
    assert picky.count("destroy") == 1 and backup.count("destroy") == 1
    lib.nvimgcodecInstanceDestroy(inst)


def test_runtime_failure_falls_through_per_sample(lib):
    """imageReady(FAIL) from one decoder hands that sample to the next one (reference src/decoder_worker.cpp:178-192)."""
    inst = make_instance(lib)
    flaky = FakeDecoderPlugin("flaky", priority=A.PRIORITY_HIGH, fill=0x33, decode_status=lambda i: A.PS_FAIL if i == 1 else A.PS_SUCCESS)
    backup = FakeDecoderPlugin("backup", priority=A.PRIORITY_NORMAL, fill=0x44)
    for p in (flaky, backup):
        assert lib.nvimgcodecExtensionCreate(inst, None, C.byref(p.ext_desc)) == 0
    dec, _ = make_decoder(lib, inst)
    jpeg = _jpeg("s64x48_420_base_q90")
    cs = [code_stream(lib, inst, jpeg) for _ in range(3)]
    ims = [host_image(lib, inst, 48, 64) for _ in range(3)]
    st = decode(lib, dec, [c[1] for c in cs], [i[0] for i in ims])
    assert st == [A.PS_SUCCESS] * 3
    assert [int(i[1].flat[0]) for i in ims] == [0x33, 0x44, 0x33]
    lib.nvimgcodecDecoderDestroy(dec)
    lib.nvimgcodecInstanceDestroy(inst)


def test_all_decoders_failing_reports_last_status(lib):
    inst = make_instance(lib)
    a = FakeDecoderPlugin("a", priority=A.PRIORITY_HIGH, can_status=A.PS_ENCODING_UNSUPPORTED)
    b = FakeDecoderPlugin("b", priority=A.PRIORITY_LOW, decode_status=A.PS_IMAGE_CORRUPTED)
    for p in (a, b):
        assert lib.nvimgcodecExtensionCreate(inst, None, C.byref(p.ext_desc)) == 0
    dec, _ = make_decoder(lib, inst)
    st_, cs, keep = code_stream(lib, inst, _jpeg("s64x48_420_base_q90"))
    im, buf = host_image(lib, inst, 48, 64)
    assert decode(lib, dec, [cs], [im]) == [A.PS_IMAGE_CORRUPTED]
    lib.nvimgcodecDecoderDestroy(dec)
    lib.nvimgcodecInstanceDestroy(inst)


def test_backend_allow_list_and_create_failure(lib):
    """Only allowed backend kinds enter the chain (test/decoder_worker_test.cpp:110-171); a decoder whose create() fails is
    treated as absent (src/decoder_worker.cpp:80-93)."""
    inst = make_instance(lib)
    gpu = FakeDecoderPlugin("gpuish", backend_kind=A.BACKEND_KIND_GPU_ONLY, priority=A.PRIORITY_HIGH, fill=0x55)
    broken = FakeDecoderPlugin("broken", priority=A.PRIORITY_VERY_HIGH, create_status=A.STATUS_INVALID_PARAMETER)
    cpu = FakeDecoderPlugin("cpu", priority=A.PRIORITY_LOW, fill=0x66)
    for p in (gpu, broken, cpu):
        assert lib.nvimgcodecExtensionCreate(inst, None, C.byref(p.ext_desc)) == 0
    dec, keep = make_decoder(lib, inst, device_id=0, backends=[A.BACKEND_KIND_CPU_ONLY], options=b":fancy_upsampling=0 cpu:x=1")
    st_, cs, k2 = code_stream(lib, inst, _jpeg("s64x48_420_base_q90"))
    im, buf = host_image(lib, inst, 48, 64)
    assert decode(lib, dec, [cs], [im]) == [A.PS_SUCCESS]
    assert buf.flat[0] == 0x66
    assert gpu.log == []                      # filtered out before creation
    assert broken.count("create") == 1 and broken.count("canDecode") == 0
    assert cpu.seen_options == b":fancy_upsampling=0 cpu:x=1" and cpu.seen_device == 0
    lib.nvimgcodecDecoderDestroy(dec)
    lib.nvimgcodecInstanceDestroy(inst)


def test_hip_decoder_never_decodes_without_a_gpu(lib):
    """With the real extension loaded but no usable device, hipjpeg_decoder::create fails and the sample ends in the
    fallback; there is no CPU pixel path hiding inside the HIP plugin."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    inst = make_instance(lib, load_ext=1)
    cpu = FakeDecoderPlugin("cpu", priority=A.PRIORITY_NORMAL, fill=0x77)
    assert lib.nvimgcodecExtensionCreate(inst, None, C.byref(cpu.ext_desc)) == 0
    dec, _ = make_decoder(lib, inst, device_id=0)
    st_, cs, keep = code_stream(lib, inst, _jpeg("s64x48_420_base_q90"))
    im, buf = host_image(lib, inst, 48, 64)
    assert decode(lib, dec, [cs], [im]) == [A.PS_SUCCESS]
    assert buf.flat[0] == 0x77
    lib.nvimgcodecDecoderDestroy(dec)
    lib.nvimgcodecInstanceDestroy(inst)


def test_can_decode_api_force_format(lib):
    """nvimgcodecDecoderCanDecode: with force_format=0 a 'possible with other parameters' status (low bits 01) counts as
    decodable (reference src/image_generic_decoder.cpp:120-121, test/api/can_de_en_code_common.h)."""
    inst = make_instance(lib)
    p = FakeDecoderPlugin("p", can_status=A.PS_SAMPLE_FORMAT_UNSUPPORTED)
    assert lib.nvimgcodecExtensionCreate(inst, None, C.byref(p.ext_desc)) == 0
    dec, _ = make_decoder(lib, inst)
    st_, cs, keep = code_stream(lib, inst, _jpeg("s64x48_420_base_q90"))
    im, buf = host_image(lib, inst, 48, 64)
    dp = A.init(A.DecodeParams, A.ST_DECODE_PARAMS)
    out = (C.c_uint32 * 1)()
    assert lib.nvimgcodecDecoderCanDecode(dec, (C.c_void_p * 1)(cs), (C.c_void_p * 1)(im), 1, C.byref(dp), out, 0) == 0
    assert out[0] == A.PS_SUCCESS
    assert lib.nvimgcodecDecoderCanDecode(dec, (C.c_void_p * 1)(cs), (C.c_void_p * 1)(im), 1, C.byref(dp), out, 1) == 0
    assert out[0] == A.PS_SAMPLE_FORMAT_UNSUPPORTED
    p.can_status = A.PS_CODESTREAM_UNSUPPORTED  # hard failure (low bits 11) is never decodable
    assert lib.nvimgcodecDecoderCanDecode(dec, (C.c_void_p * 1)(cs), (C.c_void_p * 1)(im), 1, C.byref(dp), out, 0) == 0
    assert out[0] == A.PS_CODESTREAM_UNSUPPORTED
    lib.nvimgcodecDecoderDestroy(dec)
    lib.nvimgcodecInstanceDestroy(inst)


def test_file_code_stream_and_empty_batch(lib, tmp_path):
    inst = make_instance(lib)
    p = FakeDecoderPlugin("p", fill=0x12)
    assert lib.nvimgcodecExtensionCreate(inst, None, C.byref(p.ext_desc)) == 0
    path = tmp_path / "a.jpg"
    path.write_bytes(_jpeg("s50x37_422_prog_q90"))
    cs = C.c_void_p()
    assert lib.nvimgcodecCodeStreamCreateFromFile(inst, C.byref(cs), str(path).encode()) == 0
    info = A.init(A.ImageInfo, A.ST_IMAGE_INFO)
    assert lib.nvimgcodecCodeStreamGetImageInfo(cs, C.byref(info)) == 0
    assert (info.plane_info[0].width, info.plane_info[0].height, info.chroma_subsampling) == (50, 37, A.SAMPLING_422)
    assert lib.nvimgcodecCodeStreamCreateFromFile(inst, C.byref(C.c_void_p()), b"/nonexistent/x.jpg") != 0
    dec, _ = make_decoder(lib, inst)
    im, buf = host_image(lib, inst, 37, 50)
    assert decode(lib, dec, [cs], [im]) == [A.PS_SUCCESS]
    assert decode(lib, dec, [], []) == []
    lib.nvimgcodecDecoderDestroy(dec)
    lib.nvimgcodecInstanceDestroy(inst)
