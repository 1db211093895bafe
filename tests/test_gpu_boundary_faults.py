"""The boundary under stress: oversized and undersized descriptors fail alone, and an exception anywhere inside the library
(injected at named sites through hipjpegTestSetFault) still ends with every sample reported exactly once and the decoder
usable for the next batch.  Reference behaviour: every plugin entry point is try/catch and marks the batch's samples FAIL
(extensions/nvjpeg/cuda_decoder.cpp:559-562,602-608; extensions/libjpeg_turbo/libjpeg_turbo_decoder.cpp:226-236)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, load_decode_case
from nvimagecodec_amd import _native as N
from nvimagecodec_amd import abi as A
from nvimagecodec_amd.lowlevel import BatchDecoder
from nvimagecodec_amd.synth import synth_image

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


def _forged_sof(jpeg, width, height):
    """The same file with the SOF0 dimensions overwritten: a few hundred bytes that claim a huge frame."""
    i = jpeg.find(b"\xff\xc0")
    assert i > 0
    return jpeg[: i + 5] + height.to_bytes(2, "big") + width.to_bytes(2, "big") + jpeg[i + 9:]


def test_a_forged_huge_frame_fails_alone(torch_mod):
    """a forged 65535x65535 SOF must not make the batch reserve tens of GB or fail its neighbours (the reference
    fails only the offending sample: jpeg_mem.cpp:183-196 refuses >= 2^29 samples before allocating)."""
    torch = torch_mod
    good = [_case("s64x48_420_base_q90"), _case("s50x37_420_base_q90"), _case("c1_640x480_444_base_q90")]
    forged = _forged_sof(good[0][0], 65535, 65535)
    jpegs = [good[0][0], forged, good[1][0], good[2][0]]
    dec = BatchDecoder(device=0, num_threads=2)
    outs = [torch.zeros((48, 64, 3), dtype=torch.uint8, device="cuda"), torch.zeros((8, 8, 3), dtype=torch.uint8, device="cuda"),
            torch.zeros((37, 50, 3), dtype=torch.uint8, device="cuda"), torch.zeros((480, 640, 3), dtype=torch.uint8, device="cuda")]
    for gh in (False, True):
        _, st = dec.decode(jpegs, outs=outs, check=False, gpu_huffman=gh)
        torch.cuda.synchronize()
        assert st[1] == 6 and st[0] == st[2] == st[3] == 0, st
        for k, (o, g) in enumerate(zip((outs[0], outs[2], outs[3]), good)):
            assert np.array_equal(o.cpu().numpy(), g[1] if g[1] is not None else oracle.decode(g[0])), k
    dec.close()


def test_a_short_pitch_fails_alone(torch_mod):
    torch = torch_mod
    a, b = _case("s64x48_420_base_q90"), _case("s50x37_420_base_q90")
    dec = BatchDecoder(device=0, num_threads=2)
    wide = torch.zeros((48, 64, 3), dtype=torch.uint8, device="cuda")
    narrow = torch.zeros((37, 40, 3), dtype=torch.uint8, device="cuda")   # rows of 120 bytes for a 150-byte picture row
    guard = narrow.clone()
    _, st = dec.decode([a[0], b[0]], outs=[wide, narrow], check=False, gpu_huffman=True)
    torch.cuda.synchronize()
    assert st == [0, 1]
    assert np.array_equal(wide.cpu().numpy(), a[1]) and torch.equal(narrow, guard)
    dec.close()


def _setup(lib, threads=4):
    ci = A.init(A.InstanceCreateInfo, A.ST_INSTANCE_CREATE_INFO, load_builtin_modules=1, load_extension_modules=1)
    inst = C.c_void_p()
    assert lib.nvimgcodecInstanceCreate(C.byref(inst), C.byref(ci)) == 0
    ep = A.init(A.ExecutionParams, A.ST_EXECUTION_PARAMS, device_id=0, max_num_cpu_threads=threads)
    dec = C.c_void_p()
    assert lib.nvimgcodecDecoderCreate(inst, C.byref(dec), C.byref(ep), b"") == 0
    return inst, dec


def _decode_batch(torch, lib, inst, dec, cases, shrink=None):
    """cases: [(jpeg, rgb)]; device outputs; returns (statuses, outputs).  shrink = index whose image descriptor claims a
    buffer one row too short."""
    n = len(cases)
    css, ims, outs, keep = [], [], [], []
    for i, (jpeg, rgb) in enumerate(cases):
        arr = np.frombuffer(jpeg, dtype=np.uint8)
        keep.append(arr)
        cs = C.c_void_p()
        assert lib.nvimgcodecCodeStreamCreateFromHostMem(inst, C.byref(cs), arr.ctypes.data, arr.size) == 0
        h, w = rgb.shape[:2]
        out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
        hh = h - 1 if shrink == i else h
        info = A.init(A.ImageInfo, A.ST_IMAGE_INFO, sample_format=A.SAMPLEFORMAT_I_RGB, color_spec=A.COLORSPEC_SRGB, num_planes=1,
                      buffer=out.data_ptr(), buffer_size=w * 3 * hh, buffer_kind=A.BUFFER_KIND_STRIDED_DEVICE, cuda_stream=0)
        pi = info.plane_info[0]
        pi.width, pi.height, pi.row_stride, pi.num_channels, pi.sample_type = w, hh, w * 3, 3, A.SAMPLE_DATA_TYPE_UINT8
        im = C.c_void_p()
        assert lib.nvimgcodecImageCreate(inst, C.byref(im), C.byref(info)) == 0
        css.append(cs)
        ims.append(im)
        outs.append(out)
    dp = A.init(A.DecodeParams, A.ST_DECODE_PARAMS)
    fut = C.c_void_p()
    rc = lib.nvimgcodecDecoderDecode(dec, (C.c_void_p * n)(*css), (C.c_void_p * n)(*ims), n, C.byref(dp), C.byref(fut))
    assert rc == 0 and fut
    assert lib.nvimgcodecFutureWaitForAll(fut) == 0     # a sample that never reports would hang here
    st = (C.c_uint32 * n)()
    cnt = C.c_size_t()
    lib.nvimgcodecFutureGetProcessingStatus(fut, st, C.byref(cnt))
    assert cnt.value == n
    lib.nvimgcodecFutureDestroy(fut)
    for im, cs in zip(ims, css):
        lib.nvimgcodecImageDestroy(im)
        lib.nvimgcodecCodeStreamDestroy(cs)
    torch.cuda.synchronize()
    return list(st), outs


def _cases():
    names = ["s64x48_420_base_q90", "s50x37_420_base_q90", "s33x65_422_prog_q50", "r130x70_420_base_rst7", "s64x48_444_base_q90",
             "c1_640x480_444_base_q90"]
    out = []
    for n in names:
        j, rgb = _case(n)
        out.append((j, rgb if rgb is not None else oracle.decode(j)))
    return out


def test_an_undersized_image_descriptor_fails_alone(torch_mod):
    lib = A.bind(N.load_host())
    inst, dec = _setup(lib)
    cases = _cases()
    st, outs = _decode_batch(torch_mod, lib, inst, dec, cases, shrink=2)
    assert st[2] != A.PS_SUCCESS and all(s == A.PS_SUCCESS for i, s in enumerate(st) if i != 2), st
    for i, ((_, rgb), o) in enumerate(zip(cases, outs)):
        if i != 2:
            assert np.array_equal(o.cpu().numpy(), rgb), i
    lib.nvimgcodecDecoderDestroy(dec)
    lib.nvimgcodecInstanceDestroy(inst)


@pytest.mark.parametrize("site", ["marshal", "plan", "entropy_stage", "finalize", "transfer", "launch", "resolve"])
@pytest.mark.parametrize("countdown", [1, 3])
def test_an_exception_inside_the_library_resolves_every_future(torch_mod, site, countdown):
    """Through nvimgcodecDecoderDecode: the `countdown`-th passage of `site` throws.  Every future resolves (nothing hangs,
    nothing is reported twice -- the host harness throws on a double set like the reference, src/processing_results.cpp:109),
    samples are SUCCESS with correct pixels or a failure status, and the NEXT batch decodes completely."""
    lib = A.bind(N.load_host())
    inst, dec = _setup(lib)
    cases = _cases()
    st0, _ = _decode_batch(torch_mod, lib, inst, dec, cases)   # warm: pages allocated
    assert all(s == A.PS_SUCCESS for s in st0)
    assert N.load().hipjpegTestSetFault(site.encode(), countdown) == 0
    try:
        st, outs = _decode_batch(torch_mod, lib, inst, dec, cases)
    finally:
        N.load().hipjpegTestSetFault(None, 0)
    failed = [i for i, s in enumerate(st) if s != A.PS_SUCCESS]
    if site in ("marshal", "entropy_stage"):
        assert len(failed) <= 1        # one sample's trouble stays that sample's
    for i, ((_, rgb), o) in enumerate(zip(cases, outs)):
        if i not in failed:
            assert np.array_equal(o.cpu().numpy(), rgb), (site, i)
    assert lib.hipjpegTestDoubleReports() == 0
    st2, outs2 = _decode_batch(torch_mod, lib, inst, dec, cases)
    assert all(s == A.PS_SUCCESS for s in st2), (site, st2)
    for (_, rgb), o in zip(cases, outs2):
        assert np.array_equal(o.cpu().numpy(), rgb)
    lib.nvimgcodecDecoderDestroy(dec)
    lib.nvimgcodecInstanceDestroy(inst)


def test_the_c_abi_turns_exceptions_into_status_codes(torch_mod):
    torch = torch_mod
    a = _case("s64x48_420_base_q90")
    dec = BatchDecoder(device=0, num_threads=2)
    out = torch.zeros((48, 64, 3), dtype=torch.uint8, device="cuda")
    for site in ("plan", "entropy_stage", "finalize", "transfer", "launch", "resolve"):
        N.load().hipjpegTestSetFault(site.encode(), 1)
        with pytest.raises(N.HipJpegError) as ei:
            dec.decode([a[0], a[0]], outs=[out, out.clone()], gpu_huffman=True)
        assert ei.value.status == 10, site
        N.load().hipjpegTestSetFault(None, 0)
        out.zero_()
        dec.decode([a[0]], outs=[out], gpu_huffman=True)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), a[1]), site
    dec.close()
