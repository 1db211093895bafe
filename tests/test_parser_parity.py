"""The JPEG parser as the reference's own parser tests exercise it (test/parsers/jpeg_test.cpp:244-480): four-component
colour spaces, error cases, fill bytes in front of markers, all eight EXIF orientations.  The reference's image files are
git-LFS stubs here, so every input is built from our goldens by the same byte edits the reference tests make (`replace`)
or from hand-written marker segments; the expected values are the ones those tests assert."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_decode_case
from nvimagecodec_amd import _native
from nvimagecodec_amd import abi as A
from test_host_framework import code_stream, make_instance

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    _M = json.load(_f)


@pytest.fixture()
def lib():
    return A.bind(_native.load_host())


def _jpeg(name):
    return load_decode_case(next(e for e in _M["decode"] if e["name"] == name))[0]


def _info(lib, inst, jpeg):
    st, cs, keep = code_stream(lib, inst, jpeg)
    if st != 0:
        return st, None, None
    ji = A.init(A.JpegImageInfo, A.ST_JPEG_IMAGE_INFO)
    info = A.init(A.ImageInfo, A.ST_IMAGE_INFO, struct_next=C.addressof(ji))
    st = lib.nvimgcodecCodeStreamGetImageInfo(cs, C.byref(info))
    lib.nvimgcodecCodeStreamDestroy(cs)
    return st, info, ji


def _exif_app1(orientation, big_endian=False, extra_entries=0):
    """APP1 'Exif' segment whose IFD0 holds (optionally some other tags and) Orientation (0x0112) SHORT."""
    e = ">" if big_endian else "<"
    import struct
    entries = []
    for k in range(extra_entries):  # e.g. Make/Model-like ASCII tags in front, so the orientation is not the first entry
        entries.append(struct.pack(e + "HHII", 0x010F + k, 2, 4, 0x41424300 if big_endian else 0x00434241))
    if orientation is not None:
        v = struct.pack(e + "H", orientation) + b"\x00\x00"
        entries.append(struct.pack(e + "HHI", 0x0112, 3, 1) + v)
    ifd = struct.pack(e + "H", len(entries)) + b"".join(entries) + struct.pack(e + "I", 0)
    tiff = (b"MM\x00*" if big_endian else b"II*\x00") + struct.pack(e + "I", 8) + ifd
    payload = b"Exif\x00\x00" + tiff
    return b"\xff\xe1" + (len(payload) + 2).to_bytes(2, "big") + payload


# exif_orientation.h:36-57 (rotated counts counter-clockwise)
EXPECT = {1: (0, 0, 0), 2: (0, 1, 0), 3: (180, 0, 0), 4: (0, 0, 1), 5: (90, 0, 1), 6: (270, 0, 0), 7: (270, 0, 1), 8: (90, 0, 0)}


@pytest.mark.parametrize("big_endian", [False, True])
@pytest.mark.parametrize("orientation", list(range(1, 9)))
def test_exif_orientations(lib, orientation, big_endian):
    inst = make_instance(lib)
    jpeg = _jpeg("s64x48_420_base_q90")
    st, info, _ = _info(lib, inst, jpeg[:2] + _exif_app1(orientation, big_endian, extra_entries=2) + jpeg[2:])
    assert st == 0
    assert (info.orientation.rotated % 360, info.orientation.flip_x, info.orientation.flip_y) == EXPECT[orientation]
    lib.nvimgcodecInstanceDestroy(inst)


def test_exif_without_orientation_and_unknown_values(lib):
    inst = make_instance(lib)
    jpeg = _jpeg("s64x48_420_base_q90")
    for app1 in (_exif_app1(None, extra_entries=1), _exif_app1(0), _exif_app1(9), b"\xff\xe1\x00\x08Exif\x00\x00"):
        st, info, _ = _info(lib, inst, jpeg[:2] + app1 + jpeg[2:])
        assert st == 0
        assert (info.orientation.rotated, info.orientation.flip_x, info.orientation.flip_y) == (0, 0, 0)
    lib.nvimgcodecInstanceDestroy(inst)


def test_errors_like_the_reference(lib):
    inst = make_instance(lib)
    cs = C.c_void_p()
    assert lib.nvimgcodecCodeStreamCreateFromHostMem(inst, C.byref(cs), None, 0) != 0                    # Error_CreateStream_Empty
    jpeg = bytearray(_jpeg("s64x48_420_base_q90"))
    bad_soi = bytes(jpeg[:1]) + b"\xc0" + bytes(jpeg[2:])
    assert code_stream(lib, inst, bad_soi)[0] != 0                                                       # Error_CreateStream_BadSOI
    no_sof = bytes(jpeg).replace(b"\xff\xc0", b"\xff\xfe", 1)
    st, cs, keep = code_stream(lib, inst, no_sof)
    assert st == 0                                                                                       # the parser still matches ...
    info = A.init(A.ImageInfo, A.ST_IMAGE_INFO)
    assert lib.nvimgcodecCodeStreamGetImageInfo(cs, C.byref(info)) != 0                                  # ... Error_GetInfo_NoSOF
    lib.nvimgcodecCodeStreamDestroy(cs)
    lib.nvimgcodecInstanceDestroy(inst)


def test_fill_bytes_in_front_of_markers(lib):
    """ITU-T T.81 B.1.1.2: any marker may be preceded by any number of 0xFF fill bytes (reference test 'Padding')."""
    inst = make_instance(lib)
    jpeg = _jpeg("s64x48_420_base_q90")
    jpeg = jpeg[:2] + _exif_app1(1) + jpeg[2:]
    padded = jpeg.replace(b"\xff\xe0", b"\xff\xff\xff\xff\xe0", 1).replace(b"\xff\xe1", b"\xff\xff\xe1", 1)
    padded = padded.replace(b"\xff\xdb", b"\xff\xff\xff\xdb").replace(b"\xff\xc0", b"\xff\xff\xff\xff\xff\xc0", 1)
    assert padded != jpeg
    st, info, ji = _info(lib, inst, padded)
    assert st == 0
    assert (info.sample_format, info.num_planes, info.color_spec, info.chroma_subsampling) == (A.SAMPLEFORMAT_P_RGB, 3, A.COLORSPEC_SYCC, A.SAMPLING_420)
    assert all((info.plane_info[p].width, info.plane_info[p].height, info.plane_info[p].num_channels) == (64, 48, 1) for p in range(3))
    lib.nvimgcodecInstanceDestroy(inst)


def _four_component(transform):
    """Headers of a 4-component Adobe JPEG (APP14 transform 0 = CMYK, 2 = YCCK), as far as a parser reads them."""
    soi = b"\xff\xd8"
    adobe = b"Adobe" + b"\x00\x64" + b"\x00\x00" + b"\x00\x00" + bytes([transform])
    app14 = b"\xff\xee" + (len(adobe) + 2).to_bytes(2, "big") + adobe
    dqt = b"\xff\xdb\x00\x43\x00" + bytes([16] * 64)
    sof = b"\xff\xc0" + (8 + 3 * 4).to_bytes(2, "big") + b"\x08" + (616).to_bytes(2, "big") + (792).to_bytes(2, "big") + b"\x04" + \
        b"".join(bytes([c + 1, 0x11, 0]) for c in range(4))
    sos = b"\xff\xda" + (6 + 2 * 4).to_bytes(2, "big") + b"\x04" + b"".join(bytes([c + 1, 0x00]) for c in range(4)) + b"\x00\x3f\x00"
    return soi + app14 + dqt + sof + sos + b"\x00" * 16 + b"\xff\xd9"


@pytest.mark.parametrize("transform,spec", [(0, "CMYK"), (2, "YCCK")])
def test_four_component_colour_spaces(lib, transform, spec):
    inst = make_instance(lib)
    st, info, _ = _info(lib, inst, _four_component(transform))
    assert st == 0
    assert info.sample_format == A.SAMPLEFORMAT_P_RGB and info.num_planes == 4
    assert info.color_spec == getattr(A, "COLORSPEC_" + spec)
    assert info.chroma_subsampling == A.SAMPLING_UNSUPPORTED
    for p in range(4):
        pi = info.plane_info[p]
        assert (pi.width, pi.height, pi.num_channels, pi.sample_type) == (792, 616, 1, A.SAMPLE_DATA_TYPE_UINT8)
    lib.nvimgcodecInstanceDestroy(inst)
