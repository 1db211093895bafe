"""Four-component JPEGs (CMYK / YCCK).  Reference: the GPU plugin accepts them (extensions/nvjpeg/cuda_decoder.cpp:85-89) and
the CPU path decodes them with libjpeg-turbo's JCS_CMYK output followed by its own CMYK -> RGB step
(extensions/libjpeg_turbo/jpeg_mem.cpp:168-172, 292-337).
Pinned by libjpeg-turbo vectors (tests/golden/make_golden_cmyk.py): the CMYK samples, including the YCCK conversion and the
upsampling of subsampled components.  NOT pinned by any executable (the reference cannot be built here): the final
CMYK -> RGB arithmetic -- restated from the cited lines in the oracle and in the kernel."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN

with open(os.path.join(GOLDEN, "manifest_cmyk.json")) as _f:
    _M = json.load(_f)["cmyk"]


def _load(e):
    with open(os.path.join(GOLDEN, "cmyk", e["name"] + ".jpg"), "rb") as f:
        jpeg = f.read()
    ref = np.fromfile(os.path.join(GOLDEN, "cmyk", e["name"] + ".cmyk"), dtype=np.uint8).reshape(e["height"], e["width"], 4)
    return jpeg, ref


@pytest.mark.parametrize("entry", _M, ids=lambda e: e["name"])
def test_oracle_cmyk_samples_equal_libjpeg_turbo(entry):
    jpeg, ref = _load(entry)
    assert np.array_equal(oracle.decode_cmyk(jpeg), ref)


def _reference_rgb(cmyk, adobe):
    """extensions/libjpeg_turbo/jpeg_mem.cpp:303-313, on libjpeg-turbo's own samples"""
    c, m, y, k = [cmyk[:, :, i].astype(np.int32) for i in range(4)]
    if adobe:
        return np.stack([(k * c) // 255, (k * m) // 255, (k * y) // 255], axis=2).astype(np.uint8)
    return np.stack([(255 - k) * (255 - c) // 255, (255 - k) * (255 - m) // 255, (255 - k) * (255 - y) // 255], axis=2).astype(np.uint8)


@pytest.mark.parametrize("entry", _M, ids=lambda e: e["name"])
def test_oracle_rgb_is_the_reference_formula_on_libjpeg_turbo_samples(entry):
    jpeg, ref = _load(entry)
    rgb = _reference_rgb(ref, entry["kind"] != "plain")
    assert np.array_equal(oracle.decode(jpeg), rgb)
    assert np.array_equal(oracle.decode(jpeg, oracle.FMT_BGR), rgb[:, :, ::-1])
    r, g, b = [rgb[:, :, i].astype(np.float32) for i in range(3)]
    gray = (np.float32(0.299) * r + np.float32(0.587) * g + np.float32(0.114) * b).astype(np.uint8)
    assert np.array_equal(oracle.decode(jpeg, oracle.FMT_GRAY), gray)


def test_parser_and_host_entropy_stage_take_four_components():
    from nvimagecodec_amd import lowlevel
    for e in _M[:6]:
        jpeg, _ = _load(e)
        info = lowlevel.get_image_info(jpeg)
        assert info["num_components"] == 4 and info["color_model"] == (4 if e["kind"] == "adobe2" else 3)
        coefs, _ = lowlevel.entropy_decode_host(jpeg)
        ref, _ = oracle.decode_coefficients(jpeg)
        assert len(coefs) == 4 and all(np.array_equal(a, b) for a, b in zip(coefs, ref))


@pytest.fixture(scope="module")
def dec():
    import torch
    assert torch.cuda.is_available()
    from nvimagecodec_amd.lowlevel import BatchDecoder
    d = BatchDecoder(device=0, num_threads=4)
    yield d
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("gpu_huffman", [False, True], ids=["host_entropy", "gpu_entropy"])
@pytest.mark.parametrize("fmt", ["rgb", "bgr", "rgb_planar", "y"])
def test_gpu_decodes_every_cmyk_golden(dec, fmt, gpu_huffman):
    import torch
    cases = [_load(e) for e in _M]
    outs, st = dec.decode([c[0] for c in cases], fmt=fmt, gpu_huffman=gpu_huffman)
    torch.cuda.synchronize()
    assert all(s == 0 for s in st)
    for e, (jpeg, cmyk), o in zip(_M, cases, outs):
        rgb = _reference_rgb(cmyk, e["kind"] != "plain")
        if fmt == "y":
            ref = oracle.decode(jpeg, oracle.FMT_GRAY)
        elif fmt == "bgr":
            ref = rgb[:, :, ::-1]
        elif fmt == "rgb_planar":
            ref = rgb.transpose(2, 0, 1)
        else:
            ref = rgb
        assert np.array_equal(o.cpu().numpy(), ref), (e["name"], fmt)


with open(os.path.join(GOLDEN, "manifest_plain.json")) as _f:
    _PLAIN = json.load(_f)["cmyk"]


@pytest.mark.parametrize("entry", _PLAIN, ids=lambda e: e["name"])
def test_oracle_cmyk_samples_without_fancy_upsampling_equal_libjpeg_turbo(entry):
    """do_fancy_upsampling = FALSE on four-component files: the real library's samples (tests/golden/make_golden_plain_upsampling.py)"""
    with open(os.path.join(GOLDEN, "cmyk", entry["name"] + ".jpg"), "rb") as f:
        jpeg = f.read()
    assert hashlib.sha256(np.ascontiguousarray(oracle.decode_cmyk(jpeg, fancy=False)).tobytes()).hexdigest() == entry["plain_cmyk_sha256"]


@pytest.mark.gpu
def test_cmyk_without_fancy_upsampling_and_as_raw_planes(dec):
    import torch
    for e in _PLAIN:
        with open(os.path.join(GOLDEN, "cmyk", e["name"] + ".jpg"), "rb") as f:
            jpeg = f.read()
        outs, _ = dec.decode([jpeg], fmt="rgb", fancy=False)
        torch.cuda.synchronize()
        # the oracle's samples are pinned by the real library's (test above); its RGB is the reference's formula on them
        assert np.array_equal(outs[0].cpu().numpy(), oracle.decode(jpeg, oracle.FMT_RGB, fancy=False)), e["name"]
    e = next(x for x in _M if x["subsampled"] and x["kind"] == "adobe2" and not x["progressive"])
    jpeg, _ = _load(e)
    out = torch.zeros((e["height"], e["width"]), dtype=torch.uint8, device="cuda")
    _, st = dec.decode([jpeg], fmt="yuv_planar", outs=[[out, out, out]], check=False)
    assert st == [3]   # raw planes of a four-component frame: UNSUPPORTED, the chain moves on


@pytest.mark.gpu
def test_cmyk_through_the_plugin_api(dec):
    import torch
    from nvimagecodec_amd import api
    cases = [_load(e) for e in _M[:8]]
    with api.Decoder(max_num_cpu_threads=2) as d:
        imgs = d.decode([c[0] for c in cases])
        torch.cuda.synchronize()
        for e, (jpeg, cmyk), im in zip(_M, cases, imgs):
            assert im is not None, e["name"]
            assert np.array_equal(np.asarray(im.cpu()._array), _reference_rgb(cmyk, e["kind"] != "plain")), e["name"]
