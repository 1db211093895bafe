"""Host logic (no GPU): the product's marker parser + Huffman entropy stage (nvimagecodec_amd/csrc/jpeg_syntax.cpp,
entropy_decode.cpp, reached through the C-ABI hipjpegGetImageInfo / hipjpegEntropyDecodeHost) against the oracle's
independently written entropy decoder, on every golden bitstream.  Integer work: exact equality."""
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, load_decode_case
from nvimagecodec_amd import _native as N
from nvimagecodec_amd import lowlevel

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    _M = json.load(_f)


@pytest.mark.parametrize("entry", _M["decode"], ids=lambda e: e["name"])
def test_entropy_stage_matches_oracle(entry):
    jpeg, _ = load_decode_case(entry)
    info = lowlevel.get_image_info(jpeg)
    oinfo = oracle.read_info(jpeg)
    assert (info["width"], info["height"], info["num_components"], info["sof_marker"]) == (
        oinfo["width"], oinfo["height"], oinfo["ncomp"], oinfo["sof"])
    assert info["blocks_w"] == oinfo["bw"] and info["blocks_h"] == oinfo["bh"]
    assert info["samp_w"] == oinfo["dw"] and info["samp_h"] == oinfo["dh"]
    assert info["restart_interval"] == oinfo["restart_interval"]
    coefs, qts = lowlevel.entropy_decode_host(jpeg)
    ocoefs, oqts = oracle.decode_coefficients(jpeg)
    for c in range(info["num_components"]):
        assert np.array_equal(qts[c], oqts[c])
        assert np.array_equal(coefs[c], ocoefs[c]), f"component {c}"


def test_subsampling_classification():
    names = {"444": 0, "422": 1, "420": 2, "440": 3, "411": 4, "410": 5, "gray": 6}
    seen = set()
    for e in _M["decode"]:
        if e["sub"] in seen:
            continue
        seen.add(e["sub"])
        jpeg, _ = load_decode_case(e)
        assert lowlevel.get_image_info(jpeg)["subsampling"] == names[e["sub"]], e["name"]
    assert seen == set(names)


def test_error_statuses():
    with pytest.raises(N.HipJpegError) as ei:
        lowlevel.get_image_info(b"\x89PNG\r\n\x1a\n" + b"\0" * 32)
    assert ei.value.status == 2  # BAD_JPEG
    jpeg, _ = load_decode_case(next(e for e in _M["decode"] if e["name"] == "s64x48_420_base_q90"))
    with pytest.raises(N.HipJpegError) as ei:
        lowlevel.entropy_decode_host(jpeg[: len(jpeg) * 2 // 3])
    assert ei.value.status in (4, 5)  # TRUNCATED / CORRUPT
    # flip bytes in the entropy-coded segment: must fail cleanly or decode to *something*, never crash
    rng = np.random.default_rng(5)
    for _ in range(50):
        b = bytearray(jpeg)
        for k in rng.integers(len(b) // 2, len(b) - 2, size=4):
            b[k] = int(rng.integers(0, 256))
        try:
            lowlevel.entropy_decode_host(bytes(b))
        except N.HipJpegError:
            pass


def test_arithmetic_and_12bit_are_unsupported():
    jpeg, _ = load_decode_case(next(e for e in _M["decode"] if e["name"] == "s64x48_420_base_q90"))
    b = bytearray(jpeg)
    i = b.find(b"\xff\xc0")
    b[i + 1] = 0xC9  # SOF9: arithmetic coding
    with pytest.raises(N.HipJpegError) as ei:
        lowlevel.get_image_info(bytes(b))
    assert ei.value.status == 3
    b = bytearray(jpeg)
    b[i + 4] = 12  # sample precision
    with pytest.raises(N.HipJpegError) as ei:
        lowlevel.get_image_info(bytes(b))
    assert ei.value.status == 3


def test_marker_walk_at_every_alignment_with_restart_markers_fill_bytes_and_trailing_data():
    """The parser's walk through the entropy-coded data takes 32 bytes at a time where the CPU has AVX2 (jpeg_syntax.cpp
    find_scan_end): FFs at the last byte of a chunk, RSTn markers straddling chunks, fill bytes (FF FF) in front of a marker, a
    file that ends inside a chunk, bytes behind EOI -- the scan's start is moved through 70 alignments by a comment segment of
    growing length, and what comes out must be the oracle's coefficients every time (and the GPU-algorithm emulation's, which
    uses the walk's restart positions)."""
    from nvimagecodec_amd.synth import synth_image
    img = synth_image(72, 40, seed=21)
    for sub, rst in (("420", 1), ("444", 3), ("420", 0)):
        base = oracle.encode(img, sub, 92, restart_interval=rst)
        want, _ = oracle.decode_coefficients(base)
        sos = base.index(b"\xff\xda")
        for shift in range(70):
            com = b"\xff\xfe" + (2 + shift).to_bytes(2, "big") + bytes([0x41] * shift)
            variants = [base[:sos] + com + base[sos:]]
            variants.append(variants[0] + b"\x00" * (shift % 37))                      # bytes behind EOI
            if rst:
                j = variants[0]
                k = j.index(b"\xff\xd0", sos)                                             # fill bytes in front of the first RST0
                variants.append(j[:k] + b"\xff\xff" + j[k:])
            for v in variants:
                got, _ = lowlevel.entropy_decode_host(v)
                assert all(np.array_equal(a, b) for a, b in zip(got, want)), (sub, rst, shift)
            emu, _ = lowlevel.entropy_decode_gpu_algorithm_host(variants[0])
            assert all(np.array_equal(a, b[: a.shape[0], : a.shape[1]]) for a, b in zip(emu, want)), (sub, rst, shift)
        # the file cut inside the scan: the walk ends at the end of the input, the decoder reports a truncated stream
        for cut in range(len(base) - 40, len(base) - 2, 3):
            with pytest.raises(N.HipJpegError):
                lowlevel.entropy_decode_host(base[:cut])


_SPARSE = [e for e in _M["decode"] if not e["progressive"] and e["width"] * e["height"] <= 130 * 70]


@pytest.mark.parametrize("entry", _SPARSE, ids=lambda e: e["name"])
def test_sparse_stream_expands_to_the_dense_blocks(entry):
    """Zero-run-compressed staging (round 3; csrc/entropy_decode.h): for host-decoded sequential pictures the stream of per-block records the
    device receives must hold exactly the coefficients of the dense decode -- restart intervals, every sampling, gray, odd sizes, MCU padding."""
    jpeg, _ = load_decode_case(entry)
    try:
        sparse, nbytes = lowlevel.entropy_decode_host_sparse(jpeg)
    except N.HipJpegError as e:
        assert "UNSUPPORTED" in str(e)   # several scans: stays dense
        return
    dense, _ = lowlevel.entropy_decode_host(jpeg)
    for c, (a, b) in enumerate(zip(sparse, dense)):
        assert np.array_equal(a, b), f"component {c}"
    assert nbytes < lowlevel.get_image_info(jpeg)["coef_bytes"] * 1.6


def test_sparse_stream_is_a_fraction_of_the_dense_blocks():
    from nvimagecodec_amd.synth import synth_image
    jpeg = oracle.encode(synth_image(640, 480, seed=3), "420", 90)
    _, nbytes = lowlevel.entropy_decode_host_sparse(jpeg)
    dense = lowlevel.get_image_info(jpeg)["coef_bytes"]
    assert nbytes < 0.45 * dense, (nbytes, dense)
    # progressive: the format does not apply
    prog = load_decode_case(next(e for e in _M["decode"] if e["progressive"]))[0]
    with pytest.raises(N.HipJpegError):
        lowlevel.entropy_decode_host_sparse(prog)
    # a truncated stream is named as such
    with pytest.raises(N.HipJpegError):
        lowlevel.entropy_decode_host_sparse(jpeg[: len(jpeg) // 2])
