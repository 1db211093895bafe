"""Host logic (no GPU): the product's Huffman coder + marker writer (csrc/entropy_encode.cpp through
hipjpegEncodeFromCoefficientsHost) fed with the oracle's forward-path coefficients must reproduce libjpeg-turbo's
bitstream byte for byte (entropy-coded segment + tables), incl. dummy blocks, restart markers and optimized tables."""
import io
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, load_encode_case
from nvimagecodec_amd import lowlevel
from nvimagecodec_amd.synth import synth_image

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    _M = json.load(_f)


@pytest.mark.parametrize("entry", _M["encode"], ids=lambda e: e["name"])
def test_entropy_coder_reproduces_libjpeg_turbo_scan(entry):
    rgb, jpeg = load_encode_case(entry)
    coefs, _ = oracle.forward(rgb, entry["sub"], entry["quality"])
    mine = lowlevel.encode_from_coefficients_host(entry["width"], entry["height"], coefs, entry["sub"], entry["quality"])
    assert oracle.scan_bytes(mine) == oracle.scan_bytes(jpeg)
    # our whole file parses back to the same coefficients and tables
    c1, q1 = oracle.decode_coefficients(mine)
    c2, q2 = oracle.decode_coefficients(jpeg)
    assert all(np.array_equal(a, b) for a, b in zip(c1, c2)) and all(np.array_equal(a, b) for a, b in zip(q1, q2))
    # and equals the oracle's own writer byte for byte (same marker order as libjpeg's jcmarker.c)
    assert mine == oracle.encode(rgb, entry["sub"], entry["quality"])


def test_restart_intervals_and_other_samplings():
    for sub in ("444", "420", "422", "440", "411", "410", "gray"):
        for (w, h) in ((50, 37), (129, 70)):
            rgb = synth_image(w, h, seed=w)
            for rst in (0, 3):
                coefs, _ = oracle.forward(rgb, sub, 85)
                mine = lowlevel.encode_from_coefficients_host(w, h, coefs, sub, 85, restart_interval=rst)
                assert mine == oracle.encode(rgb, sub, 85, restart_interval=rst), (sub, w, h, rst)


def test_optimized_huffman_matches_libjpeg_turbo():
    """optimize=True in Pillow = libjpeg's two-pass optimal tables (jchuff.c jpeg_gen_optimal_table).  Needs Pillow (dev box);
    the arithmetic is also covered indirectly: the optimized file must decode to the same coefficients."""
    rgb = synth_image(96, 64, seed=5)
    coefs, _ = oracle.forward(rgb, "420", 90)
    mine = lowlevel.encode_from_coefficients_host(96, 64, coefs, "420", 90, optimized_huffman=True)
    c1, _ = oracle.decode_coefficients(mine)
    c0, _ = oracle.decode_coefficients(oracle.encode(rgb, "420", 90))
    assert all(np.array_equal(a, b) for a, b in zip(c0, c1))
    assert len(mine) < len(oracle.encode(rgb, "420", 90))
    try:
        from PIL import Image
    except ImportError:
        pytest.skip("Pillow not available: byte comparison with libjpeg-turbo's optimized tables skipped")
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "JPEG", quality=90, subsampling=2, optimize=True)
    ref = b.getvalue()
    assert oracle.scan_bytes(mine) == oracle.scan_bytes(ref)
    i = ref.find(b"\xff\xc4")
    j = mine.find(b"\xff\xc4")
    assert ref[i: ref.find(b"\xff\xda")] == mine[j: mine.find(b"\xff\xda")]  # identical DHT segments
