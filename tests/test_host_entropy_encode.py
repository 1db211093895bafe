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
with open(os.path.join(GOLDEN, "manifest_encode_prog.json")) as _f:
    _MP = json.load(_f)["encode_progressive"]


def load_progressive_case(entry):
    rgb = np.fromfile(os.path.join(GOLDEN, entry["input"]), dtype=np.uint8).reshape(entry["height"], entry["width"], 3)
    with open(os.path.join(GOLDEN, "encode_prog", entry["name"] + ".jpg"), "rb") as f:
        return rgb, f.read()


@pytest.mark.parametrize("entry", _MP, ids=lambda e: e["name"])
def test_progressive_coder_reproduces_libjpeg_turbo_files(entry):
    """SOF2 output: jcparam.c's simple progression, jcphuff.c's DC/AC first/refinement coding with EOB runs and buffered
    correction bits, per-scan optimal tables, restart intervals -- the WHOLE file equals libjpeg-turbo's (tests/golden/
    make_golden_encode_progressive.py), markers included."""
    rgb, jpeg = load_progressive_case(entry)
    coefs, _ = oracle.forward(rgb, entry["sub"], entry["quality"])
    mine = lowlevel.encode_from_coefficients_host(entry["width"], entry["height"], coefs, entry["sub"], entry["quality"],
                                                  restart_interval=entry["restart"], progressive=True)
    assert mine == jpeg
    # and it is the same picture as the baseline file
    c1, q1 = oracle.decode_coefficients(mine)
    base = oracle.encode(rgb, entry["sub"], entry["quality"])
    c2, q2 = oracle.decode_coefficients(base)
    assert all(np.array_equal(a, b) for a, b in zip(c1, c2)) and all(np.array_equal(a, b) for a, b in zip(q1, q2))
    assert np.array_equal(oracle.decode(mine), oracle.decode(base))


def test_progressive_coder_long_runs_and_extreme_coefficients():
    """Cases the goldens do not reach: an all-zero picture (one EOB run over every block, split at 32767), a picture whose
    refinement scans carry more than 937 buffered correction bits in one run (jcphuff.c MAX_CORR_BITS), the largest coefficients
    a quantizer of 1 produces.  No libjpeg-turbo bytes for hand-made coefficients here: the file must decode (oracle) to exactly
    the coefficients that went in."""
    rng = np.random.default_rng(7)
    cases = []
    z = [np.zeros((184, 184, 64), np.int16)]  # 33,856 blocks > 32,767
    cases.append((184 * 8, 184 * 8, "gray", z))
    c = np.zeros((8, 64, 64), np.int16)
    c[:, :, 1:] = rng.integers(2, 4, size=(8, 64, 63))  # every AC coefficient nonzero before the last scans: correction bits only
    cases.append((512, 64, "gray", [c]))
    big = rng.integers(-1023, 1024, size=(4, 4, 64)).astype(np.int16)
    big[:, :, 0] = rng.integers(-1024, 1017, size=(4, 4))
    cases.append((32, 32, "gray", [big]))
    sparse = [np.zeros((6, 6, 64), np.int16), np.zeros((3, 3, 64), np.int16), np.zeros((3, 3, 64), np.int16)]
    sparse[0][::2, ::3, 40] = 1
    sparse[0][1, 1, 63] = -1
    sparse[1][2, 2, 17] = -5
    sparse[2][0, 0, 0] = 3
    cases.append((40, 40, "420", sparse))
    for (w, h, sub, coefs) in cases:
        for rst in (0, 3):
            mine = lowlevel.encode_from_coefficients_host(w, h, coefs, sub, 100, restart_interval=rst, progressive=True)
            back, _ = oracle.decode_coefficients(mine)
            for a, b in zip(coefs, back):
                assert np.array_equal(a, b[: a.shape[0], : a.shape[1]]), (w, h, sub, rst)


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
