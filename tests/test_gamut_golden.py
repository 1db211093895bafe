"""Out-of-gamut vectors from the real libjpeg-turbo (tests/golden/make_golden_simd_idct.py): files written from chosen coefficients
whose samples leave the 8-bit gamut, where the library's SIMD jpeg_idct_islow (what the reference's CPU path runs on x86-64:
extensions/libjpeg_turbo/jpeg_mem.cpp:174-177, external/build_libjpeg-turbo.sh:36-39) and jidctint.c part ways.  The oracle's default
variant must equal the SIMD pixels bit for bit; its jidctint.c variant must equal what JSIMD_FORCENONE=1 gave."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "manifest_gamut.json")))
ENTRIES = MANIFEST["gamut"]


def load(entry):
    data = open(os.path.join(GOLD, "gamut", entry["name"] + ".jpg"), "rb").read()
    pix = np.fromfile(os.path.join(GOLD, "gamut", entry["name"] + ".pix"), dtype=np.uint8)
    shape = (entry["height"], entry["width"]) if entry["mode"] == "L" else (entry["height"], entry["width"], 3)
    return data, pix.reshape(shape)


def test_manifest_is_consistent():
    assert len(ENTRIES) >= 40
    assert sum(not e["simd_equals_c"] for e in ENTRIES) >= 30  # these files are there BECAUSE the two routines differ on them
    for e in ENTRIES:
        _, pix = load(e)
        assert hashlib.sha256(pix.tobytes()).hexdigest() == e["simd_sha256"]


@pytest.mark.parametrize("entry", ENTRIES, ids=[e["name"] for e in ENTRIES])
def test_oracle_equals_the_simd_routine(entry):
    data, pix = load(entry)
    got = oracle.decode(data, oracle.FMT_GRAY if entry["mode"] == "L" else oracle.FMT_RGB)
    assert np.array_equal(got, pix), "%d of %d samples differ" % (int((got != pix).sum()), pix.size)


@pytest.mark.parametrize("entry", ENTRIES, ids=[e["name"] for e in ENTRIES])
def test_oracle_c_variant_equals_jidctint(entry):
    data, _ = load(entry)
    oracle.set_idct_variant(oracle.IDCT_C)
    try:
        got = oracle.decode(data, oracle.FMT_GRAY if entry["mode"] == "L" else oracle.FMT_RGB)
    finally:
        oracle.set_idct_variant(oracle.IDCT_SIMD)
    assert hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest() == entry["c_sha256"]
