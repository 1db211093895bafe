"""Files written from chosen coefficients (tests/helpers/jpeg_from_coefficients.py): the writer is pinned against the oracle's entropy
decoder (same coefficients back) and -- where Pillow is importable -- the oracle's pixels against the real libjpeg-turbo on these
files, whose dequantized values sit around the int16 edges of the library's SIMD IDCT."""
import io
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from tests.helpers import jpeg_from_coefficients as jc  # noqa: E402

CASES = [  # (name, sampling, quantizer, extreme magnitude)
    ("gray_q32_1023", [(1, 1)], 32, 1023),          # 32,736: the largest product the packed pass takes with Annex K tables
    ("gray_q33_1023", [(1, 1)], 33, 1023),          # 33,759: one past it -> butterflies on the 24-bit multiplier
    ("gray_q255_1023", [(1, 1)], 255, 1023),        # 260,865
    ("420_q32_1023", [(2, 2), (1, 1), (1, 1)], 32, 1023),
    ("420_q33_1023", [(2, 2), (1, 1), (1, 1)], 33, 1023),
    ("422_q16_1023", [(2, 1), (1, 1), (1, 1)], 16, 1023),
    ("444_q1_1023", [(1, 1), (1, 1), (1, 1)], 1, 1023),
    ("440_q32_1000", [(1, 2), (1, 1), (1, 1)], 32, 1000),
]


# the same layouts with every sample inside the range an 8-bit picture can take by a wide margin (|sample - 128| < 512, workspace values
# far inside int16): here libjpeg-turbo's C and SIMD IDCTs agree, and Pillow's build of it pins the oracle
IN_GAMUT = [
    ("gray_q32_e12", [(1, 1)], 32, 12),
    ("gray_q1_e380", [(1, 1)], 1, 380),
    ("420_q32_e12", [(2, 2), (1, 1), (1, 1)], 32, 12),
    ("420_q3_e120", [(2, 2), (1, 1), (1, 1)], 3, 120),
    ("422_q16_e24", [(2, 1), (1, 1), (1, 1)], 16, 24),
    ("444_q2_e190", [(1, 1), (1, 1), (1, 1)], 2, 190),
    ("440_q8_e48", [(1, 2), (1, 1), (1, 1)], 8, 48),
]


def make_case(name, sampling, q, extreme, width=83, height=61):
    rng = np.random.default_rng(sum(map(ord, name)))
    in_gamut = "_e" in name
    coefs = jc.random_coefficients(rng, width, height, sampling, extreme, small=max(1, 32 // q) if in_gamut else 3, dc=30 if in_gamut else 60)
    qt = [np.full(64, q, dtype=np.int32) for _ in sampling]
    for t in qt:
        t[0] = min(q, 16)  # a DC quantizer of its own: the DC term takes a separate route in the packed pass
    return jc.write_baseline(width, height, sampling, coefs, qt), coefs, qt


@pytest.mark.parametrize("case", CASES + IN_GAMUT, ids=[c[0] for c in CASES + IN_GAMUT])
def test_writer_round_trips_through_the_oracle_entropy_decoder(case):
    data, coefs, qt = make_case(*case)
    got, qts = oracle.decode_coefficients(data)
    for c in range(len(coefs)):
        assert np.array_equal(got[c].astype(np.int32), coefs[c]), case[0]
        assert np.array_equal(qts[c].astype(np.int32), qt[c])


@pytest.mark.parametrize("case", CASES + IN_GAMUT, ids=[c[0] for c in CASES + IN_GAMUT])
def test_oracle_matches_libjpeg_turbo_on_coefficient_files(case):
    """In gamut libjpeg-turbo's C and SIMD IDCTs agree.  Out of gamut (CASES) they part ways -- jidctint.c indexes the range-limit table
    modulo 1024 and keeps wide intermediates, the SIMD routines (what Pillow ships and what the reference links on x86-64) multiply and
    add in 16-bit lanes and saturate the workspace and the result.  The oracle's default variant restates the SIMD routine (round 3;
    pinned by tests/golden/gamut, tests/test_gamut_golden.py) and must equal the live library on all of them."""
    Image = pytest.importorskip("PIL.Image")
    if os.environ.get("JSIMD_FORCENONE"):
        pytest.skip("this process was told to run libjpeg-turbo's C routines")
    data, _, _ = make_case(*case)
    im = Image.open(io.BytesIO(data))
    im.draft = lambda *a, **k: None
    ref = np.asarray(im.convert("RGB") if im.mode != "L" else im)
    if case[1] == [(1, 1)]:
        assert np.array_equal(oracle.decode(data, oracle.FMT_GRAY), ref)
    else:
        assert np.array_equal(oracle.decode(data), ref)


def test_simd_and_c_idct_part_ways_out_of_gamut():
    """The two restatements in the oracle really are different functions on these files (and equal in gamut)."""
    data, _, _ = make_case(*CASES[0])
    simd = oracle.decode(data, oracle.FMT_GRAY)
    oracle.set_idct_variant(oracle.IDCT_C)
    try:
        plain = oracle.decode(data, oracle.FMT_GRAY)
        data2, _, _ = make_case(*IN_GAMUT[0])
        plain2 = oracle.decode(data2, oracle.FMT_GRAY)
    finally:
        oracle.set_idct_variant(oracle.IDCT_SIMD)
    assert not np.array_equal(simd, plain)
    assert np.array_equal(oracle.decode(data2, oracle.FMT_GRAY), plain2)
