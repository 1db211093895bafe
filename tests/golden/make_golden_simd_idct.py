#!/usr/bin/env python3
"""Golden vectors for streams whose samples leave the gamut, from the REAL libjpeg-turbo (via Pillow).  Dev-container only.

Why they exist.  The reference's CPU path decodes through jpeg_idct_islow as libjpeg-turbo dispatches it
(extensions/libjpeg_turbo/jpeg_mem.cpp:174-177); on x86-64 that is the SIMD routine (external/build_libjpeg-turbo.sh builds the
library with its defaults), and the SIMD routine works on 16-bit lanes: dequantization wraps, a block whose rows 1..7 are zero
takes a shortcut that shifts in 16 bits, in0 +- in4 / z3 / z4 are 16-bit sums, the workspace and the result saturate.  For every
stream an encoder can write this equals jidctint.c; for the files here -- written from chosen coefficients, valid baseline /
extended-sequential syntax -- it does not, and these vectors pin which of the two the oracle and the kernels must reproduce.

Every file is decoded three times in child processes: default dispatch (AVX2 here), JSIMD_FORCESSE2=1, JSIMD_FORCENONE=1 (the C
routine).  The first two must agree (asserted); their pixels are the golden output; the hash of the C routine's pixels is kept
beside it to pin the oracle's jidctint.c variant (oracle.set_idct_variant(1)).

Outputs: gamut/<name>.jpg, gamut/<name>.pix (H x W x 3 RGB, or H x W gray), manifest_gamut.json."""
import hashlib
import io
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.helpers import jpeg_from_coefficients as jc  # noqa: E402

GRAY = [(1, 1)]
S444 = [(1, 1), (1, 1), (1, 1)]
S420 = [(2, 2), (1, 1), (1, 1)]
S422 = [(2, 1), (1, 1), (1, 1)]
S440 = [(1, 2), (1, 1), (1, 1)]

# the files of tests/test_coefficient_files.py CASES: (name, sampling, quantizer, extreme magnitude)
RANDOM_CASES = [
    ("gray_q32_1023", GRAY, 32, 1023), ("gray_q33_1023", GRAY, 33, 1023), ("gray_q255_1023", GRAY, 255, 1023),
    ("420_q32_1023", S420, 32, 1023), ("420_q33_1023", S420, 33, 1023), ("422_q16_1023", S422, 16, 1023),
    ("444_q1_1023", S444, 1, 1023), ("440_q32_1000", S440, 32, 1000),
]


def random_case(name, sampling, q, extreme, width=83, height=61):
    """Same construction as tests/test_coefficient_files.py make_case (the test imports this function)."""
    rng = np.random.default_rng(sum(map(ord, name)))
    in_gamut = "_e" in name
    coefs = jc.random_coefficients(rng, width, height, sampling, extreme, small=max(1, 32 // q) if in_gamut else 3, dc=30 if in_gamut else 60)
    qt = [np.full(64, q, dtype=np.int32) for _ in sampling]
    for t in qt:
        t[0] = min(q, 16)
    return jc.write_baseline(width, height, sampling, coefs, qt), coefs, qt


def probe_files():
    """Deterministic probes, gray unless said otherwise; each returns (name, jpeg bytes)."""
    out = []

    def gray(name, blocks, q, width=None):
        nb = len(blocks)
        width = width or nb * 8
        a = np.zeros((1, -(-width // 8), 64), dtype=np.int32)
        for i, b in enumerate(blocks):
            for k, v in b.items():
                a[0, i, k] = v
        qt = np.asarray(q, dtype=np.int32) if np.ndim(q) else np.full(64, q, dtype=np.int32)
        out.append((name, jc.write_baseline(width, 8, GRAY, [a], [qt])))

    # one coefficient per block, dequantized value walking across the int16 edges (pmullw wrap, packssdw saturation)
    for pos in (1, 8, 9, 27, 63):
        blocks = []
        for c in (127, 128, 255, 256, 511, 512, 513, 767, 1023, -128, -256, -512, -513, -1023):
            blocks.append({0: 0, pos: c})
        gray(f"one_coef_k{pos}_q64", blocks, 64)
        gray(f"one_coef_k{pos}_q255", blocks, 255)
    # rows 1..7 all zero: the SIMD routine's shortcut (row 0 << 2 in 16 bits) against the same blocks with one +-1 in row 1
    row0 = []
    for dc, ac in ((600, 0), (1000, 0), (-1000, 0), (100, 1023), (100, -1023), (-900, 700), (1023, 1023), (0, 512), (0, -512), (255, 256)):
        row0.append({0: dc, 3: ac})
        row0.append({0: dc, 3: ac, 8: 1})
        row0.append({0: dc, 5: ac, 7: -ac})
    for q in (8, 9, 16, 32, 33, 64, 128, 255):
        gray(f"row0_only_q{q}", row0, q)
    # DC predictor leaving int16: +2047 per block (wraps after 16 blocks), then back down
    up = [{0: 2047 * (i + 1)} for i in range(40)] + [{0: 2047 * 40 - 2047 * (i + 1), 1: 3} for i in range(60)]
    gray("dc_walk_q1", up, 1)
    gray("dc_walk_q16", up, 16)
    # sums of two workspace values that wrap in 16 bits in pass 2 (in0 +- in4, in7 + in3, in5 + in1 of a ROW)
    rng = np.random.default_rng(20261005)
    blocks = []
    for _ in range(48):
        b = {0: int(rng.integers(-1000, 1001))}
        for k in rng.choice(np.arange(1, 64), size=int(rng.integers(1, 6)), replace=False):
            b[int(k)] = int(rng.choice([-1023, 1023, -1000, 700, 512, -512]))
        blocks.append(b)
    gray("pass2_wraps_q255", blocks, 255)
    gray("pass2_wraps_q90", blocks, 90)
    gray("pass2_wraps_q31", blocks, 31)
    # 16-bit quantization tables (extended sequential, SOF1)
    small = [{0: int(rng.integers(-5, 6)), int(rng.integers(1, 64)): int(rng.integers(-3, 4)), int(rng.integers(1, 64)): int(rng.integers(-9, 10))} for _ in range(32)]
    for q in (256, 4096, 32767, 32768, 40000, 65535):
        gray(f"q16bit_{q}", small, q)
    qmix = rng.integers(1, 65536, size=64)
    gray("q16bit_mixed", small, qmix)
    # ragged width: the right-most block is cropped
    gray("ragged_one_coef_q255", [{0: 0, 9: c} for c in (1023, -1023, 512, -512, 700)], 255, width=37)
    # colour: out-of-gamut chroma and luma through upsampling and colour conversion, every sampling the kernels specialise
    for name, sampling in (("420", S420), ("422", S422), ("444", S444), ("440", S440)):
        r2 = np.random.default_rng(sum(map(ord, name)) + 17)
        coefs = jc.random_coefficients(r2, 70, 50, sampling, 1023, small=40, dc=900)
        qt = [r2.integers(1, 256, size=64).astype(np.int32) for _ in sampling]
        out.append((f"colour_{name}_qrand", jc.write_baseline(70, 50, sampling, coefs, qt)))
    return out


def decode_child(paths, env_extra):
    """Decode the files in a child process (the SIMD dispatch is chosen once per process from the environment)."""
    code = ("import sys, io, numpy as np\nfrom PIL import Image\n"
            "for p in sys.argv[1:]:\n"
            "    im = Image.open(p)\n"
            "    a = np.asarray(im if im.mode == 'L' else im.convert('RGB'))\n"
            "    open(p + '.out', 'wb').write(np.ascontiguousarray(a).tobytes())\n")
    env = dict(os.environ)
    for k in ("JSIMD_FORCENONE", "JSIMD_FORCESSE2", "JSIMD_FORCEAVX2"):
        env.pop(k, None)
    env.update(env_extra)
    subprocess.check_call([sys.executable, "-c", code] + paths, env=env)
    return [open(p + ".out", "rb").read() for p in paths]


def main():
    from PIL import Image, features
    assert features.check_feature("libjpeg_turbo")
    outdir = os.path.join(HERE, "gamut")
    os.makedirs(outdir, exist_ok=True)
    files = [(n, random_case(n, s, q, e)[0]) for n, s, q, e in RANDOM_CASES] + probe_files()
    paths = []
    for name, data in files:
        p = os.path.join(outdir, name + ".jpg")
        with open(p, "wb") as f:
            f.write(data)
        paths.append(p)
    simd = decode_child(paths, {})
    sse2 = decode_child(paths, {"JSIMD_FORCESSE2": "1"})
    plain = decode_child(paths, {"JSIMD_FORCENONE": "1"})
    entries, differ = [], 0
    for (name, data), p, a, b, c in zip(files, paths, simd, sse2, plain):
        assert a == b, f"{name}: the AVX2 and SSE2 routines disagree"
        os.remove(p + ".out")
        im = Image.open(io.BytesIO(data))
        with open(os.path.join(outdir, name + ".pix"), "wb") as f:
            f.write(a)
        differ += a != c
        entries.append(dict(name=name, width=im.width, height=im.height, mode=im.mode, simd_sha256=hashlib.sha256(a).hexdigest(),
                            c_sha256=hashlib.sha256(c).hexdigest(), simd_equals_c=a == c))
    with open(os.path.join(HERE, "manifest_gamut.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden_simd_idct.py", "pillow": Image.__version__, "libjpeg_turbo": features.version("libjpeg_turbo"),
                   "dispatch": "default (AVX2) == JSIMD_FORCESSE2=1; c_sha256 from JSIMD_FORCENONE=1", "gamut": entries}, f, indent=1)
    print(len(entries), "vectors;", differ, "on which the SIMD and the C routine give different pictures")


if __name__ == "__main__":
    main()
