"""GPU parity tests for the decode hot path, through the C-ABI (hipjpegDecodeBatch): HIP kernels vs the CPU oracle and
the libjpeg-turbo golden vectors.  Integer/byte work => bit-exact (tolerance 0).  The reference's own tests for this
path compare with +-1 (test/extensions/common_ext_decoder_test.h:152-182) or memcmp against nvJPEG
(test/extensions/nvjpeg_ext_decoder_test.cpp:108-145)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, load_decode_case
from nvimagecodec_amd.synth import synth_image

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    _M = json.load(_f)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def dec():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from nvimagecodec_amd.lowlevel import BatchDecoder
    d = BatchDecoder(device=0, num_threads=4)
    yield d
    d.close()


def _sync():
    import torch
    torch.cuda.synchronize()


def test_all_golden_vectors_one_batch(dec):
    """Every golden bitstream (all samplings, baseline+progressive, restart intervals, odd sizes) in ONE mixed-shape batch."""
    cases = [load_decode_case(e) for e in _M["decode"]]
    jpegs = [c[0] for c in cases]
    outs, statuses = dec.decode(jpegs, fmt="rgb", fancy=True)
    _sync()
    assert all(s == 0 for s in statuses)
    for e, (jpeg, rgb), o in zip(_M["decode"], cases, outs):
        got = o.cpu().numpy()
        assert _sha(got) == e["rgb_sha256"], e["name"]
        if rgb is not None:
            assert np.array_equal(got, rgb), e["name"]


@pytest.mark.parametrize("fmt", ["bgr", "rgb_planar", "bgr_planar", "y", "yuv_planar"])
def test_output_formats(dec, fmt):
    names = ["s50x37_420_base_q90", "s33x65_422_prog_q50", "s64x48_444_base_q90", "o50x37_440_base_q90", "o64x48_411_base_q90",
             "s17x13_gray_base_q90", "r130x70_420_base_rst7", "c1_640x480_444_base_q90"]
    entries = [next(e for e in _M["decode"] if e["name"] == n) for n in names]
    jpegs = [load_decode_case(e)[0] for e in entries]
    outs, _ = dec.decode(jpegs, fmt=fmt, fancy=True)
    _sync()
    for e, j, o in zip(entries, jpegs, outs):
        if fmt == "yuv_planar":
            ref = oracle.decode_planes(j)
            assert len(o) == len(ref)
            for a, b in zip(o, ref):
                assert np.array_equal(a.cpu().numpy(), b), e["name"]
            continue
        if fmt == "y":
            ref = oracle.decode(j, oracle.FMT_GRAY)
        else:
            ref = oracle.decode(j, oracle.FMT_BGR if fmt.startswith("bgr") else oracle.FMT_RGB)
            if fmt.endswith("planar"):
                ref = ref.transpose(2, 0, 1)
        assert np.array_equal(o.cpu().numpy(), ref), (e["name"], fmt)


@pytest.mark.parametrize("fmt", ["rgb", "bgr", "rgb_planar"])
def test_tile_shapes_at_the_right_edge(dec, fmt):
    """The fused luma kernel covers a block row with 32-block tiles and, where the ragged rest is at most 16 blocks, with
    narrow 16 x 8-block tiles (two block rows per wave).  Widths around every boundary (rest of 1, 8, 15, 16, 17, 31 blocks,
    ragged pixels inside the last block), heights that end inside the first / second block row of a narrow tile, every
    sampling the kernel fuses, interleaved and planar outputs (the common-case and the generic kernel flavour)."""
    cases = []
    for blocks_w in (1, 16, 17, 32, 33, 40, 47, 48, 49, 63, 80):
        w = blocks_w * 8 - (3 if blocks_w % 2 else 0)
        for h, sub in ((8 * 9 + 5, "420"), (8 * 5 + 1, "422"), (8 * 8, "444"), (8 * 17 - 2, "gray")):
            img = synth_image(w, h, seed=w + 7 * h)
            cases.append(oracle.encode(img if sub != "gray" else img[:, :, 1].copy(), sub, 88))
    outs, statuses = dec.decode(cases, fmt=fmt, fancy=True)
    _sync()
    assert all(st == 0 for st in statuses)
    for j, o in zip(cases, outs):
        ref = oracle.decode(j, oracle.FMT_BGR if fmt == "bgr" else oracle.FMT_RGB)
        if fmt.endswith("planar"):
            ref = ref.transpose(2, 0, 1)
        assert np.array_equal(o.cpu().numpy(), ref)


def test_fancy_upsampling_off(dec):
    """fancy_upsampling=0 (extensions/libjpeg_turbo/jpeg_mem.cpp:166 do_fancy_upsampling = FALSE): every golden file against the real
    library's pixels (tests/golden/manifest_plain.json, made by driving libjpeg-turbo's C API: Pillow has no such switch), both entropy
    stages."""
    with open(os.path.join(GOLDEN, "manifest_plain.json")) as f:
        plain = json.load(f)["decode"]
    jpegs = [open(os.path.join(GOLDEN, "decode", e["name"] + ".jpg"), "rb").read() for e in plain]
    for gh in (False, True):
        outs, statuses = dec.decode(jpegs, fmt="rgb", fancy=False, gpu_huffman=gh)
        _sync()
        assert all(s == 0 for s in statuses)
        for e, j, o in zip(plain, jpegs, outs):
            assert _sha(o.cpu().numpy()) == e["plain_rgb_sha256"], (e["name"], gh)


def test_pitched_and_unaligned_outputs(dec):
    import torch
    e = next(e for e in _M["decode"] if e["name"] == "r130x70_420_base_rst1")
    jpeg, rgb = load_decode_case(e)
    h, w = e["height"], e["width"]
    for pitch, offset in ((w * 3 + 13, 0), (w * 3 + 3, 5), (512, 1)):
        buf = torch.full((h * pitch + 64,), 0xAB, dtype=torch.uint8, device="cuda")
        view = torch.as_strided(buf, (h, w, 3), (pitch, 3, 1), storage_offset=offset)
        dec.decode([jpeg], fmt="rgb", outs=[view])
        _sync()
        host = buf.cpu().numpy()
        got = np.lib.stride_tricks.as_strided(host[offset:], (h, w, 3), (pitch, 3, 1))
        assert np.array_equal(got, rgb)
        # bytes outside the image rows stay untouched
        mask = np.ones(host.shape, dtype=bool)
        for y in range(h):
            mask[offset + y * pitch: offset + y * pitch + w * 3] = False
        assert np.all(host[mask] == 0xAB)


def test_bad_images_in_a_batch_do_not_poison_neighbours(dec):
    good = [load_decode_case(next(e for e in _M["decode"] if e["name"] == n)) for n in ("s64x48_420_base_q90", "s50x37_444_prog_q90")]
    trunc = good[0][0][: len(good[0][0]) // 2]
    jpegs = [good[0][0], b"not a jpeg at all", trunc, good[1][0]]
    outs = dec.allocate_outputs(jpegs)
    outs, statuses = dec.decode(jpegs, outs=outs, check=False)
    _sync()
    assert statuses[0] == 0 and statuses[3] == 0
    assert statuses[1] == 2           # BAD_JPEG
    assert statuses[2] in (4, 5)      # TRUNCATED / CORRUPT
    assert np.array_equal(outs[0].cpu().numpy(), good[0][1])
    assert np.array_equal(outs[3].cpu().numpy(), good[1][1])


def _encode_inputs(shapes_subs, quality=90):
    """Inputs for the full-size cases, produced on the box by the oracle's (libjpeg-turbo-pinned) encoder."""
    return [oracle.encode(synth_image(w, h, seed=s), sub, quality) for (w, h, sub, s) in shapes_subs]


def test_config1_batch_1080p_420(dec):
    """BASELINE.json configs[1] shape (reduced batch for test time): 1920x1080 4:2:0 baseline -> interleaved RGB u8."""
    jpegs = _encode_inputs([(1920, 1080, "420", s) for s in range(3)])
    outs, _ = dec.decode(jpegs * 4, fmt="rgb")
    _sync()
    refs = [oracle.decode(j) for j in jpegs]
    for i, o in enumerate(outs):
        assert np.array_equal(o.cpu().numpy(), refs[i % 3])


def test_config3_mixed_shapes_420_422(dec):
    """BASELINE.json configs[3] shape on one GPU: mixed 480p..4K, 4:2:0 / 4:2:2, one batched launch."""
    spec = [(640, 480, "420", 11), (1280, 720, "422", 12), (1920, 1080, "422", 13), (2560, 1440, "420", 14), (3840, 2160, "420", 15),
            (641, 479, "422", 16), (1283, 721, "420", 17)]
    jpegs = _encode_inputs(spec)
    outs, _ = dec.decode(jpegs, fmt="rgb")
    _sync()
    for j, o in zip(jpegs, outs):
        assert np.array_equal(o.cpu().numpy(), oracle.decode(j))


def test_config4_progressive_444_planar(dec):
    """BASELINE.json configs[4] shape: progressive 4:4:4 -> planar output.  Input: committed Pillow progressive vector
    plus idempotence across repeated launches."""
    e = next(e for e in _M["decode"] if e["name"] == "c5_640x360_444_prog_q90")
    jpeg, _ = load_decode_case(e)
    outs, _ = dec.decode([jpeg] * 8, fmt="rgb_planar")
    _sync()
    ref = oracle.decode(jpeg).transpose(2, 0, 1)
    assert _sha(ref.transpose(1, 2, 0)) == e["rgb_sha256"]
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), ref)


def test_extreme_coefficients(dec):
    """Quality 1 tables (quantizers up to 255) on a saturated checkerboard push dequantized values far beyond what a photograph holds;
    the kernels must still match the oracle bit for bit."""
    img = np.zeros((64, 64, 3), dtype=np.uint8)
    img[::2, ::2] = 255
    img[1::2, 1::2] = 255
    for q in (1, 3, 100):
        for sub in ("444", "420"):
            j = oracle.encode(img, sub, q)
            outs, _ = dec.decode([j])
            _sync()
            assert np.array_equal(outs[0].cpu().numpy(), oracle.decode(j)), (q, sub)


def test_coefficient_files(dec):
    """Files written from chosen coefficients (tests/test_coefficient_files.py): dequantized values around and far beyond the int16
    edges of the IDCT's 16-bit lanes, and in-gamut files -- GPU and host entropy stages, interleaved / planar / gray outputs, bit-exact
    against the oracle (itself pinned on these very files by the live libjpeg-turbo, tests/test_coefficient_files.py)."""
    from test_coefficient_files import CASES, IN_GAMUT, make_case
    for case in CASES + IN_GAMUT:
        data = make_case(*case)[0]
        for gh in (True, False):
            for fmt in ("rgb", "rgb_planar", "y"):
                outs, st = dec.decode([data], fmt=fmt, gpu_huffman=gh)
                _sync()
                got = outs[0].cpu().numpy()
                ref = oracle.decode(data, oracle.FMT_GRAY if fmt == "y" else oracle.FMT_RGB)
                if fmt == "rgb_planar":
                    ref = ref.transpose(2, 0, 1)
                assert np.array_equal(got, ref), (case[0], gh, fmt)


def _gamut_entries():
    import json
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return gold, json.load(open(os.path.join(gold, "manifest_gamut.json")))["gamut"]


@pytest.mark.parametrize("gpu_huffman", [True, False], ids=["gpu_entropy", "host_entropy"])
def test_out_of_gamut_vectors_from_libjpeg_turbo(dec, gpu_huffman):
    """tests/golden/gamut: 43 valid baseline / extended-sequential files whose samples leave the gamut, decoded by the real libjpeg-turbo
    (SIMD dispatch, what the reference's CPU path runs).  One mixed batch, every output layout the kernels specialise; the golden pixels
    themselves are the reference here, not the oracle."""
    gold, entries = _gamut_entries()
    datas = [open(os.path.join(gold, "gamut", e["name"] + ".jpg"), "rb").read() for e in entries]
    refs = []
    for e in entries:
        pix = np.fromfile(os.path.join(gold, "gamut", e["name"] + ".pix"), dtype=np.uint8)
        refs.append(pix.reshape((e["height"], e["width"]) if e["mode"] == "L" else (e["height"], e["width"], 3)))
    colour = [i for i, e in enumerate(entries) if e["mode"] != "L"]
    gray = [i for i, e in enumerate(entries) if e["mode"] == "L"]
    # colour files -> interleaved RGB, BGR, planar RGB; gray files -> Y and (replicated) RGB
    for idx, fmt, conv in ((colour, "rgb", lambda r: r), (colour, "bgr", lambda r: r[:, :, ::-1]), (colour, "rgb_planar", lambda r: r.transpose(2, 0, 1)),
                           (gray, "y", lambda r: r), (gray, "rgb", lambda r: np.repeat(r[:, :, None], 3, axis=2))):
        outs, st = dec.decode([datas[i] for i in idx], fmt=fmt, gpu_huffman=gpu_huffman)
        _sync()
        assert all(s == 0 for s in st), st
        for i, o in zip(idx, outs):
            got = o.cpu().numpy()
            ref = conv(refs[i])
            assert np.array_equal(got, ref), (entries[i]["name"], fmt, int((got != ref).sum()))


def test_zero_run_compressed_staging_of_host_decoded_pictures(dec):
    """Host Huffman decoding (the north-star split): sequential pictures cross PCIe as sparse streams (csrc/entropy_decode.h) that the pixel
    kernels expand in LDS; progressive and multi-scan pictures stay dense; both kinds in one batch with GPU-decoded neighbours.  Every golden,
    every output layout the kernels specialise, bit-exact; the transfer shrinks to a fraction of the dense blocks."""
    import json
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        entries = json.load(f)["decode"]
    cases = [load_decode_case(e) for e in entries]
    jpegs = [c[0] for c in cases]
    for fmt in ("rgb", "rgb_planar", "yuv_planar"):
        outs, st = dec.decode(jpegs, fmt=fmt, gpu_huffman=False)
        _sync()
        stats = dec.stats()
        n_seq = sum(1 for e in entries if not e["progressive"])
        assert 0 < stats["sparse_images"] <= n_seq and stats["sparse_images"] >= n_seq - 20
        for e, (jpeg, rgb), o in zip(entries, cases, outs):
            if fmt == "yuv_planar":
                for a, b in zip(o, oracle.decode_planes(jpeg)):
                    assert np.array_equal(a.cpu().numpy(), b), e["name"]
                continue
            ref = rgb if rgb is not None else oracle.decode(jpeg)
            if fmt == "rgb_planar":
                ref = ref.transpose(2, 0, 1)
            assert np.array_equal(o.cpu().numpy(), ref), (e["name"], fmt)
    big = [oracle.encode(synth_image(1920, 1080, seed=11 + k), "420", 90) for k in range(2)] + [oracle.encode(synth_image(1001, 701, seed=3), "422", 75)]
    outs, st = dec.decode(big, gpu_huffman=False)
    _sync()
    stats = dec.stats()
    assert stats["sparse_images"] == 3 and stats["h2d_bytes"] < 0.45 * stats["coef_bytes"], stats
    for j, o in zip(big, outs):
        assert np.array_equal(o.cpu().numpy(), oracle.decode(j))
