"""The GPU entropy decoder's algorithm (self-synchronizing subsequence decoding, csrc/huffman_gpu_core.h) verified WITHOUT a
GPU: hipjpegEntropyDecodeGpuAlgorithmHost runs the kernels' own decode routine lane by lane on the host.  Coefficients must
equal the oracle's (and therefore the host entropy decoder's) exactly."""
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, load_decode_case
from nvimagecodec_amd import _native as N
from nvimagecodec_amd import lowlevel
from nvimagecodec_amd.synth import synth_image

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    _M = json.load(_f)


@pytest.mark.parametrize("entry", _M["decode"], ids=lambda e: e["name"])
def test_algorithm_matches_oracle_or_declines(entry):
    """Baseline streams: self-synchronising subsequence decode.  Progressive streams: the walk + replay algorithm of
    progressive_gpu_core.h (sequential walk per scan for the block positions, then every block on its own) -- except with
    restart markers, which keep the host entropy stage."""
    jpeg, _ = load_decode_case(entry)
    eligible = not (entry["progressive"] and "_rst" in entry["name"])
    try:
        coefs, passes = lowlevel.entropy_decode_gpu_algorithm_host(jpeg)
    except N.HipJpegError as e:
        assert e.status == 3 and not eligible  # UNSUPPORTED
        return
    assert eligible and (passes >= 1 or entry["progressive"])
    ref, _ = oracle.decode_coefficients(jpeg)
    for c, (a, b) in enumerate(zip(coefs, ref)):
        assert np.array_equal(a, b), f"component {c}"


def test_progressive_walk_and_replay_on_larger_images():
    """Many end-of-band runs, long refinement scans, every sampling: coefficients equal the oracle's."""
    import io
    try:
        from PIL import Image
    except ImportError:
        pytest.skip("Pillow (libjpeg-turbo) makes the progressive inputs")
    for (w, h, sub, q) in ((640, 360, 0, 90), (333, 517, 2, 75), (800, 600, 1, 50), (1280, 720, 2, 95), (257, 129, 0, 20)):
        b = io.BytesIO()
        Image.fromarray(synth_image(w, h, seed=w + q)).save(b, "JPEG", quality=q, subsampling=sub, progressive=True)
        jpeg = b.getvalue()
        coefs, _ = lowlevel.entropy_decode_gpu_algorithm_host(jpeg)
        ref, _ = oracle.decode_coefficients(jpeg)
        assert all(np.array_equal(a, b) for a, b in zip(coefs, ref)), (w, h, sub, q)
    b = io.BytesIO()
    Image.fromarray(synth_image(200, 120, seed=5)[:, :, 0]).save(b, "JPEG", quality=85, progressive=True)
    coefs, _ = lowlevel.entropy_decode_gpu_algorithm_host(b.getvalue())
    ref, _ = oracle.decode_coefficients(b.getvalue())
    assert all(np.array_equal(x, y) for x, y in zip(coefs, ref))


def test_damaged_progressive_streams_get_the_host_verdict():
    """Bit flips / cuts inside progressive scans: the walk + replay must reject exactly the streams the host decoder rejects
    and agree on the coefficients of the ones both accept."""
    import random
    cases = [load_decode_case(e)[0] for e in _M["decode"] if e["progressive"] and "_rst" not in e["name"] and e["width"] <= 64]
    rng = random.Random(777)
    checked = accepted = 0
    for _ in range(3000):
        j = bytearray(rng.choice(cases))
        first_sos = j.find(b"\xff\xda")
        k = rng.randrange(first_sos + 14, len(j) - 2)
        if rng.randrange(2):
            j[k] ^= 1 << rng.randrange(8)
        else:
            j[k] = rng.randrange(256)
        j = bytes(j)

        def run(fn):
            try:
                return 0, fn(j)[0]
            except N.HipJpegError as e:
                return e.status, None
        sg, cg = run(lowlevel.entropy_decode_gpu_algorithm_host)
        if sg in (2, 3):
            continue  # the damage hit a marker segment or made the stream ineligible: host entropy stage only
        sh, ch = run(lowlevel.entropy_decode_host)
        checked += 1
        if sh == 0 and sg == 0:
            accepted += 1
            assert all(np.array_equal(a[: b.shape[0]], b) for a, b in zip(cg, ch)), j.hex()
        else:
            # a stream the kernels reject goes to the host decoder, which names the error -- what must never happen is the
            # kernels accepting what the host decoder rejects
            assert not (sg == 0 and sh != 0), (sh, sg, j.hex())
    assert checked > 1000 and accepted > 100


def test_many_subsequences_need_several_sync_passes():
    # a detailed image at high quality: thousands of subsequences, long blocks -> the correction wave needs several passes
    for (w, h, sub, q) in ((1280, 720, "420", 95), (800, 600, "444", 90), (1023, 511, "422", 60)):
        jpeg = oracle.encode(synth_image(w, h, seed=w), sub, q)
        coefs, passes = lowlevel.entropy_decode_gpu_algorithm_host(jpeg)
        ref, _ = oracle.decode_coefficients(jpeg)
        assert all(np.array_equal(a, b) for a, b in zip(coefs, ref))
        assert 1 <= passes < 64


def test_corrupt_and_truncated_streams_are_reported():
    jpeg = oracle.encode(synth_image(320, 240, seed=3), "420", 90)
    with pytest.raises(N.HipJpegError) as ei:
        lowlevel.entropy_decode_gpu_algorithm_host(jpeg[: len(jpeg) * 2 // 3] + b"\xff\xd9")
    assert ei.value.status in (4, 5)


def _with_fill_byte(jpeg):
    """FF 00 inside the scan -> FF FF 00: a fill byte in front of a stuffed FF.  libjpeg-turbo skips fill bytes, so the
    stream still decodes to the same coefficients -- but the GPU stage's plain 'drop the 00 after an FF' rule does not
    cover it, so such streams must stay on the host entropy stage."""
    info = lowlevel.get_image_info(jpeg)
    sos = jpeg.rfind(b"\xff\xda")
    i = jpeg.find(b"\xff\x00", sos + 14)
    assert info["num_scans"] == 1 and i > 0
    return jpeg[:i] + b"\xff" + jpeg[i:]


def test_streams_with_fill_bytes_stay_on_the_host_stage():
    jpeg = oracle.encode(synth_image(200, 120, seed=5), "420", 95)
    odd = _with_fill_byte(jpeg)
    with pytest.raises(N.HipJpegError) as ei:
        lowlevel.entropy_decode_gpu_algorithm_host(odd)
    assert ei.value.status == 3
    a, _ = oracle.decode_coefficients(jpeg)
    b = lowlevel.entropy_decode_host(odd)[0]
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("interval", [1, 2, 5, 17, 60, 1000])
def test_restart_intervals(interval):
    """DRI streams: markers are dropped with the byte stuffing, decoders re-synchronise exactly at every interval boundary
    and the DC predictors start over."""
    for (w, h, sub, q) in ((320, 240, "420", 90), (257, 129, "444", 75), (200, 120, "gray", 95), (640, 96, "422", 50)):
        jpeg = oracle.encode(synth_image(w, h, seed=w + interval), sub, q, restart_interval=interval)
        coefs, passes = lowlevel.entropy_decode_gpu_algorithm_host(jpeg)
        ref, _ = oracle.decode_coefficients(jpeg)
        assert all(np.array_equal(a, b) for a, b in zip(coefs, ref)), (w, h, sub, interval)


def test_damaged_restart_intervals_are_reported():
    """CPU twin of tests/test_gpu_huffman.py::test_damaged_restart_intervals_are_handed_to_the_host_decoder: the kernels' own
    walk routines must notice a trajectory that crossed a restart boundary inside an MCU."""
    b7 = oracle.encode(synth_image(400, 300, seed=5), "420", 85, restart_interval=7)
    b1 = oracle.encode(synth_image(333, 222, seed=7), "420", 30, restart_interval=1)
    patched = bytearray(b7)
    patched[10574:10574 + 19] = bytes.fromhex("15995f381651ef13b8d8206e808f6cc4a8f1cb")
    flipped = bytearray(b1)
    flipped[1992], flipped[2394] = 196, 112
    for bad in (bytes(patched), bytes(flipped), b1[:3601] + b1[3607:]):
        with pytest.raises(N.HipJpegError):
            lowlevel.entropy_decode_gpu_algorithm_host(bad)
    for good in (b7, b1):
        coefs, _ = lowlevel.entropy_decode_gpu_algorithm_host(good)
        ref, _ = oracle.decode_coefficients(good)
        assert all(np.array_equal(a, b) for a, b in zip(coefs, ref))


def test_damaged_streams_get_the_host_verdict():
    """Host entropy decoder and GPU algorithm must agree on WHETHER a damaged stream decodes (a final symbol that reaches
    into the slack behind the data used to be accepted by the GPU algorithm only), and on the coefficients when it does."""
    import random
    cases = [load_decode_case(e)[0] for e in _M["decode"] if not e["progressive"] and e["width"] <= 64]
    rng = random.Random(20261004)
    checked = 0
    for _ in range(4000):
        j = bytearray(rng.choice(cases))
        sos = j.rfind(b"\xff\xda")
        lo = sos + 4 + j[sos + 3]
        if lo >= len(j) - 3:
            continue
        k = rng.randrange(lo, len(j) - 2)
        mode = rng.randrange(3)
        if mode == 0:
            j[k] ^= 1 << rng.randrange(8)
        elif mode == 1:
            j = j[:k] + j[-2:]
        else:
            j[k] = rng.randrange(256)
        j = bytes(j)

        def run(fn):
            try:
                return 0, fn(j)[0]
            except N.HipJpegError as e:
                return e.status, None
        sg, cg = run(lowlevel.entropy_decode_gpu_algorithm_host)
        if sg == 3:
            continue  # the damage made the stream ineligible (marker inside the scan): host stage only
        sh, ch = run(lowlevel.entropy_decode_host)
        assert (sh == 0) == (sg == 0), (sh, sg, j.hex())
        if sh == 0:
            assert all(np.array_equal(a[: b.shape[0]], b) for a, b in zip(cg, ch))
        checked += 1
    assert checked > 2000
