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
    jpeg, _ = load_decode_case(entry)
    eligible = not entry["progressive"] and entry["restart"] == 0
    try:
        coefs, passes = lowlevel.entropy_decode_gpu_algorithm_host(jpeg)
    except N.HipJpegError as e:
        assert e.status == 3 and not eligible  # UNSUPPORTED: progressive / restart markers keep the host entropy stage
        return
    assert eligible and passes >= 1
    ref, _ = oracle.decode_coefficients(jpeg)
    for c, (a, b) in enumerate(zip(coefs, ref)):
        assert np.array_equal(a, b), f"component {c}"


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
