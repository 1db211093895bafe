"""GPU tests of the geometry pass: region of interest = exactly the pixels of the full decode (the reference CPU path's crop
semantics, extensions/libjpeg_turbo/jpeg_mem.cpp:206-240) and EXIF orientation = the stored picture brought upright
(ref src/parsers/exif_orientation.h:36-57 + extensions/nvjpeg/type_convert.cpp:43-64).  Expected values: the oracle's full
decode, cropped and turned with numpy.  The reference applies orientation only inside closed-source nvJPEG, so the mapping
itself is pinned by the EXIF definition and by Pillow's ImageOps.exif_transpose on files that carry the tag (tests/test_exif_turns.py), not by
a reference run."""
import numpy as np
import pytest

import oracle
from helpers.geometry import upright
from nvimagecodec_amd.synth import synth_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dec():
    import torch
    assert torch.cuda.is_available()
    from nvimagecodec_amd.lowlevel import BatchDecoder
    d = BatchDecoder(0, num_threads=4)
    yield d
    d.close()


CASES = [(1920, 1080, "420", 90), (641, 481, "422", 85), (333, 200, "444", 75), (200, 300, "gray", 90), (77, 50, "411", 80)]


@pytest.mark.parametrize("gpu_huffman", [False, True])
def test_all_orientations_and_regions_interleaved(dec, gpu_huffman):
    import torch
    jpegs, transforms, expect = [], [], []
    for k, (w, h, sub, q) in enumerate(CASES):
        j = oracle.encode(synth_image(w, h, seed=40 + k), sub, q)
        full = oracle.decode(j)
        for orientation in range(1, 9):
            roi = None if orientation % 2 else (w // 5, h // 7, w - w // 3, h - h // 9)
            jpegs.append(j)
            transforms.append((roi, orientation))
            crop = full if roi is None else full[roi[1]:roi[3], roi[0]:roi[2]]
            expect.append(upright(crop, orientation))
    outs, statuses = dec.decode(jpegs, fmt="rgb", gpu_huffman=gpu_huffman, transforms=transforms)
    torch.cuda.synchronize()
    assert all(s == 0 for s in statuses)
    for o, e, t in zip(outs, expect, transforms):
        assert tuple(o.shape) == e.shape, t
        assert np.array_equal(o.cpu().numpy(), e), t


@pytest.mark.parametrize("gpu_huffman", [False, True])
def test_regions_against_the_real_librarys_pixels(dec, gpu_huffman):
    """378 regions of interest decoded by libjpeg-turbo itself with the reference's recipe (extensions/libjpeg_turbo/jpeg_mem.cpp:206-240;
    tests/golden/make_golden_plain_upsampling.py), fancy upsampling on and off, every sampling, progressive and restart-interval files."""
    import hashlib
    import json
    import os
    import torch
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "manifest_plain.json")) as f:
        rois = json.load(f)["roi"]
    for fancy in (True, False):
        todo = [e for e in rois if e["fancy"] == fancy]
        jpegs = [open(os.path.join(GOLDEN, "decode", e["name"] + ".jpg"), "rb").read() for e in todo]
        transforms = [((e["roi"][0], e["roi"][1], e["roi"][0] + e["roi"][2], e["roi"][1] + e["roi"][3]), 1) for e in todo]
        outs, statuses = dec.decode(jpegs, fmt="rgb", fancy=fancy, gpu_huffman=gpu_huffman, transforms=transforms)
        torch.cuda.synchronize()
        assert all(s == 0 for s in statuses)
        for e, o in zip(todo, outs):
            assert hashlib.sha256(np.ascontiguousarray(o.cpu().numpy()).tobytes()).hexdigest() == e["rgb_sha256"], e


def test_planar_and_gray_formats_and_edge_regions(dec):
    import torch
    w, h = 500, 333
    j = oracle.encode(synth_image(w, h, seed=77), "420", 90)
    full = oracle.decode(j)
    regions = [(0, 0, 1, 1), (w - 1, h - 1, w, h), (0, 0, w, 8), (255, 31, 257, 33), (3, 5, 260, 40)]
    for fmt in ("bgr", "rgb_planar", "y"):
        transforms = [(r, o) for r in regions for o in (1, 6, 7)]
        outs, _ = dec.decode([j] * len(transforms), fmt=fmt, transforms=transforms)
        torch.cuda.synchronize()
        for out, (r, o) in zip(outs, transforms):
            crop = full[r[1]:r[3], r[0]:r[2]]
            if fmt == "bgr":
                e = upright(crop[:, :, ::-1], o)
            elif fmt == "rgb_planar":
                e = np.stack([upright(crop[:, :, c], o) for c in range(3)])
            else:
                e = upright(oracle.decode(j, oracle.FMT_GRAY)[r[1]:r[3], r[0]:r[2]], o)
            assert np.array_equal(out.cpu().numpy(), e), (fmt, r, o)


def test_bad_regions_and_subsampled_planes_are_rejected(dec):
    j = oracle.encode(synth_image(64, 48, seed=1), "420", 90)
    outs = dec.allocate_outputs([j] * 3)
    _, st = dec.decode([j] * 3, outs=outs, transforms=[((0, 0, 65, 48), 1), ((10, 10, 10, 20), 1), (None, 9)], check=False)
    assert st == [1, 1, 1]
    outs = dec.allocate_outputs([j], "yuv_planar")
    _, st = dec.decode([j], fmt="yuv_planar", outs=outs, transforms=[(None, 6)], check=False)
    assert st == [3]
    # and the batch after it decodes plainly again (the geometry is consumed by one batch)
    o, st = dec.decode([j])
    assert st == [0] and np.array_equal(o[0].cpu().numpy(), oracle.decode(j))
