"""The EXIF turn mapping, pinned twice without a GPU: by the definition (TIFF/EXIF tag 0x0112: "the 0th row / column of the stored picture
is the visual top / left, right, ...") on a 2 x 3 array, and by an independent implementation -- Pillow's ImageOps.exif_transpose -- on JPEG files
that carry the tag.  The reference applies orientation only inside closed-source nvJPEG (extensions/nvjpeg/type_convert.cpp:43-64 maps the
parser's struct, src/parsers/exif_orientation.h:36-57); tests/test_gpu_geometry.py checks the kernels against `upright`, this file checks
`upright` and the parser's struct against the two pins."""
import io

import numpy as np
import pytest

from helpers.geometry import upright
from nvimagecodec_amd.synth import synth_image


def test_upright_helper_matches_the_exif_definition():
    a = np.arange(6).reshape(2, 3)  # rows: [0 1 2], [3 4 5]
    assert np.array_equal(upright(a, 6), [[3, 0], [4, 1], [5, 2]])      # turn 90 degrees clockwise
    assert np.array_equal(upright(a, 8), [[2, 5], [1, 4], [0, 3]])      # turn 270 degrees clockwise
    assert np.array_equal(upright(a, 5), a.T)                            # transpose
    assert np.array_equal(upright(a, 7), [[5, 2], [4, 1], [3, 0]])      # transverse
    assert np.array_equal(upright(a, 3), [[5, 4, 3], [2, 1, 0]])


@pytest.mark.parametrize("orientation", range(1, 9))
def test_upright_is_what_pillow_does_with_a_tagged_file(orientation):
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageOps
    im = Image.fromarray(synth_image(56, 40, seed=orientation))
    exif = Image.Exif()
    exif[0x0112] = orientation
    b = io.BytesIO()
    im.save(b, "JPEG", quality=90, exif=exif.tobytes())
    stored = Image.open(io.BytesIO(b.getvalue()))
    assert stored.getexif()[0x0112] == orientation
    want = np.asarray(ImageOps.exif_transpose(stored))
    got = upright(np.asarray(stored), orientation)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("orientation", range(1, 9))
def test_parser_struct_turns_like_pillow(orientation):
    """The parser reports (rotated counter-clockwise by 0/90/180/270, then flip_x / flip_y) like the reference's
    exif_orientation.h:36-57; applying THAT to the stored pixels must give Pillow's upright picture."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageOps
    from nvimagecodec_amd import _native
    from nvimagecodec_amd import abi as A
    from test_host_framework import make_instance
    from test_parser_parity import _info
    lib = A.bind(_native.load_host())
    im = Image.fromarray(synth_image(48, 32, seed=20 + orientation))
    exif = Image.Exif()
    exif[0x0112] = orientation
    b = io.BytesIO()
    im.save(b, "JPEG", quality=90, exif=exif.tobytes())
    inst = make_instance(lib)
    st, info, _ = _info(lib, inst, b.getvalue())
    assert st == A.STATUS_SUCCESS
    o = info.orientation
    a = np.asarray(Image.open(io.BytesIO(b.getvalue())))
    turned = np.rot90(a, o.rotated // 90)  # counter-clockwise
    if o.flip_x:
        turned = turned[:, ::-1]
    if o.flip_y:
        turned = turned[::-1]
    want = np.asarray(ImageOps.exif_transpose(Image.open(io.BytesIO(b.getvalue()))))
    assert np.array_equal(turned, want)
    lib.nvimgcodecInstanceDestroy(inst)
