"""DHT validation as jdhuff.c jpeg_make_d_derived_tbl does it: the code space may be neither over-subscribed nor FILLED -- "no code is
allowed to be all ones".  A complete code would let a decoder read the one-bits behind a stream's end as symbols without end;
parser (through the C API's host-side scan helper) and oracle both refuse such files."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from nvimagecodec_amd import _native as N  # noqa: E402
from nvimagecodec_amd.synth import synth_image  # noqa: E402


def _with_dc_table(jpeg, bits, vals):
    """The file with its first DHT segment (DC table 0) replaced."""
    b = bytes(jpeg)
    i = b.find(b"\xff\xc4")
    L = (b[i + 2] << 8) | b[i + 3]
    assert b[i + 4] == 0x00
    seg = bytes([0x00]) + bytes(bits) + bytes(vals)
    return b[:i] + b"\xff\xc4" + (2 + len(seg)).to_bytes(2, "big") + seg + b[i + 2 + L:]


def _parses(jpeg):
    import ctypes
    import numpy as np
    a = np.frombuffer(jpeg, dtype=np.uint8)
    counts = (ctypes.c_uint32 * 64)()
    return N.load().hipjpegTestScanChunkDrops(a.ctypes.data, a.size, 0, counts, 64) >= 0


def test_complete_huffman_code_is_refused():
    jpeg = oracle.encode(synth_image(32, 24, seed=3), "444", 90)
    assert _parses(jpeg)
    # 2 codes of length 1 fill the code space: 0 and 1 -- the second is all ones
    complete = _with_dc_table(jpeg, [2] + [0] * 15, [0, 1])
    # lengths 1,2,3,3: 0, 10, 110, 111 -- complete as well
    complete2 = _with_dc_table(jpeg, [1, 1, 2] + [0] * 13, [0, 1, 2, 3])
    # lengths 1,2,3: 0, 10, 110 -- leaves 111 free: fine
    legal = _with_dc_table(jpeg, [1, 1, 1] + [0] * 13, [0, 1, 2])
    # over-subscribed: three codes of length 1
    over = _with_dc_table(jpeg, [3] + [0] * 15, [0, 1, 2])
    assert _parses(legal)
    for bad in (complete, complete2, over):
        assert not _parses(bad)
        with pytest.raises(oracle.OracleError):
            oracle.decode(bad)
