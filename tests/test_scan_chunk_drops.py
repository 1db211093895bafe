"""The parser's per-chunk counts of dropped bytes (ScanHeader::chunk_drops: the 00 behind an FF, both bytes of an RSTn marker, per
16,384-byte chunk of the entropy-coded segment) against a byte-by-byte count -- with stuffed FFs and restart markers placed on, before
and across chunk boundaries and the 32-byte steps of the AVX2 walk.  The GPU entropy stage's compact kernel works from these counts."""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_decode_case
from nvimagecodec_amd import _native

CHUNK = 16384


def _headers():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        m = json.load(f)
    jpeg = load_decode_case(next(e for e in m["decode"] if e["name"] == "s64x48_444_base_q90"))[0]
    sos = jpeg.index(b"\xff\xda")
    length = (jpeg[sos + 2] << 8) | jpeg[sos + 3]
    return jpeg[:sos + 2 + length]


def _reference(scan):
    drops = [0] * ((len(scan) + CHUNK - 1) // CHUNK)
    for i in range(len(scan) - 1):
        if scan[i] == 0xFF and scan[i + 1] == 0x00:
            drops[(i + 1) // CHUNK] += 1
        elif scan[i] == 0xFF and 0xD0 <= scan[i + 1] <= 0xD7:
            drops[i // CHUNK] += 1
            drops[(i + 1) // CHUNK] += 1
    return drops


def _parsed(data):
    lib = _native.load()
    out = (ctypes.c_uint32 * 64)()
    buf = (ctypes.c_uint8 * len(data)).from_buffer_copy(data)
    n = lib.hipjpegTestScanChunkDrops(ctypes.addressof(buf), len(data), 0, out, 64)
    return n, list(out)[:max(n, 0)]


@pytest.mark.parametrize("seed", range(6))
def test_counts_match_a_byte_by_byte_count(seed):
    rng = np.random.default_rng(seed)
    head = _headers()
    size = [3 * CHUNK + 77, 2 * CHUNK, CHUNK - 1, CHUNK + 1, 5 * CHUNK + 31, 40][seed]
    scan = bytearray(int(x) for x in rng.integers(0, 0xFF, size=size))  # no FF yet
    rst = 0

    def stuff(pos):
        if 0 <= pos and pos + 1 < len(scan):
            scan[pos], scan[pos + 1] = 0xFF, 0x00

    def marker(pos):
        nonlocal rst
        if 0 <= pos and pos + 1 < len(scan):
            scan[pos], scan[pos + 1] = 0xFF, 0xD0 + (rst & 7)
            rst += 1

    # around every chunk boundary and around the 32-byte steps in front of it
    for b in range(CHUNK, len(scan), CHUNK):
        for k, off in enumerate((-66, -34, -33, -32, -3, -1, 2, 30, 31, 33, 64)):
            if (k + seed) % 3 == 0:
                marker(b + off)
            else:
                stuff(b + off)
    for pos in rng.integers(0, max(len(scan) - 2, 1), size=len(scan) // 97):
        p = int(pos)
        if all(scan[q] != 0xFF for q in range(max(p - 2, 0), min(p + 3, len(scan)))):
            stuff(p)
    # markers must count RST0, RST1, ... in file order: renumber front to back
    n = 0
    for i in range(len(scan) - 1):
        if scan[i] == 0xFF and 0xD0 <= scan[i + 1] <= 0xD7:
            scan[i + 1] = 0xD0 + (n & 7)
            n += 1
    if scan[-1] == 0xFF:
        scan[-1] = 0x12
    data = head + bytes(scan) + b"\xff\xd9"
    got_n, got = _parsed(data)
    ref = _reference(scan)
    assert got_n == len(ref)
    assert got == ref


def test_a_stuffed_ff_on_the_last_byte_of_a_chunk_counts_in_the_next_one():
    """The AVX2 walk's branch-free path splits a 32-byte step's count when the FF sits on a chunk's last byte (its 00 opens the next)."""
    head = _headers()
    for extra in ((), (CHUNK - 20,), (CHUNK - 32, CHUNK - 7), (CHUNK + 5,)):
        scan = bytearray(b"\x55" * (2 * CHUNK + 100))
        for pos in (CHUNK - 1, 2 * CHUNK - 1) + tuple(extra):
            scan[pos], scan[pos + 1] = 0xFF, 0x00
        n, got = _parsed(head + bytes(scan) + b"\xff\xd9")
        assert n == 3 and got == _reference(scan), extra
        assert got[1] >= 1 and got[2] == 1
