"""Progressive (SOF2) scans on the GPU entropy stage (csrc/progressive_gpu_core.h): a sequential walk per scan finds where every
block's data starts, then one lane per block replays all scans of its block.  Bit-exact against the libjpeg-turbo goldens and
the oracle; streams the kernels cannot vouch for go to the host decoder, whose verdict the caller sees.
Reference: the nvJPEG plugin accepts SOF2 (extensions/nvjpeg/cuda_decoder.cpp:75-81)."""
import io
import json
import os
import random

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, load_decode_case
from nvimagecodec_amd.synth import synth_image

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    _M = json.load(_f)
_PROG = [e for e in _M["decode"] if e["progressive"]]


@pytest.fixture(scope="module")
def dec():
    import torch
    assert torch.cuda.is_available()
    from nvimagecodec_amd.lowlevel import BatchDecoder
    d = BatchDecoder(device=0, num_threads=4)
    yield d
    d.close()


def _sync():
    import torch
    torch.cuda.synchronize()


def test_every_progressive_golden_in_one_batch_with_baseline_neighbours(dec):
    """All 45 progressive goldens (4:4:4 / 4:2:2 / 4:2:0 / gray, odd sizes, with and without restart markers) mixed with
    baseline streams in ONE batch: the eligible ones take the GPU walk + replay, restart-marker ones the host entropy stage."""
    entries = _PROG + [e for e in _M["decode"] if not e["progressive"]][:20]
    cases = [load_decode_case(e) for e in entries]
    outs, st = dec.decode([c[0] for c in cases], fmt="rgb", gpu_huffman=True)
    _sync()
    assert all(s == 0 for s in st)
    n_gpu_prog = sum(1 for e in _PROG if "_rst" not in e["name"])
    assert dec.stats()["gpu_entropy_images"] >= n_gpu_prog
    for e, (jpeg, rgb), o in zip(entries, cases, outs):
        ref = rgb if rgb is not None else oracle.decode(jpeg)
        assert np.array_equal(o.cpu().numpy(), ref), e["name"]


@pytest.mark.parametrize("fmt", ["bgr", "rgb_planar", "y", "yuv_planar"])
def test_progressive_output_formats(dec, fmt):
    names = ["s50x37_420_prog_q90", "s33x65_422_prog_q50", "s64x48_444_prog_q90", "s17x13_gray_prog_q90", "c5_640x360_444_prog_q90"]
    jpegs = [load_decode_case(next(e for e in _PROG if e["name"] == n))[0] for n in names]
    outs, _ = dec.decode(jpegs, fmt=fmt, gpu_huffman=True)
    _sync()
    assert dec.stats()["gpu_entropy_images"] == len(jpegs)
    for n, j, o in zip(names, jpegs, outs):
        if fmt == "yuv_planar":
            for a, b in zip(o, oracle.decode_planes(j)):
                assert np.array_equal(a.cpu().numpy(), b), n
            continue
        ref = oracle.decode(j, {"bgr": oracle.FMT_BGR, "rgb_planar": oracle.FMT_RGB, "y": oracle.FMT_GRAY}[fmt])
        if fmt == "rgb_planar":
            ref = ref.transpose(2, 0, 1)
        assert np.array_equal(o.cpu().numpy(), ref), (n, fmt)


def _pil_progressive(im, quality, sub):
    try:
        from PIL import Image
    except ImportError:
        pytest.skip("Pillow (libjpeg-turbo) makes the larger progressive inputs")
    b = io.BytesIO()
    if im.ndim == 2:
        Image.fromarray(im).save(b, "JPEG", quality=quality, progressive=True)
    else:
        Image.fromarray(im).save(b, "JPEG", quality=quality, subsampling={"444": 0, "422": 1, "420": 2}[sub], progressive=True)
    return b.getvalue()


def test_progressive_shapes_qualities_and_long_end_of_band_runs(dec):
    """Flat regions give end-of-band runs over thousands of blocks (EOB14 + 14 extra bits), q100 gives dense refinement scans,
    widths around the 64-block hand-over groups of the walk pipeline."""
    imgs = []
    flat = np.full((600, 800, 3), 128, dtype=np.uint8)
    flat[200:260, 300:420] = synth_image(120, 60, seed=3)
    imgs.append((flat, 90, "420"))
    imgs.append((flat, 90, "444"))
    for (w, h, q, sub) in ((511, 64, 100, "444"), (512, 8, 100, "420"), (513, 9, 30, "422"), (1024, 1032, 75, "420"), (8, 8, 90, "444"), (72, 520, 5, "420")):
        imgs.append((synth_image(w, h, seed=w + h), q, sub))
    imgs.append((synth_image(640, 480, seed=9)[:, :, 1].copy(), 85, "gray"))
    jpegs = [_pil_progressive(im, q, sub) for im, q, sub in imgs]
    outs, st = dec.decode(jpegs, fmt="rgb", gpu_huffman=True)
    _sync()
    assert all(s == 0 for s in st) and dec.stats()["gpu_entropy_images"] == len(jpegs)
    for j, o in zip(jpegs, outs):
        assert np.array_equal(o.cpu().numpy(), oracle.decode(j))


def test_damaged_progressive_streams_get_the_host_path_statuses(dec):
    """Bit flips inside the scans: GPU path and host path must report the same status per image and, where both decode,
    the same pixels (the kernels hand anything they cannot vouch for to the host entropy decoder)."""
    base = [load_decode_case(e)[0] for e in _PROG if "_rst" not in e["name"] and e["width"] >= 33]
    rng = random.Random(4242)
    damaged = []
    for _ in range(96):
        j = bytearray(rng.choice(base))
        first_sos = j.find(b"\xff\xda")
        for _ in range(rng.randrange(1, 3)):
            k = rng.randrange(first_sos + 14, len(j) - 2)
            j[k] ^= 1 << rng.randrange(8)
        damaged.append(bytes(j))
    outs_h = dec.allocate_outputs(damaged, "rgb")
    outs_g = dec.allocate_outputs(damaged, "rgb")
    keep = [i for i, o in enumerate(outs_h) if o is not None]
    damaged = [damaged[i] for i in keep]
    outs_h = [outs_h[i] for i in keep]
    outs_g = [outs_g[i] for i in keep]
    _, st_h = dec.decode(damaged, outs=outs_h, check=False, gpu_huffman=False)
    _sync()
    _, st_g = dec.decode(damaged, outs=outs_g, check=False, gpu_huffman=True)
    _sync()
    assert st_h == st_g
    assert any(s != 0 for s in st_h) and any(s == 0 for s in st_h)
    for s, a, b in zip(st_h, outs_h, outs_g):
        if s == 0:
            assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())


def test_progressive_through_the_pipelined_entry_points(dec):
    import torch
    jpegs = [load_decode_case(e)[0] for e in _PROG if "_rst" not in e["name"]][:24]
    refs = [oracle.decode(j) for j in jpegs]
    ring = [dec.allocate_outputs(jpegs, "rgb") for _ in range(3)]
    for k in range(5):
        dec.submit(jpegs, ring[k % 3], gpu_huffman=True)
        if k >= 2:
            assert all(s == 0 for s in dec.wait())
    dec.wait()
    dec.wait()
    torch.cuda.synchronize()
    for outs in ring:
        for r, o in zip(refs, outs):
            assert np.array_equal(o.cpu().numpy(), r)


def test_deeper_pipeline_for_progressive_batches(dec):
    """hipjpegSetPipelineDepth: six batches of progressive and baseline images in flight (each page its own entropy stream),
    different content per batch, every output checked; the depth cannot change while batches are in flight and is bounded."""
    import torch
    from nvimagecodec_amd import _native as N
    prog = [load_decode_case(e)[0] for e in _PROG if "_rst" not in e["name"]][:12]
    base = [load_decode_case(e)[0] for e in _M["decode"] if not e["progressive"] and e["sub"] != "gray" and e["pixels"]][:12]
    depth = 6
    dec.set_pipeline_depth(depth)
    batches = [(prog + base)[k:] + (prog + base)[:k] for k in range(depth)]
    refs = [[oracle.decode(j) for j in b] for b in batches]
    ring = [dec.allocate_outputs(b, "rgb") for b in batches]
    for rep in range(2):
        for k in range(depth):
            dec.submit(batches[k], ring[k], gpu_huffman=True)
        with pytest.raises(N.HipJpegError):
            dec.set_pipeline_depth(3)  # batches in flight
        with pytest.raises(N.HipJpegError):
            dec.submit(batches[0], ring[0], gpu_huffman=True)  # every page is taken
        for k in range(depth):
            assert all(s == 0 for s in dec.wait())
        torch.cuda.synchronize()
        for rb, outs in zip(refs, ring):
            for r, o in zip(rb, outs):
                assert np.array_equal(o.cpu().numpy(), r)
                o.zero_()
    with pytest.raises(N.HipJpegError):
        dec.set_pipeline_depth(9)
    dec.set_pipeline_depth(3)


def _sos_offsets(jpeg):
    """Byte offsets just behind every SOS header of a file (where the scan's entropy-coded bytes begin)."""
    offs, i, b = [], 2, bytes(jpeg)
    while i + 4 <= len(b):
        assert b[i] == 0xFF
        m = b[i + 1]
        if m == 0xD9:
            break
        L = (b[i + 2] << 8) | b[i + 3]
        i += 2 + L
        if m == 0xDA:
            offs.append(i)
            while i + 1 < len(b) and not (b[i] == 0xFF and b[i + 1] != 0 and not 0xD0 <= b[i + 1] <= 0xD7):
                i += 1
    return offs


def test_progressive_scan_without_entropy_coded_bytes(dec):
    """a progressive file cut right behind an SOS header (its 2nd and its last), or with EOI directly behind one, has a scan
    of zero bytes.  Such files must not reach the walk kernel (whose reader would look at word -1): the host decoder names the error,
    the healthy neighbours in the batch decode bit-exactly, and the statuses are those of the host entropy stage."""
    e = next(e for e in _PROG if e["name"] == "s64x48_444_prog_q90")
    good, rgb = load_decode_case(e)
    offs = _sos_offsets(good)
    assert len(offs) >= 3
    bad = [good[:offs[1]], good[:offs[-1]], good[:offs[1]] + b"\xff\xd9", good[:offs[-1]] + b"\xff\xd9"]
    batch = [good, bad[0], good, bad[1], bad[2], good, bad[3]]
    outs_h, st_h = dec.decode(batch, fmt="rgb", gpu_huffman=False, check=False)
    _sync()
    outs_g, st_g = dec.decode(batch, fmt="rgb", gpu_huffman=True, check=False)
    _sync()
    assert list(st_g) == list(st_h)
    assert all(st_g[i] != 0 for i in (1, 3, 4, 6)) and all(st_g[i] == 0 for i in (0, 2, 5))
    for i in (0, 2, 5):
        assert np.array_equal(outs_g[i].cpu().numpy(), rgb if rgb is not None else oracle.decode(good))
