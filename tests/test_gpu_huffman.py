"""GPU tests of the GPU entropy stage (flag HIPJPEG_FLAG_GPU_HUFFMAN / plugin option gpu_huffman): full decode with Huffman
decoding on the device must give the same pixels as the oracle, bit for bit, and broken streams must come back with the same
statuses as on the host path."""
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


@pytest.fixture(scope="module")
def dec():
    import torch
    assert torch.cuda.is_available()
    from nvimagecodec_amd.lowlevel import BatchDecoder
    d = BatchDecoder(0, num_threads=4)
    yield d
    d.close()


def _sync():
    import torch
    torch.cuda.synchronize()


def test_all_goldens_mixed_eligibility_one_batch(dec):
    """Eligible streams (baseline with one interleaved scan, with or without restart intervals; progressive without restart
    markers) go through the GPU entropy kernels, the others (progressive with restart markers) through the host entropy
    stage -- in the same batch."""
    cases = [load_decode_case(e) for e in _M["decode"]]
    outs, statuses = dec.decode([c[0] for c in cases], gpu_huffman=True)
    _sync()
    st = dec.stats()
    assert st["gpu_entropy_images"] == sum(1 for e in _M["decode"] if not (e["progressive"] and "_rst" in e["name"]))
    assert all(s == 0 for s in statuses)
    import hashlib
    for e, (jpeg, rgb), o in zip(_M["decode"], cases, outs):
        got = o.cpu().numpy()
        assert hashlib.sha256(got.tobytes()).hexdigest() == e["rgb_sha256"], e["name"]


def test_large_images_and_formats(dec):
    spec = [(1920, 1080, "420", 90, 1), (1920, 1080, "444", 50, 2), (3840, 2160, "422", 92, 3), (1283, 721, "411", 30, 4), (640, 480, "gray", 75, 5),
            (2560, 1440, "440", 85, 6)]
    jpegs = [oracle.encode(synth_image(w, h, seed=s), sub, q) for (w, h, sub, q, s) in spec]
    for fmt in ("rgb", "bgr_planar", "yuv_planar"):
        outs, _ = dec.decode(jpegs, fmt=fmt, gpu_huffman=True)
        _sync()
        assert dec.stats()["gpu_entropy_images"] == len(jpegs)
        for j, o in zip(jpegs, outs):
            if fmt == "yuv_planar":
                for a, b in zip(o, oracle.decode_planes(j)):
                    assert np.array_equal(a.cpu().numpy(), b)
            elif fmt == "rgb":
                assert np.array_equal(o.cpu().numpy(), oracle.decode(j))
            else:
                assert np.array_equal(o.cpu().numpy(), oracle.decode(j, oracle.FMT_BGR).transpose(2, 0, 1))


def test_bad_streams_get_the_host_path_statuses(dec):
    good = oracle.encode(synth_image(320, 240, seed=9), "420", 90)
    trunc = good[: len(good) * 2 // 3] + b"\xff\xd9"
    flipped = bytearray(good)
    for k in range(len(good) // 2, len(good) // 2 + 40):
        flipped[k] ^= 0x5A
    flipped = bytes(flipped).replace(b"\xff", b"\xfe")[: len(good)]
    flipped = good[:700] + flipped[700:-2] + b"\xff\xd9"
    jpegs = [good, trunc, good, flipped]
    outs = dec.allocate_outputs(jpegs)
    _, st_gpu = dec.decode(jpegs, outs=outs, gpu_huffman=True, check=False)
    _sync()
    first = outs[0].cpu().numpy().copy()
    _, st_cpu = dec.decode(jpegs, outs=outs, gpu_huffman=False, check=False)
    _sync()
    assert st_gpu[0] == 0 and st_gpu[2] == 0 and st_gpu[1] in (4, 5)
    assert [s == 0 for s in st_gpu] == [s == 0 for s in st_cpu]
    assert np.array_equal(first, oracle.decode(good))
    if st_cpu[3] == 0:  # corrupted bits can still be decodable: then both paths must agree on the pixels as well
        a = outs[3].cpu().numpy().copy()
        dec.decode(jpegs, outs=outs, gpu_huffman=True, check=False)
        _sync()
        assert np.array_equal(outs[3].cpu().numpy(), a)


def test_config1_batch_1080p_gpu_huffman_repeated(dec):
    jpegs = [oracle.encode(synth_image(1920, 1080, seed=s), "420", 90) for s in range(3)] * 6
    refs = {}
    for rep in range(3):
        outs, _ = dec.decode(jpegs, gpu_huffman=True)
        _sync()
        for j, o in zip(jpegs, outs):
            if j not in refs:
                refs[j] = oracle.decode(j)
            assert np.array_equal(o.cpu().numpy(), refs[j])


def test_randomly_damaged_streams_never_disagree_with_the_host_path(dec):
    """Deterministic fuzzing: bit flips, byte overwrites, truncations and spliced scans in one batch.  Whatever the damage,
    the GPU entropy stage must end in the same verdict (decodable or not) as the host entropy stage, and where both decode
    the pixels must be identical -- and no kernel may fault on the way."""
    import random
    import torch
    rng = random.Random(20240607)
    bases = [oracle.encode(synth_image(w, h, seed=s), sub, q) for (w, h, sub, q, s) in
             ((640, 360, "420", 90, 1), (321, 243, "422", 75, 2), (200, 200, "444", 95, 3), (512, 64, "gray", 60, 4))]
    jpegs = []
    for _ in range(48):
        b = bytearray(rng.choice(bases))
        sos = bytes(b).rfind(b"\xff\xda") + 14
        kind = rng.randrange(5)
        if kind == 0:    # flip a few bits inside the scan
            for _ in range(rng.randrange(1, 6)):
                i = rng.randrange(sos, len(b) - 2)
                b[i] ^= 1 << rng.randrange(8)
        elif kind == 1:  # overwrite a run of bytes
            i = rng.randrange(sos, len(b) - 40)
            for k in range(rng.randrange(1, 32)):
                b[i + k] = rng.randrange(256)
        elif kind == 2:  # truncate, keep an EOI
            b = b[: rng.randrange(sos + 1, len(b) - 2)] + b"\xff\xd9"
        elif kind == 3:  # drop a chunk from the middle of the scan
            i = rng.randrange(sos, len(b) - 200)
            del b[i : i + rng.randrange(1, 150)]
        else:            # append garbage scan data before the EOI
            b = b[:-2] + bytes(rng.randrange(256) for _ in range(rng.randrange(1, 300))) + b"\xff\xd9"
        jpegs.append(bytes(b))
    outs_g = dec.allocate_outputs(jpegs)
    _, st_gpu = dec.decode(jpegs, outs=outs_g, gpu_huffman=True, check=False)
    torch.cuda.synchronize()
    got = [o.cpu().numpy().copy() if o is not None else None for o in outs_g]
    _, st_cpu = dec.decode(jpegs, outs=outs_g, gpu_huffman=False, check=False)
    torch.cuda.synchronize()
    assert [s == 0 for s in st_gpu] == [s == 0 for s in st_cpu]
    for i, (s, o) in enumerate(zip(st_cpu, outs_g)):
        if s == 0:
            assert np.array_equal(got[i], o.cpu().numpy()), i
    # the decoder is still healthy afterwards
    outs, _ = dec.decode(bases[:2], gpu_huffman=True)
    torch.cuda.synchronize()
    assert np.array_equal(outs[0].cpu().numpy(), oracle.decode(bases[0]))


def test_damaged_restart_intervals_are_handed_to_the_host_decoder(dec):
    """Found by tests/campaigns/fuzz_damage.py: inside a damaged restart interval the walk of the GPU stage can cross the next boundary
    in the middle of an MCU and look healthy again one interval later (the block count happens to add up).  The invariant it
    now checks: the first block of every interval starts exactly at its boundary, and no walk reads past a boundary.  The
    three streams below used to decode "successfully" on the GPU while the host entropy decoder rejects them."""
    import torch
    b7 = oracle.encode(synth_image(400, 300, seed=5), "420", 85, restart_interval=7)
    b1 = oracle.encode(synth_image(333, 222, seed=7), "420", 30, restart_interval=1)
    patched = bytearray(b7)
    patched[10574:10574 + 19] = bytes.fromhex("15995f381651ef13b8d8206e808f6cc4a8f1cb")
    flipped = bytearray(b1)
    flipped[1992], flipped[2394] = 196, 112
    cut = b1[:3601] + b1[3607:]
    jpegs = [bytes(patched), bytes(flipped), bytes(cut), b7, b1]
    outs = dec.allocate_outputs(jpegs)
    _, st_gpu = dec.decode(jpegs, outs=outs, gpu_huffman=True, check=False)
    torch.cuda.synchronize()
    good = [o.cpu().numpy().copy() for o in outs[3:]]
    _, st_cpu = dec.decode(jpegs, outs=outs, gpu_huffman=False, check=False)
    torch.cuda.synchronize()
    assert list(st_gpu) == list(st_cpu) and all(s != 0 for s in st_gpu[:3]) and st_gpu[3] == 0 and st_gpu[4] == 0
    assert np.array_equal(good[0], oracle.decode(b7)) and np.array_equal(good[1], oracle.decode(b1))


def test_pipelined_submit_wait_with_three_batches_in_flight(dec):
    """hipjpegDecodeBatchSubmit / Wait: results come back in submission order, identical to the one-shot path, also when a
    batch carries a stream the kernels have to hand back to the host decoder and when batches mix both entropy stages."""
    import torch
    good = [oracle.encode(synth_image(320 + 16 * k, 200 + 8 * k, seed=70 + k), "420" if k % 2 else "422", 85) for k in range(6)]
    prog = next(load_decode_case(e)[0] for e in _M["decode"] if e["progressive"] and e["width"] >= 64)
    broken = good[0][: len(good[0]) // 2] + b"\xff\xd9"
    batches = [good[:4], [good[4], prog, good[5]], [good[1], broken, good[2]], good[2:6], [prog], good[:3]]
    refs = {j: oracle.decode(j) for b in batches for j in b if j != broken}
    outs = [dec.allocate_outputs(b) for b in batches]
    results = []
    for k, b in enumerate(batches):
        dec.submit(b, outs[k])
        if k >= 2:
            results.append(dec.wait(check=False))
    results.append(dec.wait(check=False))
    results.append(dec.wait(check=False))
    torch.cuda.synchronize()
    assert len(results) == len(batches)
    for k, (b, st) in enumerate(zip(batches, results)):
        for i, j in enumerate(b):
            if j == broken:
                assert st[i] in (4, 5), (k, i)
            else:
                assert st[i] == 0, (k, i)
                assert np.array_equal(outs[k][i].cpu().numpy(), refs[j]), (k, i)


def test_restart_interval_streams_on_the_gpu(dec):
    import torch
    spec = [(1920, 1080, "420", 90, 120), (1920, 1080, "420", 90, 1), (1283, 721, "422", 80, 7), (640, 480, "444", 95, 40), (500, 333, "gray", 70, 3),
            (3840, 2160, "420", 85, 240)]
    jpegs = [oracle.encode(synth_image(w, h, seed=11 * k + 1), sub, q, restart_interval=r) for k, (w, h, sub, q, r) in enumerate(spec)]
    outs, _ = dec.decode(jpegs, gpu_huffman=True)
    torch.cuda.synchronize()
    assert dec.stats()["gpu_entropy_images"] == len(jpegs)
    for j, o in zip(jpegs, outs):
        assert np.array_equal(o.cpu().numpy(), oracle.decode(j))


def test_periodic_streams_are_handed_to_the_host_decoder_not_walked_group_by_group(dec):
    """Stripes / test patterns: every MCU codes the same bits, the stream is periodic, and a decoder that started in the wrong state can
    stay on a stable wrong trajectory -- corrections then travel through the image one subsequence after the other (one launch per
    group of 255: 33 launches and 157 ms for a 4096 x 2048 picture when this was found).  The tail / ripple kernels have a round
    budget (192: noise at q98 needs 125) and resolve() a launch budget; images that exhaust them go to the host entropy decoder.  Pixels are exact either way; a
    photograph in the same batch stays on the GPU path."""
    import time
    import torch
    from nvimagecodec_amd.synth import synth_image
    stripes = np.full((2048, 4096, 3), 137, np.uint8)
    stripes[:, ::16] = 30
    flat = np.full((1024, 1024, 3), 90, np.uint8)
    photo = synth_image(640, 480, seed=12)
    jpegs = [oracle.encode(stripes, "420", 90), oracle.encode(flat, "444", 90), oracle.encode(photo, "420", 90),
             oracle.encode(stripes[:1024, :1024], "444", 75)]
    refs = [oracle.decode(j) for j in jpegs]
    outs, st = dec.decode(jpegs, gpu_huffman=True, check=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs, st = dec.decode(jpegs, gpu_huffman=True, check=False)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert list(st) == [0, 0, 0, 0]
    for o, r in zip(outs, refs):
        assert np.array_equal(o.cpu().numpy(), r)
    s = dec.stats()
    assert s["gpu_entropy_images"] == 4 and s["sync_launches"] <= 6
    assert 1 <= dec.host_fallbacks() <= 2      # the striped pictures; never the flat one or the photograph
    assert elapsed < 0.1, elapsed              # generous: ~25 ms with the host decoder taking the stripes


def test_hybrid_huffman_threshold_splits_a_batch_by_area(dec):
    """hipjpegSetHybridHuffmanThreshold / plugin option hybrid_huffman_threshold (reference extensions/nvjpeg/cuda_decoder.cpp:188-209,
    512-521): with the GPU entropy stage on, only images of MORE than the threshold's pixels take it, the rest the host decoder -- same
    pixels either way."""
    small = oracle.encode(synth_image(64, 48, seed=5), "420", 90)
    large = oracle.encode(synth_image(320, 240, seed=6), "420", 90)
    batch = [small, large, small, large, large]
    try:
        dec.set_hybrid_huffman_threshold(64 * 48)  # "more than": the 64 x 48 pictures stay on the host
        outs, st = dec.decode(batch, gpu_huffman=True)
        _sync()
        assert dec.stats()["gpu_entropy_images"] == 3
        for j, o in zip(batch, outs):
            assert np.array_equal(o.cpu().numpy(), oracle.decode(j))
        dec.set_hybrid_huffman_threshold(10 ** 9)
        outs, st = dec.decode(batch, gpu_huffman=True)
        _sync()
        assert dec.stats()["gpu_entropy_images"] == 0
    finally:
        dec.set_hybrid_huffman_threshold(0)
    outs, st = dec.decode(batch, gpu_huffman=True)
    _sync()
    assert dec.stats()["gpu_entropy_images"] == 5


def test_zero_copy_input_from_pinned_memory(dec):
    """Bitstreams in page-locked memory (pinned torch tensors) are DMAed from where they lie (hipjpegDecodeBatchZeroCopyImages says how
    many); a batch may mix them with pageable inputs, with host-decoded and with progressive images -- same pixels as always.  Also through
    the pipelined Submit/Wait entry points, three batches in flight."""
    import torch
    srcs = [oracle.encode(synth_image(320 + 16 * i, 200 + 8 * i, seed=70 + i), "420" if i % 2 else "444", 90) for i in range(6)]
    entries = [e for e in _M["decode"] if e["progressive"]][:2]
    prog = [load_decode_case(e)[0] for e in entries]
    pinned = [torch.frombuffer(bytearray(j), dtype=torch.uint8).pin_memory() for j in srcs]
    batch = [pinned[0], srcs[1], pinned[2], prog[0], pinned[3], srcs[4], pinned[5], prog[1]]
    raw = [srcs[0], srcs[1], srcs[2], prog[0], srcs[3], srcs[4], srcs[5], prog[1]]
    refs = [oracle.decode(j) for j in raw]
    outs, st = dec.decode(batch, gpu_huffman=True)
    _sync()
    assert dec.stats()["zero_copy_images"] == 4
    for o, r in zip(outs, refs):
        assert np.array_equal(o.cpu().numpy(), r)
    outs, st = dec.decode(batch, gpu_huffman=False)   # host Huffman: nothing to DMA from the caller's memory
    _sync()
    assert dec.stats()["zero_copy_images"] == 0
    for o, r in zip(outs, refs):
        assert np.array_equal(o.cpu().numpy(), r)
    rings = [dec.allocate_outputs(raw, "rgb") for _ in range(3)]
    for k in range(5):
        dec.submit(batch, rings[k % 3], gpu_huffman=True)
        if k > 1:
            dec.wait()
    dec.wait()
    dec.wait()
    _sync()
    for ring in rings:
        for o, r in zip(ring, refs):
            assert np.array_equal(o.cpu().numpy(), r)
