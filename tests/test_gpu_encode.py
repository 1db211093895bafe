"""GPU parity tests for the encode hot path (BASELINE.json configs[2]): HIP colour-convert + downsample + FDCT + quantize,
through the C-ABI (hipjpegEncodeBatch).  Integer work => exact equality of quantized coefficients with the oracle and of
the bitstream with libjpeg-turbo's (golden vectors).  The reference's own encode test compares against a direct nvJPEG
encode byte for byte (test/extensions/nvjpeg_ext_encoder_test.cpp:109-151)."""
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, load_encode_case
from nvimagecodec_amd.synth import synth_image

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    _M = json.load(_f)


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module", params=["host_huffman", "gpu_huffman"])
def enc(torch_mod, request):
    """Every test runs twice: entropy coding on the host thread pool, and on the GPU (gpu_huffman_encode.hip) -- the files
    must be byte-identical either way."""
    from nvimagecodec_amd.lowlevel import BatchEncoder
    e = BatchEncoder(0, num_threads=4, gpu_huffman=request.param == "gpu_huffman")
    yield e
    e.close()


def _real(coefs_oracle, info_like):
    return coefs_oracle


def _check_against_oracle(enc, torch, images, subs, quals, input_format="rgb", **kw):
    dev = [torch.from_numpy(np.ascontiguousarray(im)).cuda() for im in images]
    streams = enc.encode(dev, subsampling=list(subs), quality=list(quals), input_format=input_format, **kw)
    for i, (im, sub, q) in enumerate(zip(images, subs, quals)):
        rgb = im
        if input_format == "bgr":
            rgb = im[:, :, ::-1]
        elif input_format.endswith("planar"):
            rgb = im.transpose(1, 2, 0)
            if input_format == "bgr_planar":
                rgb = rgb[:, :, ::-1]
        elif input_format == "gray":
            rgb = np.repeat(im[:, :, None], 3, axis=2)
        ref_coefs, _ = oracle.forward(rgb, sub, q)
        got = enc.coefficients(i)
        assert len(got) == len(ref_coefs)
        for c, (g, r) in enumerate(zip(got, ref_coefs)):
            rh, rw = g.shape[:2]
            assert np.array_equal(g, r[:rh, :rw]), f"image {i} component {c} ({sub}, q{q})"
        assert streams[i] == oracle.encode(rgb, sub, q, restart_interval=kw.get("restart_interval", 0)), f"image {i} bitstream"
    return streams


def test_golden_inputs_reproduce_libjpeg_turbo_bitstreams(enc, torch_mod):
    cases = [(e, *load_encode_case(e)) for e in _M["encode"] if e["sub"] != "gray"]
    images = [c[1] for c in cases]
    streams = _check_against_oracle(enc, torch_mod, images, [c[0]["sub"] for c in cases], [c[0]["quality"] for c in cases])
    for (e, rgb, jpeg), s in zip(cases, streams):
        assert oracle.scan_bytes(s) == oracle.scan_bytes(jpeg), e["name"]


def test_gray_goldens(enc, torch_mod):
    cases = [(e, *load_encode_case(e)) for e in _M["encode"] if e["sub"] == "gray"]
    grays = [np.ascontiguousarray(c[1][:, :, 0]) for c in cases]
    streams = _check_against_oracle(enc, torch_mod, grays, ["gray"] * len(cases), [c[0]["quality"] for c in cases], input_format="gray")
    for (e, rgb, jpeg), s in zip(cases, streams):
        assert oracle.scan_bytes(s) == oracle.scan_bytes(jpeg), e["name"]


@pytest.mark.parametrize("fmt", ["bgr", "rgb_planar", "bgr_planar"])
def test_input_formats(enc, torch_mod, fmt):
    imgs = [synth_image(w, h, seed=w + h) for (w, h) in ((50, 37), (64, 48), (129, 70), (257, 65))]
    if fmt == "bgr":
        feed = [np.ascontiguousarray(im[:, :, ::-1]) for im in imgs]
    elif fmt == "rgb_planar":
        feed = [np.ascontiguousarray(im.transpose(2, 0, 1)) for im in imgs]
    else:
        feed = [np.ascontiguousarray(im[:, :, ::-1].transpose(2, 0, 1)) for im in imgs]
    _check_against_oracle(enc, torch_mod, feed, ["420", "444", "422", "420"], [90, 75, 50, 95], input_format=fmt)


def test_all_samplings_odd_sizes_and_restart(enc, torch_mod):
    sizes = [(1, 1), (7, 9), (17, 13), (33, 65), (255, 63), (300, 200)]
    for sub in ("444", "422", "420", "440", "411", "410"):
        imgs = [synth_image(w, h, seed=3 * w + h) for (w, h) in sizes]
        _check_against_oracle(enc, torch_mod, imgs, [sub] * len(imgs), [90] * len(imgs))
    imgs = [synth_image(130, 70, seed=9)]
    _check_against_oracle(enc, torch_mod, imgs, ["420"], [90], restart_interval=5)


def test_pitched_input(enc, torch_mod):
    torch = torch_mod
    im = synth_image(131, 47, seed=4)
    for pitch, off in ((131 * 3 + 5, 0), (131 * 3 + 3, 7), (512, 1)):
        buf = torch.zeros(47 * pitch + 64, dtype=torch.uint8, device="cuda")
        view = torch.as_strided(buf, (47, 131, 3), (pitch, 3, 1), storage_offset=off)
        view.copy_(torch.from_numpy(im).cuda())
        s = enc.encode([view], "420", 90)
        assert s[0] == oracle.encode(im, "420", 90)


def test_pair_kernel_ragged_edges(enc, torch_mod):
    """forward_pair_kernel (two lanes per block) takes interleaved RGB/BGR at any base address and pitch (8-byte pieces fetched
    unaligned): padded pitches and tight ones at odd offsets, ragged right / bottom edges (libjpeg's edge replication), images smaller
    than one MCU, widths that end inside / exactly on / one block past a 32-block tile, 4:2:0 / 4:2:2 / 4:4:4, both byte orders --
    and the same views through the one-lane-per-block kernel (HIPJPEG_ENCODE_ONE_LANE_KERNEL), all against the oracle's bytes."""
    import os
    torch = torch_mod
    sizes = [(1, 1), (7, 9), (17, 13), (33, 65), (250, 63), (257, 66), (264, 70), (300, 200), (519, 131)]
    for fmt in ("rgb", "bgr"):
        for sub in ("420", "422", "444"):
            views, refs = [], []
            for (w, h) in sizes:
                im = synth_image(w, h, seed=5 * w + h)
                pitch = (3 * w + 7) // 8 * 8 + 8
                buf = torch.zeros(h * pitch + 64, dtype=torch.uint8, device="cuda")
                assert buf.data_ptr() % 8 == 0
                view = torch.as_strided(buf, (h, w, 3), (pitch, 3, 1))
                src = im[:, :, ::-1] if fmt == "bgr" else im
                view.copy_(torch.from_numpy(np.ascontiguousarray(src)).cuda())
                views.append(view)
                refs.append(im)
            tight = []
            for k, ((w, h), im) in enumerate(zip(sizes, refs)):
                off = (1, 3, 5, 7, 2)[k % 5]
                buf = torch.zeros(h * w * 3 + 64, dtype=torch.uint8, device="cuda")
                view = torch.as_strided(buf, (h, w, 3), (w * 3, 3, 1), storage_offset=off)
                src = im[:, :, ::-1] if fmt == "bgr" else im
                view.copy_(torch.from_numpy(np.ascontiguousarray(src)).cuda())
                tight.append(view)
            expected = [oracle.encode(im, sub, 85) for im in refs]
            for what, vs in (("padded", views), ("tight", tight)):
                streams = enc.encode(vs, subsampling=sub, quality=85, input_format=fmt)
                for (w, h), s, e in zip(sizes, streams, expected):
                    assert s == e, f"{what} {w}x{h} {sub} {fmt}"
            os.environ["HIPJPEG_ENCODE_ONE_LANE_KERNEL"] = "1"
            try:
                streams = enc.encode(tight, subsampling=sub, quality=85, input_format=fmt)
            finally:
                del os.environ["HIPJPEG_ENCODE_ONE_LANE_KERNEL"]
            for (w, h), s, e in zip(sizes, streams, expected):
                assert s == e, f"one-lane kernel {w}x{h} {sub} {fmt}"


def test_config2_1080p_420_q90_batch_and_roundtrip(enc, torch_mod):
    """BASELINE.json configs[2] shape (reduced batch): 1920x1080 RGB -> q90 4:2:0; then decode our own streams on the GPU:
    encode->decode must equal the oracle's encode->decode (size-independent round-trip property at full size)."""
    torch = torch_mod
    from nvimagecodec_amd.lowlevel import BatchDecoder
    imgs = [synth_image(1920, 1080, seed=40 + s) for s in range(2)]
    streams = _check_against_oracle(enc, torch, imgs * 3, ["420"] * 6, [90] * 6)
    dec = BatchDecoder(0, 4)
    outs, _ = dec.decode(streams[:2])
    torch.cuda.synchronize()
    for s, o in zip(streams[:2], outs):
        assert np.array_equal(o.cpu().numpy(), oracle.decode(s))
    dec.close()


def test_optimized_huffman_and_bad_params(enc, torch_mod):
    torch = torch_mod
    from nvimagecodec_amd import _native as N
    im = synth_image(96, 64, seed=5)
    t = torch.from_numpy(im).cuda()
    std = enc.encode([t], "420", 90)[0]
    opt = enc.encode([t], "420", 90, optimized_huffman=True)[0]
    assert len(opt) < len(std)
    a, _ = oracle.decode_coefficients(std)
    b, _ = oracle.decode_coefficients(opt)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # gray input cannot produce a colour stream
    g = torch.from_numpy(np.ascontiguousarray(im[:, :, 0])).cuda()
    st = enc.device_stage([g], "420", 90, input_format="gray")
    assert st == [3]  # UNSUPPORTED


def test_progressive_output_equals_libjpeg_turbo_files(enc, torch_mod):
    """progressive=1 (nvimgcodecJpegImageInfo_t::encoding = PROGRESSIVE_DCT_HUFFMAN through the plugin): device stage as ever, then
    the host coder's SOF2 writer -- whole files byte-identical to libjpeg-turbo's progressive output (47 vectors, colour and
    gray, restart intervals), mixed with baseline images in one batch; then a 1080p picture decoded again on the GPU."""
    torch = torch_mod
    with open(os.path.join(GOLDEN, "manifest_encode_prog.json")) as f:
        entries = json.load(f)["encode_progressive"]

    def load(e):
        rgb = np.fromfile(os.path.join(GOLDEN, e["input"]), dtype=np.uint8).reshape(e["height"], e["width"], 3)
        with open(os.path.join(GOLDEN, "encode_prog", e["name"] + ".jpg"), "rb") as f:
            return rgb, f.read()

    for rst in sorted({e["restart"] for e in entries}):
        for gray in (False, True):
            cases = [(e, *load(e)) for e in entries if e["restart"] == rst and (e["sub"] == "gray") == gray]
            if not cases:
                continue
            feed = [torch.from_numpy(np.ascontiguousarray(c[1][:, :, 0] if gray else c[1])).cuda() for c in cases]
            out = enc.encode(feed, subsampling=[c[0]["sub"] for c in cases], quality=[c[0]["quality"] for c in cases],
                             input_format="gray" if gray else "rgb", restart_interval=rst, progressive=True)
            for (e, rgb, jpeg), s in zip(cases, out):
                assert s == jpeg, e["name"]
    from nvimagecodec_amd.lowlevel import BatchDecoder
    im = synth_image(1920, 1080, seed=77)
    prog = enc.encode([torch.from_numpy(im).cuda()], "420", 90, progressive=True)[0]
    base = oracle.encode(im, "420", 90)
    assert b"\xff\xc2" in prog[:700] and len(prog) < len(base)
    dec = BatchDecoder(0, 4)
    outs, st = dec.decode([prog])
    torch.cuda.synchronize()
    assert list(st) == [0] and np.array_equal(outs[0].cpu().numpy(), oracle.decode(base))
    dec.close()
    try:
        import io
        from PIL import Image
    except ImportError:
        return
    b = io.BytesIO()
    Image.fromarray(im).save(b, "JPEG", quality=90, subsampling=2, progressive=True)
    assert prog == b.getvalue()


def test_pipelined_submit_wait(enc, torch_mod):
    """hipjpegEncodeBatchSubmit / Wait: two batches in flight, results identical to the one-shot call, in submission order."""
    torch = torch_mod
    batches = [[synth_image(97 + 16 * k, 61 + 8 * k, seed=10 * k + j) for j in range(5)] for k in range(5)]
    dev = [[torch.from_numpy(im).cuda() for im in b] for b in batches]
    subs = ["420", "444", "422", "420", "440"]
    results = []
    for k, d in enumerate(dev):
        enc.submit(d, subs[k], 80 + k)
        if k > 0:
            results.append(enc.wait())
    results.append(enc.wait())
    for k, (statuses, streams) in enumerate(results):
        assert statuses == [0] * 5
        for im, s in zip(batches[k], streams):
            assert s == oracle.encode(im, subs[k], 80 + k), k


def _planes_like_libjpeg(rgb, hs, vs):
    """Y / Cb / Cr planes at component size for dimensions that are multiples of the MCU (no edge padding involved):
    jccolor.c fixed-point conversion, jcsample.c box filter with the alternating bias."""
    r, g, b = [rgb[:, :, i].astype(np.int64) for i in range(3)]
    y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16
    cb = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16
    cr = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16

    def down(p):
        h, w = p.shape
        if hs == 1 and vs == 1:
            return p
        if hs == 2 and vs == 1:
            bias = np.arange(w // 2) & 1
            return (p[:, 0::2] + p[:, 1::2] + bias) >> 1
        bias = 1 + (np.arange(w // 2) & 1)
        return (p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2] + bias) >> 2
    return [y.astype(np.uint8), down(cb).astype(np.uint8), down(cr).astype(np.uint8)]


@pytest.mark.parametrize("sub,hs,vs", [("420", 2, 2), ("422", 2, 1), ("444", 1, 1)])
def test_planar_ycbcr_input_goes_into_the_stream_as_it_is(sub, hs, vs):
    """NVIMGCODEC_SAMPLEFORMAT_P_YUV (reference extensions/nvjpeg/cuda_encoder.cpp:109,362-368 -> nvjpegEncodeYUV): planes that
    are the components already.  (i) MCU-multiple sizes: fed with the planes libjpeg itself would have made from an RGB picture,
    the file must be libjpeg-turbo's file for that picture (oracle.encode, pinned by the encode goldens).  (ii) ragged sizes:
    coefficients equal the oracle's restatement of the same rule (replicate the plane's last column / row) -- parity unpinned,
    nvJPEG's padding rule is not in the reference."""
    import torch
    from nvimagecodec_amd.lowlevel import BatchEncoder
    enc = BatchEncoder(device=0, num_threads=2)
    for gpu_huffman in (False, True):
        enc.gpu_huffman = gpu_huffman
        rgb = synth_image(160, 96, seed=11)
        planes = _planes_like_libjpeg(rgb, hs, vs)
        out = enc.encode([[torch.from_numpy(p).cuda() for p in planes]], sub, 90, "yuv_planar")
        assert out[0] == oracle.encode(rgb, sub, 90), (sub, gpu_huffman)
    rng = np.random.default_rng(5)
    for (w, h) in ((50, 37), (17, 13), (129, 71)):
        planes = [rng.integers(0, 256, (h, w), dtype=np.uint8), rng.integers(0, 256, ((h + vs - 1) // vs, (w + hs - 1) // hs), dtype=np.uint8),
                  rng.integers(0, 256, ((h + vs - 1) // vs, (w + hs - 1) // hs), dtype=np.uint8)]
        enc.gpu_huffman = False
        out = enc.encode([[torch.from_numpy(p).cuda() for p in planes]], sub, 75, "yuv_planar")
        ref, _ = oracle.forward_planes(planes, w, h, sub, 75)
        got, _ = oracle.decode_coefficients(out[0])
        for c in range(3):
            assert np.array_equal(got[c], ref[c]), (sub, w, h, c)
    enc.close()


@pytest.mark.parametrize("fmt", ["rgb_planar", "bgr_planar"])
def test_pair_kernel_planar_input(enc, torch_mod, fmt):
    """Planar RGB / BGR (CHW tensors) through forward_pair_kernel<.., PLANAR>: ragged edges, images smaller than one MCU, widths around a
    32-block tile, every sampling it takes -- and the same tensors through the one-lane-per-block kernel; all against the oracle's bytes."""
    import os
    torch = torch_mod
    sizes = [(1, 1), (7, 9), (17, 13), (33, 65), (250, 63), (257, 66), (264, 70), (300, 200), (519, 131)]
    for sub in ("420", "422", "444"):
        refs = [synth_image(w, h, seed=7 * w + h) for (w, h) in sizes]
        feed = [torch.from_numpy(np.ascontiguousarray((im[:, :, ::-1] if fmt == "bgr_planar" else im).transpose(2, 0, 1))).cuda() for im in refs]
        expected = [oracle.encode(im, sub, 85) for im in refs]
        streams = enc.encode(feed, subsampling=sub, quality=85, input_format=fmt)
        for (w, h), s, e in zip(sizes, streams, expected):
            assert s == e, f"pair kernel {w}x{h} {sub} {fmt}"
        os.environ["HIPJPEG_ENCODE_ONE_LANE_KERNEL"] = "1"
        try:
            streams = enc.encode(feed, subsampling=sub, quality=85, input_format=fmt)
        finally:
            del os.environ["HIPJPEG_ENCODE_ONE_LANE_KERNEL"]
        for (w, h), s, e in zip(sizes, streams, expected):
            assert s == e, f"one-lane kernel {w}x{h} {sub} {fmt}"


def test_optimized_huffman_on_the_gpu_coder_equals_the_host_coder(torch_mod):
    """optimized_huffman (nvimgcodecJpegEncodeParams_t, reference extensions/nvjpeg/cuda_encoder.cpp:348-357) on the GPU entropy coder: symbol
    statistics on the device, jpeg_gen_optimal_table on the host, coding with the image's own tables on the device.  Files byte-identical
    to the host coder's (itself pinned to libjpeg-turbo's optimized output through the decode goldens' tables and the progressive files),
    every sampling, gray, odd sizes, mixed in one batch with Annex-K images; and the files decode to the same coefficients as the plain ones."""
    torch = torch_mod
    from nvimagecodec_amd.lowlevel import BatchEncoder
    shapes = [(96, 64, "420"), (333, 217, "444"), (640, 480, "422"), (17, 13, "420"), (250, 250, "gray"), (1280, 720, "420"), (64, 200, "440")]
    imgs, subs = [], []
    for i, (w, h, sub) in enumerate(shapes):
        im = synth_image(w, h, seed=300 + i)
        if i % 3 == 2:  # noise: long codes, every run/size symbol in use
            im = np.random.default_rng(i).integers(0, 256, size=im.shape, dtype=np.uint8)
        imgs.append(np.ascontiguousarray(im[:, :, 1]) if sub == "gray" else im)
        subs.append(sub)
    host = BatchEncoder(0, num_threads=4, gpu_huffman=False)
    gpu = BatchEncoder(0, num_threads=4, gpu_huffman=True)
    try:
        for fmt, idx in (("rgb", [i for i, s in enumerate(subs) if s != "gray"]), ("gray", [i for i, s in enumerate(subs) if s == "gray"])):
            feed = [torch.from_numpy(imgs[i]).cuda() for i in idx]
            ssub = [subs[i] for i in idx]
            for q in (35, 90, 100):
                want = host.encode(feed, ssub, q, input_format=fmt, optimized_huffman=True)
                got = gpu.encode(feed, ssub, q, input_format=fmt, optimized_huffman=True)
                assert gpu.stats()["gpu_entropy_images"] == len(feed)   # coded on the device, none handed to the host coder
                plain = gpu.encode(feed, ssub, q, input_format=fmt)
                for k, (a, b, c) in enumerate(zip(want, got, plain)):
                    assert a == b, (fmt, q, idx[k])
                    assert len(b) <= len(c)
                    ca, cb = oracle.decode_coefficients(b)[0], oracle.decode_coefficients(c)[0]
                    assert all(np.array_equal(x, y) for x, y in zip(ca, cb))
    finally:
        host.close()
        gpu.close()
