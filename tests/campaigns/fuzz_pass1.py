"""Randomized campaign (dev tool, run on the GPU box) for the IDCT kernels' int16-lane arithmetic (the SIMD routine of
libjpeg-turbo restated): EVERY picture against the CPU oracle, bit for bit.
  (a) pictures of random shape, sampling and quality 1..100 through the GPU and the host entropy stage;
  (b) files written from chosen coefficients (tests/helpers/jpeg_from_coefficients.py) with the largest dequantized value drawn
      around the int16 edge of 32,767 and far beyond, 8- and 16-bit quantization tables, blocks whose rows 1..7 are empty.
usage: python tests/campaigns/fuzz_pass1.py [seed] [rounds]"""
import os
os.environ.setdefault("HIPJPEG_ENABLE_TEST_HOOKS", "1")  # kernel_flavours / host_fallbacks are test hooks of the library
import random
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import oracle
from nvimagecodec_amd.lowlevel import BatchDecoder
from nvimagecodec_amd.synth import synth_image
from tests.helpers import jpeg_from_coefficients as jc

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = random.Random(seed)
nrng = np.random.default_rng(seed)
dec = BatchDecoder(0, 8)
t0 = time.time()
n = 0
used = np.zeros(4, dtype=np.int64)  # K1, K2 by layout (generic / interleaved / planar)
SAMPLINGS = {"gray": [(1, 1)], "444": [(1, 1)] * 3, "422": [(2, 1), (1, 1), (1, 1)], "420": [(2, 2), (1, 1), (1, 1)], "440": [(1, 2), (1, 1), (1, 1)]}
for rnd in range(rounds):
    jpegs = []
    for _ in range(48):
        w, h = rng.choice([rng.randrange(1, 40), rng.randrange(40, 400)]), rng.choice([rng.randrange(1, 40), rng.randrange(40, 300)])
        sub = rng.choice(["420", "420", "422", "444", "gray"])
        im = synth_image(w, h, seed=rng.randrange(1 << 30))
        if rng.random() < 0.3:  # noise: large high-frequency coefficients
            im = nrng.integers(0, 256, size=im.shape, dtype=np.uint8)
        jpegs.append(oracle.encode(im if sub != "gray" else im[:, :, 1].copy(), sub, rng.choice([rng.randrange(1, 101), rng.randrange(80, 101), 90])))
    for _ in range(16):
        name = rng.choice(list(SAMPLINGS))
        q = rng.choice([1, 2, 16, 31, 32, 33, 34, 64, 128, 255, 256, 4097, 32768, 65535])
        extreme = rng.choice([1023, 1023, 1000, 512, 100, min(1023, 32767 // q), min(1023, 32767 // q + 1)])
        w, h = rng.randrange(8, 120), rng.randrange(8, 90)
        coefs = jc.random_coefficients(nrng, w, h, SAMPLINGS[name], extreme, dense=rng.randrange(0, 12), dc=rng.choice([60, 1000]))
        if rng.random() < 0.4:  # rows 1..7 empty in most blocks: the SIMD routine's shortcut
            for a in coefs:
                keep = nrng.random(a.shape[:2]) < 0.2
                a[~keep, 8:] = 0
        qt = [np.full(64, q, dtype=np.int32) for _ in SAMPLINGS[name]]
        for t in qt:
            t[0] = rng.choice([1, 8, 16, q])
        jpegs.append(jc.write_baseline(w, h, SAMPLINGS[name], coefs, qt))
    fmt = rng.choice(["rgb", "bgr", "rgb_planar", "y"])
    for gh in (True, False):
        outs, st = dec.decode(jpegs, fmt=fmt, gpu_huffman=gh)
        torch.cuda.synchronize()
        assert all(s == 0 for s in st), st
        plane, luma = dec.kernel_flavours()
        used += np.array(plane + luma) > 0
        for i, (j, o) in enumerate(zip(jpegs, outs)):
            ref = oracle.decode(j, {"rgb": oracle.FMT_RGB, "bgr": oracle.FMT_BGR, "rgb_planar": oracle.FMT_RGB, "y": oracle.FMT_GRAY}[fmt])
            if fmt == "rgb_planar":
                ref = ref.transpose(2, 0, 1)
            assert np.array_equal(o.cpu().numpy(), ref), ("decode vs oracle", seed, rnd, i, fmt, gh)
        n += len(jpegs)
    print("round %d ok: %d decodes against the oracle, %.1f s; batches that used K1 / K2 generic / interleaved / planar: %s"
          % (rnd, n, time.time() - t0, used.tolist()), flush=True)
print("pass-1 campaign ok", n)
