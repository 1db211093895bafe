"""Damaged PROGRESSIVE streams (dev tool, run on the GPU box): bit flips, overwrites, truncations, cuts, garbage tails and header damage in
SOF2 files of every sampling and scan script the goldens have.  The GPU walk + replay must reach the same verdict as the host entropy
stage, identical pixels where both decode, and no kernel may fault or hang (run it under `timeout`).  Undamaged files are mixed in so that a
batch always has images the walker takes; the count of GPU-decoded images is printed.
usage: python tests/campaigns/fuzz_progressive.py [seed [rounds]]"""
import glob
import io
import os
import random
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import oracle
from nvimagecodec_amd.lowlevel import BatchDecoder
from nvimagecodec_amd.synth import synth_image

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = random.Random(seed)
dec = BatchDecoder(0, 8)
here = os.path.dirname(os.path.abspath(__file__))
bases = [open(p, "rb").read() for p in sorted(glob.glob(os.path.join(here, "..", "golden", "decode", "*prog*.jpg"))) if os.path.getsize(p) > 1500]
try:
    from PIL import Image
    for k, (w, h, sub, q) in enumerate(((640, 360, 2, 90), (321, 243, 1, 75), (200, 200, 0, 95), (400, 304, 2, 50), (1280, 720, 0, 92))):
        b = io.BytesIO()
        Image.fromarray(synth_image(w, h, seed=70 + k)).save(b, "JPEG", quality=q, subsampling=sub, progressive=True)
        bases.append(b.getvalue())
except ImportError:
    pass
assert len(bases) >= 10, len(bases)
t0 = time.time()
total = agree_ok = agree_bad = gpu_taken = 0
for rnd in range(rounds):
    jpegs, kinds = [], []
    for n in range(48):
        b = bytearray(rng.choice(bases))
        first_sos = bytes(b).find(b"\xff\xda")
        kind = rng.randrange(8)
        if kind == 0:
            for _ in range(rng.randrange(1, 8)):
                i = rng.randrange(first_sos, len(b) - 2)
                b[i] ^= 1 << rng.randrange(8)
        elif kind == 1:
            i = rng.randrange(first_sos, len(b) - 40)
            for k in range(rng.randrange(1, 32)):
                b[i + k] = rng.randrange(256)
        elif kind == 2:
            b = b[: rng.randrange(first_sos + 1, len(b) - 2)] + b"\xff\xd9"
        elif kind == 3:
            i = rng.randrange(first_sos, len(b) - 200)
            del b[i: i + rng.randrange(1, 150)]
        elif kind == 4:
            b = b[:-2] + bytes(rng.randrange(256) for _ in range(rng.randrange(1, 300))) + b"\xff\xd9"
        elif kind == 5:  # damage the header tables / frame fields a little
            i = rng.randrange(20, first_sos - 1)
            b[i] ^= 1 << rng.randrange(8)
        # kinds 6, 7: left whole
        jpegs.append(bytes(b))
        kinds.append(kind)
    keep = []
    for j, k in zip(jpegs, kinds):
        try:
            dec.allocate_outputs([j])
            keep.append((j, k))
        except Exception:
            pass  # a header too broken to size an output
    if not keep:
        continue
    jpegs, kinds = [j for j, _ in keep], [k for _, k in keep]
    outs_g = dec.allocate_outputs(jpegs)
    _, st_gpu = dec.decode(jpegs, outs=outs_g, gpu_huffman=True, check=False)
    torch.cuda.synchronize()
    gpu_taken += dec.stats()["gpu_entropy_images"]
    got = [o.cpu().numpy().copy() if o is not None else None for o in outs_g]
    _, st_cpu = dec.decode(jpegs, outs=outs_g, gpu_huffman=False, check=False)
    torch.cuda.synchronize()
    if [s == 0 for s in st_gpu] != [s == 0 for s in st_cpu]:
        os.makedirs("gpurun_out/fuzz_fail", exist_ok=True)
        for i, (a, b) in enumerate(zip(st_gpu, st_cpu)):
            if (a == 0) != (b == 0):
                print("DISAGREE round %d index %d: gpu %d host %d kind %s len %d" % (rnd, i, a, b, kinds[i], len(jpegs[i])), flush=True)
                open("gpurun_out/fuzz_fail/p%d_r%d_i%d.jpg" % (seed, rnd, i), "wb").write(jpegs[i])
        raise SystemExit(1)
    for i, (s, o) in enumerate(zip(st_cpu, outs_g)):
        if s == 0:
            if not np.array_equal(got[i], o.cpu().numpy()):
                os.makedirs("gpurun_out/fuzz_fail", exist_ok=True)
                open("gpurun_out/fuzz_fail/p%d_r%d_i%d_pixels.jpg" % (seed, rnd, i), "wb").write(jpegs[i])
                print("PIXELS DIFFER round %d index %d kind %s" % (rnd, i, kinds[i]), flush=True)
                raise SystemExit(1)
            agree_ok += 1
        else:
            agree_bad += 1
    total += len(jpegs)
    print("round %d: %d streams, %d decodable, %d rejected, %d taken by the GPU walk, %.1f s" % (rnd, total, agree_ok, agree_bad, gpu_taken, time.time() - t0), flush=True)
outs, _ = dec.decode(bases[:2], gpu_huffman=True)
torch.cuda.synchronize()
assert np.array_equal(outs[0].cpu().numpy(), oracle.decode(bases[0]))
print("progressive damage campaign ok", total, agree_ok, agree_bad, gpu_taken)
