"""Random output pitches / base alignments / byte orders x widths around the tile boundaries (dev tool, GPU box): interleaved
outputs with arbitrary pitch and offset, checked against the oracle INCLUDING the bytes around the image (nothing outside
the image rows may be written) -- the staged 16-byte store path, the direct 8-byte path and the byte path of the luma kernel,
wide and narrow tiles."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from nvimagecodec_amd.lowlevel import BatchDecoder
from nvimagecodec_amd.synth import synth_image

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = random.Random(seed)
dec = BatchDecoder(0, 8)
n = 0
for rnd in range(rounds):
    fmt = rng.choice(["rgb", "bgr"])
    jpegs, views, bufs, meta = [], [], [], []
    for _ in range(16):
        w = rng.choice([8 * rng.randrange(1, 100) - rng.randrange(0, 8), rng.randrange(1, 800), 16 * rng.randrange(1, 40)])
        h = rng.randrange(1, 200)
        sub = rng.choice(["420", "422", "444", "420"])
        j = oracle.encode(synth_image(w, h, seed=rng.randrange(1 << 30)), sub, rng.choice([60, 90]))
        pitch = 3 * w + rng.choice([0, 0, 1, 3, 5, 8, 13, 16, 29])
        if rng.random() < 0.4:
            pitch = (pitch + 15) // 16 * 16
        off = rng.choice([0, 0, 16, 1, 3, 8, 5])
        buf = torch.full((h * pitch + 64,), 0xAB, dtype=torch.uint8, device="cuda")
        view = torch.as_strided(buf, (h, w, 3), (pitch, 3, 1), storage_offset=off)
        jpegs.append(j); views.append(view); bufs.append(buf); meta.append((w, h, pitch, off))
    _, st = dec.decode(jpegs, fmt=fmt, outs=views, gpu_huffman=rng.random() < 0.5)
    torch.cuda.synchronize()
    assert all(s == 0 for s in st)
    for j, buf, (w, h, pitch, off) in zip(jpegs, bufs, meta):
        ref = oracle.decode(j, oracle.FMT_BGR if fmt == "bgr" else oracle.FMT_RGB)
        host = buf.cpu().numpy()
        got = np.lib.stride_tricks.as_strided(host[off:], (h, w, 3), (pitch, 3, 1))
        assert np.array_equal(got, ref), ("pixels", rnd, w, h, pitch, off, fmt)
        mask = np.ones(host.shape, dtype=bool)
        for y in range(h):
            mask[off + y * pitch: off + y * pitch + w * 3] = False
        assert np.all(host[mask] == 0xAB), ("wrote outside the image", rnd, w, h, pitch, off, fmt)
    n += len(jpegs)
print("output campaign ok", n)
