"""Random regions of interest x EXIF orientations x image widths around the tile boundaries (dev tool, GPU box): the geometry
pass on top of the tiled luma kernel must give exactly the oracle's full decode, cropped and turned with numpy."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from nvimagecodec_amd.lowlevel import BatchDecoder
from nvimagecodec_amd.synth import synth_image
from test_gpu_geometry import upright

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = random.Random(seed)
dec = BatchDecoder(0, 8)
n = 0
for rnd in range(rounds):
    jpegs, transforms, expect = [], [], []
    for _ in range(6):
        w = rng.choice([8 * rng.randrange(1, 130) - rng.randrange(0, 8), rng.randrange(1, 1100)])
        h = rng.randrange(1, 500)
        sub = rng.choice(["420", "422", "444", "gray", "420"])
        img = synth_image(w, h, seed=rng.randrange(1 << 30))
        j = oracle.encode(img if sub != "gray" else img[:, :, 1].copy(), sub, rng.choice([50, 85, 95]))
        full = oracle.decode(j)
        for _ in range(6):
            o = rng.randrange(1, 9)
            if rng.random() < 0.8:
                x0 = rng.randrange(0, w); x1 = rng.randrange(x0 + 1, w + 1)
                y0 = rng.randrange(0, h); y1 = rng.randrange(y0 + 1, h + 1)
                roi = (x0, y0, x1, y1)
            else:
                roi = None
            jpegs.append(j); transforms.append((roi, o))
            crop = full if roi is None else full[roi[1]:roi[3], roi[0]:roi[2]]
            expect.append(upright(crop, o))
    gh = rng.random() < 0.5
    outs, st = dec.decode(jpegs, fmt="rgb", gpu_huffman=gh, transforms=transforms)
    torch.cuda.synchronize()
    assert all(s == 0 for s in st), st
    for o, e, t in zip(outs, expect, transforms):
        assert tuple(o.shape) == e.shape, (t, o.shape, e.shape)
        assert np.array_equal(o.cpu().numpy(), e), ("geometry mismatch", rnd, t, e.shape)
    n += len(jpegs)
print("geometry campaign ok", n)
