"""Random encode parameters against the oracle (dev tool, GPU box): input layouts (interleaved / planar / gray, RGB / BGR, odd
pitches), every sampling, restart intervals, qualities 1..100 -- bitstreams must equal the oracle's byte for byte; with
per-image optimized Huffman tables (which the oracle does not write) the decoded pixels must equal the decode of the
oracle's standard-table stream (same coefficients)."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from nvimagecodec_amd.lowlevel import BatchEncoder
from nvimagecodec_amd.synth import synth_image

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = random.Random(seed)
encs = [BatchEncoder(0, num_threads=8, gpu_huffman=True), BatchEncoder(0, num_threads=8, gpu_huffman=False)]
n = 0
for rnd in range(rounds):
    fmt = rng.choice(["rgb", "bgr", "rgb_planar", "bgr_planar", "gray"])
    ri = rng.choice([0, 0, 0, 1, 2, 5, 17])
    opt = rng.random() < 0.25
    imgs, feeds, subs, quals = [], [], [], []
    for _ in range(12):
        w, h = rng.choice([rng.randrange(1, 40), rng.randrange(40, 400)]), rng.choice([rng.randrange(1, 40), rng.randrange(40, 300)])
        im = synth_image(w, h, seed=rng.randrange(1 << 30))
        sub = "gray" if fmt == "gray" else rng.choice(["420", "422", "444", "440", "411", "410"])
        imgs.append(im); subs.append(sub); quals.append(rng.randrange(1, 101))
        if fmt == "gray":
            g = np.ascontiguousarray(im[:, :, 1]); imgs[-1] = np.repeat(g[:, :, None], 3, axis=2)  # R = G = B: Y is the sample
            feeds.append(torch.from_numpy(g).cuda())
        elif fmt.endswith("planar"):
            src = im[:, :, ::-1] if fmt.startswith("bgr") else im
            feeds.append(torch.from_numpy(np.ascontiguousarray(src.transpose(2, 0, 1))).cuda())
        else:
            src = im[:, :, ::-1] if fmt == "bgr" else im
            pitch = 3 * w + rng.choice([0, 0, 3, 5, 8])
            buf = torch.zeros(h * pitch + 64, dtype=torch.uint8, device="cuda")
            v = torch.as_strided(buf, (h, w, 3), (pitch, 3, 1), storage_offset=rng.choice([0, 0, 8, 3]))
            v.copy_(torch.from_numpy(np.ascontiguousarray(src)).cuda()); feeds.append(v)
    enc = encs[rnd & 1]
    streams = enc.encode(feeds, subsampling=subs, quality=quals, input_format=fmt, restart_interval=ri, optimized_huffman=opt)
    for i, s in enumerate(streams):
        ref = oracle.encode(imgs[i], subs[i], quals[i], restart_interval=ri)
        if opt:
            assert np.array_equal(oracle.decode(s), oracle.decode(ref)), ("optimized tables", rnd, i, imgs[i].shape, subs[i], quals[i], fmt, ri)
        else:
            assert s == ref, ("bitstream", rnd, i, imgs[i].shape, subs[i], quals[i], fmt, ri)
    n += len(streams)
print("encode campaign ok", n)
