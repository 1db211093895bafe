"""Large and extreme shapes (dev tool, GPU box): very wide, very tall, 4K/8K-class images, qualities 1..100, every fused sampling,
with and without restart intervals -- GPU entropy route vs host entropy route of the product (bit for bit), a sample against
the CPU oracle.  Many subsequences per image: the multi-workgroup synchronisation of the entropy stage works hardest here."""
import sys, os, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from nvimagecodec_amd.lowlevel import BatchDecoder, BatchEncoder
from nvimagecodec_amd.synth import synth_image

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = random.Random(seed)
dec = BatchDecoder(0, 8)
enc = BatchEncoder(0, num_threads=8, gpu_huffman=True)
n = 0
t0 = time.time()
for rnd in range(rounds):
    shapes = [rng.choice([(rng.randrange(3000, 8200), rng.randrange(8, 64)), (rng.randrange(8, 64), rng.randrange(3000, 8200)),
                          (rng.randrange(2000, 4200), rng.randrange(1500, 3200)), (rng.randrange(500, 2000), rng.randrange(500, 2000))]) for _ in range(6)]
    imgs = [synth_image(w, h, seed=rng.randrange(1 << 30)) for (w, h) in shapes]
    subs = [rng.choice(["420", "422", "444"]) for _ in imgs]
    quals = [rng.choice([1, 5, 30, 75, 90, 98, 100]) for _ in imgs]
    ri = rng.choice([0, 0, 1, 9, 400])
    feeds = [torch.from_numpy(im).cuda() for im in imgs]
    jpegs = enc.encode(feeds, subsampling=subs, quality=quals, restart_interval=ri)
    outs_g, st_g = dec.decode(jpegs, fmt="rgb", gpu_huffman=True)
    torch.cuda.synchronize()
    got = [o.cpu().numpy().copy() for o in outs_g]
    stats = dec.stats()
    outs_h, st_h = dec.decode(jpegs, fmt="rgb", gpu_huffman=False)
    torch.cuda.synchronize()
    assert all(s == 0 for s in st_g) and all(s == 0 for s in st_h), (st_g, st_h)
    for i, (a, b) in enumerate(zip(got, outs_h)):
        assert np.array_equal(a, b.cpu().numpy()), ("decode routes disagree", rnd, i, shapes[i], subs[i], quals[i], ri)
    i = rng.randrange(len(jpegs))
    if shapes[i][0] * shapes[i][1] < 6_000_000:
        assert np.array_equal(got[i], oracle.decode(jpegs[i])), ("decode vs oracle", rnd, i, shapes[i])
        assert jpegs[i] == oracle.encode(imgs[i], subs[i], quals[i], restart_interval=ri), ("encode vs oracle", rnd, i, shapes[i])
    n += len(jpegs)
    print("round %d ok (%d images, gpu-entropy images %d, sync launches %d, %.1f s)" % (rnd, n, stats["gpu_entropy_images"], stats["sync_launches"], time.time() - t0), flush=True)
print("large-shape campaign ok", n)
