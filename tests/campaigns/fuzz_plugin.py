"""Random batches through the nvImageCodec-API route (dev tool, GPU box): batch sizes around the piece boundaries of the plugin's
pipelining, corrupt files sprinkled in, decoder options; every good sample must equal the oracle, every corrupt one must come
back as None, order preserved."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from nvimagecodec_amd import api
from nvimagecodec_amd.synth import synth_image

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = random.Random(seed)
pool = []
for k in range(24):
    w, h = rng.randrange(1, 400), rng.randrange(1, 300)
    sub = rng.choice(["420", "422", "444", "gray", "420"])
    img = synth_image(w, h, seed=k)
    j = oracle.encode(img if sub != "gray" else np.repeat(img[:, :, 1:2], 3, axis=2), sub, rng.choice([50, 90]), restart_interval=rng.choice([0, 0, 3]))
    pool.append((j, oracle.decode(j)))
n = 0
for rnd in range(rounds):
    opts = rng.choice(["", "hipjpeg_decoder:pipeline_chunks=1", "hipjpeg_decoder:pipeline_chunks=2", "hipjpeg_decoder:pipeline_chunks=3",
                       "hipjpeg_decoder:gpu_huffman=0"])
    with api.Decoder(max_num_cpu_threads=rng.choice([1, 4, 8]), options=opts) as dec:
        for rep in range(3):
            count = rng.choice([1, 2, 3, 5, 95, 96, 97, 191, 192, 193, 260, rng.randrange(1, 120)])
            picks = [rng.randrange(len(pool)) for _ in range(count)]
            jpegs, bad = [], set()
            for i, p in enumerate(picks):
                j = pool[p][0]
                if rng.random() < 0.05:
                    j = j[: max(30, len(j) // 3)]
                    bad.add(i)
                jpegs.append(j)
            imgs = dec.decode(jpegs)
            torch.cuda.synchronize()
            assert len(imgs) == count
            for i, (p, im) in enumerate(zip(picks, imgs)):
                if i in bad:
                    assert im is None, ("corrupt sample decoded", rnd, rep, i)
                else:
                    ref = pool[p][1]
                    got = im.cpu()._array
                    exp = ref if got.shape == ref.shape else ref[:, :, :1]
                    assert np.array_equal(got, exp), ("pixels", rnd, rep, i, opts, count, got.shape, ref.shape)
            n += count
print("plugin campaign ok", n)
