"""Randomized cross-check campaign (dev tool, run on the GPU box): many valid streams of random shape / quality / sampling /
restart interval / Huffman tables through independent routes of the product, which must agree bit for bit:
  decode: GPU entropy stage vs host entropy stage (same pixel kernels), a sample also against the CPU oracle;
  encode: two-lanes-per-block kernel (aligned input, and the same pixels at a misaligned base) vs one-lane-per-block kernel,
          GPU entropy coder vs host entropy coder; a sample also against the CPU oracle."""
import sys, os, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from nvimagecodec_amd.lowlevel import BatchDecoder, BatchEncoder
from nvimagecodec_amd.synth import synth_image

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = random.Random(seed)
dec = BatchDecoder(0, 8)
enc_g = BatchEncoder(0, num_threads=8, gpu_huffman=True)
enc_h = BatchEncoder(0, num_threads=8, gpu_huffman=False)
t0 = time.time()
n_dec = n_enc = 0
for rnd in range(rounds):
    # ---- encode side
    imgs, subs, quals = [], [], []
    for _ in range(40):
        w = rng.choice([rng.randrange(1, 64), rng.randrange(64, 700), 8 * rng.randrange(1, 90), 16 * rng.randrange(1, 45)])
        h = rng.choice([rng.randrange(1, 64), rng.randrange(64, 500), 8 * rng.randrange(1, 60)])
        imgs.append(synth_image(w, h, seed=rng.randrange(1 << 30)))
        subs.append(rng.choice(["420", "420", "422", "444"]))
        quals.append(rng.choice([rng.randrange(1, 101), 90, 75, 100, 50]))
    aligned, skewed = [], []
    for im in imgs:
        h, w, _ = im.shape
        pitch = (3 * w + 7) // 8 * 8
        buf = torch.zeros(h * pitch + 64, dtype=torch.uint8, device="cuda")
        a = torch.as_strided(buf, (h, w, 3), (pitch, 3, 1)); a.copy_(torch.from_numpy(im).cuda()); aligned.append(a)
        buf2 = torch.zeros(h * pitch + 64, dtype=torch.uint8, device="cuda")
        b = torch.as_strided(buf2, (h, w, 3), (pitch, 3, 1), storage_offset=3); b.copy_(torch.from_numpy(im).cuda()); skewed.append(b)
    s_pair_g = enc_g.encode(aligned, subsampling=subs, quality=quals)
    # the same pixels at a base address that is not a multiple of 8: once through the pair kernel (it takes any alignment), once through
    # the one-lane-per-block kernel (HIPJPEG_ENCODE_ONE_LANE_KERNEL is read per batch) -- three routes, one answer
    s_pair_skewed = enc_g.encode(skewed, subsampling=subs, quality=quals)
    os.environ["HIPJPEG_ENCODE_ONE_LANE_KERNEL"] = "1"
    try:
        s_gen_h = enc_h.encode(skewed, subsampling=subs, quality=quals)
    finally:
        del os.environ["HIPJPEG_ENCODE_ONE_LANE_KERNEL"]
    for i, (x, y) in enumerate(zip(s_pair_g, s_pair_skewed)):
        assert x == y, ("pair kernel: aligned and unaligned input disagree", rnd, i, imgs[i].shape, subs[i], quals[i])
    for i, (x, y) in enumerate(zip(s_pair_g, s_gen_h)):
        assert x == y, ("encode routes disagree", rnd, i, imgs[i].shape, subs[i], quals[i])
    for i in rng.sample(range(len(imgs)), 4):
        assert s_pair_g[i] == oracle.encode(imgs[i], subs[i], quals[i]), ("encode vs oracle", rnd, i, imgs[i].shape, subs[i], quals[i])
    n_enc += len(imgs)
    # ---- decode side: our own files plus restart-interval / optimized-table variants from the oracle encoder
    jpegs = list(s_pair_g)
    for i in range(12):
        im = imgs[rng.randrange(len(imgs))]
        jpegs.append(oracle.encode(im, rng.choice(["420", "422", "444"]), rng.choice([30, 60, 85, 95]), restart_interval=rng.choice([1, 2, 3, 7, 16, 100])))
    opt = [aligned[rng.randrange(len(aligned))] for _ in range(8)]
    jpegs += enc_h.encode(opt, subsampling=[rng.choice(["420", "444", "422"]) for _ in opt], quality=[rng.choice([40, 80, 97]) for _ in opt],
                          optimized_huffman=True)  # per-image Huffman tables (the host coder writes them)
    fmt = rng.choice(["rgb", "bgr", "rgb_planar"])
    outs_g, st_g = dec.decode(jpegs, fmt=fmt, gpu_huffman=True)
    torch.cuda.synchronize()
    got = [o.cpu().numpy().copy() for o in outs_g]
    outs_h, st_h = dec.decode(jpegs, fmt=fmt, gpu_huffman=False)
    torch.cuda.synchronize()
    assert all(s == 0 for s in st_g) and all(s == 0 for s in st_h), (st_g, st_h)
    for i, (a, b) in enumerate(zip(got, outs_h)):
        assert np.array_equal(a, b.cpu().numpy()), ("decode routes disagree", rnd, i, a.shape, fmt)
    for i in rng.sample(range(len(jpegs)), 5):
        ref = oracle.decode(jpegs[i], oracle.FMT_BGR if fmt == "bgr" else oracle.FMT_RGB)
        if fmt.endswith("planar"):
            ref = ref.transpose(2, 0, 1)
        assert np.array_equal(got[i], ref), ("decode vs oracle", rnd, i, fmt)
    n_dec += len(jpegs)
    print("round %d ok: %d encodes, %d decodes so far, %.1f s, gpu-entropy images in last batch: %d" % (rnd, n_enc, n_dec, time.time() - t0, dec.stats()["gpu_entropy_images"]), flush=True)
print("fuzz campaign ok", n_enc, n_dec)
