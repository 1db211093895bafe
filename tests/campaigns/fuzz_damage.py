"""Damaged-stream campaign (dev tool, run on the GPU box): bit flips, overwrites, truncations, cuts and garbage tails.  The GPU
entropy stage must reach the same verdict as the host entropy stage, identical pixels where both decode, and no kernel may
fault or hang.  (tests/test_gpu_huffman.py holds a fixed-seed version of this.)"""
import sys, os, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from nvimagecodec_amd.lowlevel import BatchDecoder
from nvimagecodec_amd.synth import synth_image

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = random.Random(seed)
dec = BatchDecoder(0, 8)
bases = [oracle.encode(synth_image(w, h, seed=s), sub, q, restart_interval=ri) for (w, h, sub, q, s, ri) in
         ((640, 360, "420", 90, 1, 0), (321, 243, "422", 75, 2, 0), (200, 200, "444", 95, 3, 0), (512, 64, "gray", 60, 4, 0),
          (400, 300, "420", 85, 5, 7), (1280, 720, "420", 92, 6, 0), (333, 222, "420", 30, 7, 1))]
t0 = time.time()
total = agree_ok = agree_bad = 0
for rnd in range(rounds):
    jpegs, kinds = [], []
    for _ in range(64):
        b = bytearray(rng.choice(bases))
        sos = bytes(b).rfind(b"\xff\xda") + 14
        kind = rng.randrange(6)
        if kind == 0:
            for _ in range(rng.randrange(1, 8)):
                i = rng.randrange(sos, len(b) - 2); b[i] ^= 1 << rng.randrange(8)
        elif kind == 1:
            i = rng.randrange(sos, len(b) - 40)
            for k in range(rng.randrange(1, 32)): b[i + k] = rng.randrange(256)
        elif kind == 2:
            b = b[: rng.randrange(sos + 1, len(b) - 2)] + b"\xff\xd9"
        elif kind == 3:
            i = rng.randrange(sos, len(b) - 200); del b[i : i + rng.randrange(1, 150)]
        elif kind == 4:
            b = b[:-2] + bytes(rng.randrange(256) for _ in range(rng.randrange(1, 300))) + b"\xff\xd9"
        else:  # damage the header tables / frame fields a little
            i = rng.randrange(20, sos - 1); b[i] ^= 1 << rng.randrange(8)
        jpegs.append(bytes(b))
        kinds.append(kind)
    try:
        outs_g = dec.allocate_outputs(jpegs)
    except Exception:
        # a header too broken to size an output: decode one by one would report it; skip the batch entry-wise
        keep = []
        for j in jpegs:
            try:
                dec.allocate_outputs([j]); keep.append(j)
            except Exception:
                pass
        jpegs = keep
        if not jpegs:
            continue
        outs_g = dec.allocate_outputs(jpegs)
    _, st_gpu = dec.decode(jpegs, outs=outs_g, gpu_huffman=True, check=False)
    torch.cuda.synchronize()
    got = [o.cpu().numpy().copy() if o is not None else None for o in outs_g]
    _, st_cpu = dec.decode(jpegs, outs=outs_g, gpu_huffman=False, check=False)
    torch.cuda.synchronize()
    if [s == 0 for s in st_gpu] != [s == 0 for s in st_cpu]:
        os.makedirs("gpurun_out/fuzz_fail", exist_ok=True)
        for i, (a, b) in enumerate(zip(st_gpu, st_cpu)):
            if (a == 0) != (b == 0):
                print("DISAGREE round %d index %d: gpu %d host %d kind %s len %d" % (rnd, i, a, b, kinds[i], len(jpegs[i])), flush=True)
                open("gpurun_out/fuzz_fail/s%d_r%d_i%d.jpg" % (seed, rnd, i), "wb").write(jpegs[i])
        raise SystemExit(1)
    for i, (s, o) in enumerate(zip(st_cpu, outs_g)):
        if s == 0:
            assert np.array_equal(got[i], o.cpu().numpy()), (rnd, i)
            agree_ok += 1
        else:
            agree_bad += 1
    total += len(jpegs)
    print("round %d: %d streams, %d decodable, %d rejected, %.1f s" % (rnd, total, agree_ok, agree_bad, time.time() - t0), flush=True)
outs, _ = dec.decode(bases[:2], gpu_huffman=True)
torch.cuda.synchronize()
assert np.array_equal(outs[0].cpu().numpy(), oracle.decode(bases[0]))
print("damage campaign ok", total, agree_ok, agree_bad)
