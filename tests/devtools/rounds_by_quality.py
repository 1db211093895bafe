"""Dev tool: correction rounds of the GPU entropy stage by quality / sampling (HIPJPEG_DEBUG_TIMING=1 prints them per batch), and whether
any image was handed to the host decoder."""
import os
os.environ.setdefault("HIPJPEG_ENABLE_TEST_HOOKS", "1")
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["HIPJPEG_DEBUG_TIMING"] = "1"
import torch
import oracle
from nvimagecodec_amd.lowlevel import BatchDecoder
from nvimagecodec_amd.synth import synth_image

dec = BatchDecoder(0, 8)
imgs = [synth_image(1920, 1080, seed=60 + k) for k in range(4)]
noise = np.random.default_rng(3).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
for q in (90, 95, 98):
    for sub in ("420", "444"):
        jpegs = [oracle.encode(im, sub, q) for im in imgs] * 8
        try:
            jpegs += [oracle.encode(noise, sub, q)] * 2
        except oracle.OracleError:  # the oracle's output buffer is sized for pictures, not for noise at the highest qualities
            continue
        print("== quality", q, sub, "bytes/image", len(jpegs[0]), "noise", len(jpegs[-1]), flush=True)
        outs, st = dec.decode(jpegs, gpu_huffman=True, check=False)
        torch.cuda.synchronize()
        print("   statuses ok:", all(s == 0 for s in st), "host fallbacks:", dec.host_fallbacks(), "sync launches:", dec.stats()["sync_launches"], flush=True)
