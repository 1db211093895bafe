"""Dev tool: pipelined end-to-end decode (host JPEG bytes -> RGB in HBM, three batches in flight) for other picture sizes and batch
sizes than the bench's -- e.g. the 500 x 375 pictures of an ImageNet-style loader.
usage: python tools/e2e_sizes.py WIDTH HEIGHT BATCH [quality]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nvimagecodec_amd.lowlevel import BatchDecoder
from nvimagecodec_amd.synth import synth_image

W, H, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
Q = int(sys.argv[4]) if len(sys.argv) > 4 else 90
src = [bench._pil_encode(synth_image(W, H, seed=300 + s), Q, "420") for s in range(16)]
jpegs = [src[i % len(src)] for i in range(B)]
dec = BatchDecoder(0, bench.usable_cpus())
outs = [dec.allocate_outputs(jpegs) for _ in range(3)]
for k in range(3):
    dec.submit(jpegs, outs[k])
for k in range(3):
    dec.wait()
torch.cuda.synchronize()
K = 60
ts = []
t0 = time.perf_counter()
for i in range(K):
    a = time.perf_counter()
    dec.submit(jpegs, outs[i % 3])
    if i > 1:
        dec.wait()
    ts.append(time.perf_counter() - a)
dec.wait(); dec.wait()
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / K
ts.sort()
print("%dx%d q%d, batch %d (%.0f KB per file): %.3f ms per batch (median iteration %.3f) = %.0f images/s, %.0f MP/s" % (
    W, H, Q, B, sum(len(j) for j in jpegs) / B / 1024, t * 1e3, ts[K // 2] * 1e3, B / t, B * W * H / t / 1e6))
