"""Profiling driver (dev tool): pipelined encode submit/wait of 256 x 1080p batches.  Run under rocprofv3 --kernel-trace --memory-copy-trace."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvimagecodec_amd.lowlevel import BatchEncoder
from nvimagecodec_amd.synth import synth_image
src = [torch.from_numpy(synth_image(1920, 1080, seed=s)).cuda() for s in range(4)]
imgs = [src[i % 4] for i in range(256)]
enc = BatchEncoder(0, num_threads=0)
for _ in range(3):  # every page sizes its arenas on first use
    enc.submit(imgs, "420", 90, "rgb", gpu_huffman=True)
for _ in range(3):
    enc.wait(fetch=False)
torch.cuda.synchronize(); t0 = time.time()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for i in range(K):
    enc.submit(imgs, "420", 90, "rgb", gpu_huffman=True)
    if i > 1:
        enc.wait(fetch=False)
enc.wait(fetch=False)
enc.wait(fetch=False)
t = (time.time() - t0) / K
print("pipelined: %.2f ms/batch = %.0f images/s" % (t * 1e3, 256 / t))
