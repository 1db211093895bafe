"""Profiling driver (dev tool): pipelined encode submit/wait of 256 x 1080p batches.  Run under rocprofv3 --kernel-trace --memory-copy-trace."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvimagecodec_amd.lowlevel import BatchEncoder
from nvimagecodec_amd.synth import synth_image
src = [torch.from_numpy(synth_image(1920, 1080, seed=s)).cuda() for s in range(4)]
imgs = [src[i % 4] for i in range(256)]
enc = BatchEncoder(0, num_threads=0)
enc.submit(imgs, "420", 90, "rgb", gpu_huffman=True); enc.wait(fetch=False)
torch.cuda.synchronize(); t0 = time.time()
K = 10
for i in range(K):
    enc.submit(imgs, "420", 90, "rgb", gpu_huffman=True)
    if i > 1:
        enc.wait(fetch=False)
enc.wait(fetch=False)
enc.wait(fetch=False)
print("pipelined: %.2f ms/batch" % ((time.time() - t0) / K * 1e3))
