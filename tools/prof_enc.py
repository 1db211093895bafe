"""Profiling driver (dev tool): encode device stage + GPU entropy coder of one 256 x 1080p batch, N times.  Run under rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvimagecodec_amd.lowlevel import BatchEncoder
from nvimagecodec_amd.synth import synth_image
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sub = sys.argv[2] if len(sys.argv) > 2 else "420"
src = [torch.from_numpy(synth_image(1920, 1080, seed=s)).cuda() for s in range(4)]
imgs = [src[i % 4] for i in range(256)]
enc = BatchEncoder(0, num_threads=0)
for _ in range(n):
    enc.device_stage(imgs, sub, 90, "rgb")
    torch.cuda.synchronize()
    enc.host_stage(gpu_huffman=True)
print("done")
