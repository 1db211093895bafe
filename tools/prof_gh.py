"""Profiling driver (dev tool): one 256x1080p batch, GPU entropy stage run N times.  Run under rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nvimagecodec_amd.lowlevel import BatchDecoder
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
src, _ = bench.make_inputs()
jpegs = [src[i % len(src)] for i in range(B)]
dec = BatchDecoder(0, bench.usable_cpus())
outs = dec.allocate_outputs(jpegs)
dec.host_stage(jpegs, outs, gpu_huffman=True)
dec.transfer()
torch.cuda.synchronize()
for _ in range(n):
    dec.device_stage(which=3)
torch.cuda.synchronize()
print("done", dec.stats())
