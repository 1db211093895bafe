"""Profiling driver (dev tool): pipelined submit/wait of 256x1080p batches.  Run under rocprofv3 --kernel-trace --memory-copy-trace."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nvimagecodec_amd.lowlevel import BatchDecoder
src, _ = bench.make_inputs()
B = 256
jpegs = [src[i % len(src)] for i in range(B)]
dec = BatchDecoder(0, bench.usable_cpus())
outs = [dec.allocate_outputs(jpegs), dec.allocate_outputs(jpegs), dec.allocate_outputs(jpegs)]
for k in range(3):  # every page sizes its arenas on first use
    dec.submit(jpegs, outs[k])
for k in range(3):
    dec.wait()
torch.cuda.synchronize(); t0 = time.time()
K = 9
for i in range(K):
    dec.submit(jpegs, outs[i % 3])
    if i > 1:
        dec.wait()
dec.wait(); dec.wait()
torch.cuda.synchronize()
print("pipelined: %.2f ms/batch" % ((time.time() - t0) / K * 1e3))
