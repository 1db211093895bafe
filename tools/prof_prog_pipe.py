"""Profiling driver (dev tool): BASELINE configs[4] (128 x 1080p progressive 4:4:4 -> planar RGB) through Submit/Wait with three
batches in flight.  Run under `rocprofv3 --kernel-trace`; tools/prog_pipe_trace.sh prints the timeline of the entropy kernels."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nvimagecodec_amd.lowlevel import BatchDecoder
from nvimagecodec_amd.synth import synth_image

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 6
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prog = [bench._pil_encode(synth_image(bench.WIDTH, bench.HEIGHT, seed=900 + k), 90, "444", progressive=True) for k in range(4)]
batch = [prog[i % 4] for i in range(128)]
dec = BatchDecoder(0, bench.usable_cpus())
dec.set_pipeline_depth(depth)
ring = [dec.allocate_outputs(batch, "rgb_planar") for _ in range(depth)]
for k in range(depth):  # every page sizes its arenas on first use
    dec.submit(batch, ring[k], fmt="rgb_planar")
for k in range(depth):
    dec.wait()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(nb):
    ts = time.perf_counter()
    dec.submit(batch, ring[i % depth], fmt="rgb_planar")
    tm = time.perf_counter()
    if i >= depth - 1:
        dec.wait()
    print("batch %d: submit %.1f ms, wait %.1f ms" % (i, (tm - ts) * 1e3, (time.perf_counter() - tm) * 1e3))
for _ in range(depth - 1):
    dec.wait()
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / nb
print("pipelined: %.1f ms per batch, %.0f images/s" % (t * 1e3, 128 / t))
