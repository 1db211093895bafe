"""Dev tool: pipelined end-to-end decode (configs[1]) with N decoder instances in one process, each driven by its own thread and with
three batches in flight (ctypes releases the GIL inside the C calls) -- how much of the gap between one instance and the GPU's
rate is the single caller thread."""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
import torch
import bench
from nvimagecodec_amd.lowlevel import BatchDecoder

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 24
src, _ = bench.make_inputs()
jpegs = [src[i % len(src)] for i in range(256)]
threads_each = max(1, bench.usable_cpus() // n_inst)
decs = [BatchDecoder(0, threads_each) for _ in range(n_inst)]
rings = [[d.allocate_outputs(jpegs) for _ in range(3)] for d in decs]
streams = [torch.cuda.Stream() for _ in range(n_inst)]
for d, r, s in zip(decs, rings, streams):
    for k in range(3):  # every page sizes its arenas on first use
        d.submit(jpegs, r[k], stream=s)
    for k in range(3):
        d.wait()
torch.cuda.synchronize()


def run(d, ring, s):
    for i in range(K):
        d.submit(jpegs, ring[i % 3], stream=s)
        if i > 1:
            d.wait()
    d.wait()
    d.wait()


t0 = time.perf_counter()
ths = [threading.Thread(target=run, args=(d, r, s)) for d, r, s in zip(decs, rings, streams)]
for t in ths:
    t.start()
for t in ths:
    t.join()
torch.cuda.synchronize()
t = time.perf_counter() - t0
print("%d instance(s), %d host threads each: %.2f ms per batch, %.0f images/s" % (n_inst, threads_each, t / (K * n_inst) * 1e3, 256 * K * n_inst / t))
