"""Dev tool: where the caller's thread spends its time in the pipelined end-to-end decode (configs[1], three batches in flight)."""
import gc
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nvimagecodec_amd.lowlevel import BatchDecoder
src, _ = bench.make_inputs()
jpegs = [src[i % len(src)] for i in range(256)]
dec = BatchDecoder(0, bench.usable_cpus())
outs = [dec.allocate_outputs(jpegs) for _ in range(3)]
for k in range(3):  # every page sizes its arenas on first use
    dec.submit(jpegs, outs[k])
for k in range(3):
    dec.wait()
torch.cuda.synchronize()
K = int(os.environ.get('E2E_STEPS', '60'))
ts, tw = [], []
if os.environ.get('E2E_NO_GC'):
    gc.disable()
t0 = time.perf_counter()
for i in range(K):
    a = time.perf_counter()
    dec.submit(jpegs, outs[i % 3])
    b = time.perf_counter()
    if i > 1:
        dec.wait()
    c = time.perf_counter()
    ts.append(b - a); tw.append(c - b)
dec.wait(); dec.wait()
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / K
it = sorted(x + y for x, y in zip(ts, tw))
print("iteration ms: min %.2f  p25 %.2f  median %.2f  p75 %.2f  max %.2f" % (it[0] * 1e3, it[K // 4] * 1e3, it[K // 2] * 1e3, it[3 * K // 4] * 1e3, it[-1] * 1e3))
print("pipelined: %.2f ms/batch = %.0f images/s; submit %.2f ms (median), wait %.2f ms (median)" % (
    t * 1e3, 256 / t, sorted(ts)[K // 2] * 1e3, sorted(tw)[K // 2] * 1e3))
