"""Dev tool: configs[1] from bitstreams resident in HBM with 1 / 2 / 3 decoder instances taking turns on streams of their own -- how much of a
batch's idle phases (the tail kernels of the entropy stage) the next batch's kernels fill.  usage (GPU box): python tools/steps_in_flight.py [steps [only this many in flight]]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nvimagecodec_amd.lowlevel import BatchDecoder

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sources, _ = bench.make_inputs()
jpegs = [sources[i % len(sources)] for i in range(bench.BATCH)]
only = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for n in ((only,) if only else (1, 2, 3)):
    decs = [BatchDecoder(device=0, num_threads=bench.usable_cpus()) for _ in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    outs = [d.allocate_outputs(jpegs, "rgb") for d in decs]
    for d, o in zip(decs, outs):
        d.host_stage(jpegs, o, "rgb", fancy=True, gpu_huffman=True)
        d.transfer()
    torch.cuda.synchronize()

    stagger = os.environ.get("STAGGER", "1") != "0"
    last_entropy = [None]

    def step(k):
        d, s = decs[k % n], streams[k % n]
        # the entropy stages take turns (each waits for the one before it, whichever stream that ran on): a batch's pixel kernels -- bound by
        # HBM -- then run beside the NEXT batch's entropy stage -- bound by instruction issue and latency -- instead of beside its pixel kernels
        if stagger and n > 1 and last_entropy[0] is not None:
            s.wait_event(last_entropy[0])
        d.device_stage(stream=s, which=6)
        if stagger and n > 1:
            last_entropy[0] = torch.cuda.Event()
            last_entropy[0].record(s)
        d.device_stage(stream=s, which=0)
        d.device_stage(stream=s, which=1)

    for k in range(3 * n):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    ok = all(all(s == 0 for s in d.statuses(bench.BATCH)) for d in decs)
    print("%d in flight: %.3f ms per step, %.0f images/s, statuses ok %s" % (n, t * 1e3, bench.BATCH / t, ok))
    for d in decs:
        d.close()
