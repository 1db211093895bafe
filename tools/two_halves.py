"""Experiment (dev tool, GPU box): the resident-bitstream decode step of bench.py for 256 x 1080p as ONE batch on one stream against the
same pictures as TWO half batches (two decoder handles, two streams) running side by side, and as FOUR quarters.  Does the latency-bound
part of one slice's entropy stage (tail / ripple kernels) hide behind the other slices' throughput-bound kernels?
usage: python tools/two_halves.py [steps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nvimagecodec_amd.lowlevel import BatchDecoder

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sources, _ = bench.make_inputs()
jpegs = [sources[i % len(sources)] for i in range(bench.BATCH)]


def run(nslices, stagger):
    n = bench.BATCH // nslices
    decs = [BatchDecoder(0, 4) for _ in range(nslices)]
    streams = [torch.cuda.Stream() for _ in range(nslices)]
    outs = []
    for d, k in zip(decs, range(nslices)):
        part = jpegs[k * n:(k + 1) * n]
        o = d.allocate_outputs(part, "rgb")
        d.host_stage(part, o, "rgb", fancy=True, gpu_huffman=True)
        d.transfer()
        outs.append(o)
    torch.cuda.synchronize()

    def step():
        evs = []
        for k, (d, s) in enumerate(zip(decs, streams)):
            if stagger and evs:
                s.wait_event(evs[-1])
            d.device_stage(stream=s, which=6)
            if stagger:
                e = torch.cuda.Event()
                e.record(s)
                evs.append(e)
        for d, s in zip(decs, streams):
            d.device_stage(stream=s, which=0)
            d.device_stage(stream=s, which=1)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    for d in decs:
        assert all(s == 0 for s in d.statuses(n))
        d.close()
    print("%d slice(s)%s: %.3f ms per 256 images, %.0f images/s" % (nslices, " staggered" if stagger else "", t * 1e3, bench.BATCH / t), flush=True)


run(1, False)
run(2, False)
run(4, False)
run(8, False)
