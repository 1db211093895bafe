"""Dev tool: the full decode step of the bench (GPU entropy stage + K1 + K2 on 256 x 1080p 4:2:0, bitstreams resident in HBM) over JPEG
qualities -- which pass-1 arithmetic the host gives the batch, ms per stage, images/s.
usage: python tools/quality_sweep.py [qualities...]"""
import os
os.environ.setdefault("HIPJPEG_ENABLE_TEST_HOOKS", "1")  # kernel_flavours / host_fallbacks are test hooks of the library
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nvimagecodec_amd.lowlevel import BatchDecoder
from nvimagecodec_amd.synth import synth_image

quals = [int(a) for a in sys.argv[1:]] or [30, 50, 70, 75, 85, 90, 95]
imgs = [synth_image(bench.WIDTH, bench.HEIGHT, seed=1234 + s) for s in range(bench.NUM_SOURCES)]
dec = BatchDecoder(0, bench.usable_cpus())
PLANE = ("24-bit", "32-bit", "packed")
LUMA = tuple("%s %s" % (a, l) for a in ("24-bit", "32-bit", "packed") for l in ("generic", "common", "planar"))
for q in quals:
    src = [bench._pil_encode(im, q, "420") for im in imgs]
    jpegs = [src[i % len(src)] for i in range(256)]
    outs = dec.allocate_outputs(jpegs, "rgb")
    dec.host_stage(jpegs, outs, "rgb", fancy=True, gpu_huffman=True)
    dec.transfer()
    plane, luma = dec.kernel_flavours()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    tot = [0.0, 0.0, 0.0]
    K = 20
    for k in range(10 + K):
        ev[0].record(); dec.device_stage(which=6)
        ev[1].record(); dec.device_stage(which=0)
        ev[2].record(); dec.device_stage(which=1)
        ev[3].record()
        torch.cuda.synchronize()
        if k >= 10:
            for i in range(3):
                tot[i] += ev[i].elapsed_time(ev[i + 1]) / K
    assert all(s == 0 for s in dec.statuses(256))
    step = sum(tot)
    print("q%-3d %6.0f KB/image  K1 %-7s K2 %-15s entropy %.3f ms  K1 %.3f  K2 %.3f  step %.3f ms = %6.0f images/s" % (
        q, sum(len(j) for j in jpegs) / 256 / 1024, PLANE[max(range(3), key=lambda e: plane[e])], LUMA[max(range(9), key=lambda e: luma[e])],
        tot[0], tot[1], tot[2], step, 256 / step * 1e3), flush=True)
    del outs
