"""Dev tool: the device stage of configs[1] as two launches (K1 over the batch, then K2) against K1 / K2
alternating over slices of HIPJPEG_PIXEL_CHUNK images (`which` 7), so that a slice's chroma planes (1 MB per 1080p image) are still in
the Infinity Cache when K2 reads them.  Prints ms per 256-image step for both and checks that the pictures are the same.
usage: HIPJPEG_PIXEL_CHUNK=16 python tools/chunk_k12.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nvimagecodec_amd.lowlevel import BatchDecoder

src, _ = bench.make_inputs()
jpegs = [src[i % len(src)] for i in range(256)]
dec = BatchDecoder(0, bench.usable_cpus())
outs = dec.allocate_outputs(jpegs, "rgb")
dec.host_stage(jpegs, outs, "rgb", fancy=True, gpu_huffman=True)
dec.transfer()
dec.device_stage(which=3)
torch.cuda.synchronize()


def timed(fn, reps=40):
    for _ in range(40):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def two_launches():
    dec.device_stage(which=0)
    dec.device_stage(which=1)


t_two = timed(two_launches)
ref = [o.clone() for o in outs]
for o in outs:
    o.zero_()
t_chunk = timed(lambda: dec.device_stage(which=7))
same = all(torch.equal(a, b) for a, b in zip(outs, ref))
print("chunk %s images: two launches %.4f ms, alternating slices %.4f ms per step; same pictures: %s" % (
    os.environ.get("HIPJPEG_PIXEL_CHUNK", "16"), t_two, t_chunk, same))
