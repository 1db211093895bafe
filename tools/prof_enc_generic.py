"""Dev tool: 256 x 1080p interleaved RGB at a base address that is not a multiple of 8 through the encode device stage, ms per batch by
HIP events -- the pair kernel by default (it takes any alignment), the one-lane-per-block kernel with HIPJPEG_ENCODE_ONE_LANE_KERNEL=1."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvimagecodec_amd.lowlevel import BatchEncoder
from nvimagecodec_amd.synth import synth_image

W, H = 1920, 1080
src = [torch.from_numpy(synth_image(W, H, seed=s)).cuda() for s in range(4)]
imgs = []
for i in range(256):
    buf = torch.zeros(H * W * 3 + 64, dtype=torch.uint8, device="cuda")
    a = torch.as_strided(buf, (H, W, 3), (W * 3, 3, 1), storage_offset=3)
    a.copy_(src[i % 4])
    imgs.append(a)
enc = BatchEncoder(device=0, num_threads=8)
enc.device_stage(imgs, "420", 90, "rgb")
for _ in range(10):
    enc.relaunch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    enc.relaunch()
e1.record()
torch.cuda.synchronize()
print("%s, 256 x 1080p interleaved RGB at an odd base, 4:2:0: %.3f ms per batch" % (
    "one-lane-per-block kernel" if os.environ.get("HIPJPEG_ENCODE_ONE_LANE_KERNEL") else "pair kernel", e0.elapsed_time(e1) / 10))

# planar RGB (CHW) input
planes = [s.permute(2, 0, 1).contiguous() for s in src]
pl = [planes[i % 4].clone() for i in range(256)]
enc.device_stage(pl, "420", 90, "rgb_planar")
for _ in range(10):
    enc.relaunch()
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    enc.relaunch()
e1.record()
torch.cuda.synchronize()
print("%s, 256 x 1080p planar RGB, 4:2:0: %.3f ms per batch" % (
    "one-lane-per-block kernel" if os.environ.get("HIPJPEG_ENCODE_ONE_LANE_KERNEL") else "pair kernel", e0.elapsed_time(e1) / 10))
