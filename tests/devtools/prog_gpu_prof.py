"""Dev tool: one batch of 128 progressive 1080p 4:4:4 images through the GPU entropy stage, twice (put under rocprofv3)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from nvimagecodec_amd.lowlevel import BatchDecoder
from nvimagecodec_amd.synth import synth_image
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sub = sys.argv[2] if len(sys.argv) > 2 else "444"
srcs = [bench._pil_encode(synth_image(1920, 1080, seed=900 + k), 90, sub, progressive=True) for k in range(4)]
batch = [srcs[i % 4] for i in range(n)]
dec = BatchDecoder(0, 8)
outs = dec.allocate_outputs(batch, "rgb_planar")
for rep in range(2):
    t0 = time.perf_counter()
    dec.decode(batch, fmt="rgb_planar", outs=outs, gpu_huffman=True)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print("batch %d 1080p %s progressive: %.1f ms -> %.0f img/s" % (n, sub, t * 1e3, n / t), dec.stats()["gpu_entropy_images"])
