"""Dev tool: K1 + K2 on 256 pictures whose interleaved RGB rows are NOT 16-byte aligned (1918 x 1080, tight pitch 5,754 bytes) -- the
output configuration that went to the generic luma kernel until round 2's unaligned staged stores; ms per batch and the flavour taken."""
import os
os.environ.setdefault("HIPJPEG_ENABLE_TEST_HOOKS", "1")  # kernel_flavours / host_fallbacks are test hooks of the library
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nvimagecodec_amd.lowlevel import BatchDecoder
from nvimagecodec_amd.synth import synth_image

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1918
src = [bench._pil_encode(synth_image(W, 1080, seed=77 + s), 90, "420") for s in range(8)]
jpegs = [src[i % 8] for i in range(256)]
dec = BatchDecoder(0, bench.usable_cpus())
outs = dec.allocate_outputs(jpegs, "rgb")
dec.host_stage(jpegs, outs, "rgb", fancy=True, gpu_huffman=True)
dec.transfer()
dec.device_stage(which=3)
print("flavours (plane, luma):", dec.kernel_flavours(), "row pitch", outs[0].stride(0))
for _ in range(20):
    dec.device_stage(which=0); dec.device_stage(which=1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    dec.device_stage(which=0); dec.device_stage(which=1)
e1.record(); torch.cuda.synchronize()
print("K1 + K2, 256 x %dx1080: %.4f ms per batch" % (W, e0.elapsed_time(e1) / 20))
