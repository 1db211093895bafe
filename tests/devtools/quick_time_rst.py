"""Ad-hoc: GPU entropy stage on streams with restart intervals (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from nvimagecodec_amd.lowlevel import BatchDecoder
from nvimagecodec_amd.synth import synth_image
B = 256
for interval in (0, 120, 8):
    src = [oracle.encode(synth_image(1920, 1080, seed=s), "420", 90, restart_interval=interval) for s in range(4)]
    jpegs = [src[i % 4] for i in range(B)]
    dec = BatchDecoder(0, num_threads=0)
    outs = dec.allocate_outputs(jpegs)
    for rep in range(3):
        dec.host_stage(jpegs, outs, gpu_huffman=True); dec.transfer(); torch.cuda.synchronize()
        t0 = time.time(); dec.device_stage(which=3); torch.cuda.synchronize(); t1 = time.time()
    dec.device_stage()
    torch.cuda.synchronize()
    print("restart interval %d: entropy stage %.2f ms, parity %s" % (interval, (t1 - t0) * 1e3, np.array_equal(outs[1].cpu().numpy(), oracle.decode(src[1]))), flush=True)
    dec.close()
