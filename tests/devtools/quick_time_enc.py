"""Ad-hoc timing of the encode phases (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from nvimagecodec_amd.lowlevel import BatchEncoder
from nvimagecodec_amd.synth import synth_image

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
src = [torch.from_numpy(synth_image(1920, 1080, seed=s)).cuda() for s in range(4)]
imgs = [src[i % 4] for i in range(B)]
enc = BatchEncoder(0, num_threads=0)
for mode in (True, False):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        enc.device_stage(imgs, "420", 90, "rgb"); torch.cuda.synchronize(); t1 = time.time()
        enc.host_stage(gpu_huffman=mode); t2 = time.time()
        print("gpu_huffman=%s: device stage %.2f ms, entropy stage %.2f ms -> %.0f img/s end to end" % (mode, (t1 - t0) * 1e3, (t2 - t1) * 1e3, B / (t2 - t0)), flush=True)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    K = 16
    for i in range(K):
        enc.submit(imgs, "420", 90, "rgb", gpu_huffman=True)
        if i > 1:
            enc.wait(fetch=False)
    enc.wait(fetch=False)
    enc.wait(fetch=False)
    t1 = time.time()
    print("pipelined submit/wait: %.2f ms/batch %.0f img/s" % ((t1 - t0) / K * 1e3, K * B / (t1 - t0)), flush=True)
s = enc.bitstreams()
print("bytes/img", len(s[0]), "identical to oracle:", s[1] == oracle.encode(synth_image(1920, 1080, seed=1), "420", 90))
