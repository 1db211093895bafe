"""Ad-hoc timing of the nvImageCodec-API route (Python surface -> host harness -> plugin).  Dev tool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, oracle
from nvimagecodec_amd import api
src, _ = bench.make_inputs()
B = 256
jpegs = [src[i % len(src)] for i in range(B)]
for opts in ("", "hipjpeg_decoder:pipeline_chunks=1", "hipjpeg_decoder:pipeline_chunks=2", "hipjpeg_decoder:gpu_huffman=0"):
    with api.Decoder(max_num_cpu_threads=bench.usable_cpus(), options=opts) as dec:
        imgs = dec.decode(jpegs)
        torch.cuda.synchronize()
        for rep in range(3):
            t0 = time.time()
            imgs = dec.decode(jpegs)
            torch.cuda.synchronize()
            t1 = time.time()
            print("options=%r: %.1f ms/batch %.0f img/s" % (opts, (t1 - t0) * 1e3, B / (t1 - t0)), flush=True)
        print("parity", np.array_equal(np.asarray(imgs[1].cpu()._array), oracle.decode(jpegs[1])))
