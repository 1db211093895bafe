"""Ad-hoc timing of the three decode phases (dev tool; bench.py is the contract benchmark)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from nvimagecodec_amd.lowlevel import BatchDecoder
from nvimagecodec_amd.synth import synth_image

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nsrc = 4
t = time.time()
src = [oracle.encode(synth_image(1920, 1080, seed=s), "420", 90) for s in range(nsrc)]
print("inputs: %.1fs, %d bytes avg" % (time.time() - t, sum(map(len, src)) / nsrc), flush=True)
jpegs = [src[i % nsrc] for i in range(B)]
dec = BatchDecoder(0, num_threads=0)
outs = dec.allocate_outputs(jpegs)
for rep in range(3):
    t0 = time.time(); dec.host_stage(jpegs, outs); t1 = time.time()
    dec.transfer(); torch.cuda.synchronize(); t2 = time.time()
    print("host stage %.1f ms (%.1f img/s)  h2d %.1f ms (%.1f GB/s)" % ((t1 - t0) * 1e3, B / (t1 - t0), (t2 - t1) * 1e3, dec.stats()["coef_bytes"] / (t2 - t1) / 1e9), flush=True)
st = dec.stats(); print(st)
for rep in range(3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); 
    for _ in range(5): dec.device_stage()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    byts = st["coef_bytes"] + st["output_bytes"]
    print("device stage %.3f ms/batch  %.0f img/s  %.1f GB/s algorithmic" % (ms, B / ms * 1e3, byts / ms / 1e6), flush=True)
ref = oracle.decode(src[1]); print("parity", np.array_equal(outs[1].cpu().numpy(), ref))
