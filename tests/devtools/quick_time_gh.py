"""Ad-hoc timing of the decode phases with the GPU entropy stage (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from nvimagecodec_amd.lowlevel import BatchDecoder
from nvimagecodec_amd.synth import synth_image

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nsrc = 4
src = [oracle.encode(synth_image(1920, 1080, seed=s), "420", 90) for s in range(nsrc)]
jpegs = [src[i % nsrc] for i in range(B)]
dec = BatchDecoder(0, num_threads=0)
outs = dec.allocate_outputs(jpegs)
for rep in range(3):
    t0 = time.time(); dec.host_stage(jpegs, outs, gpu_huffman=True); t1 = time.time()
    dec.transfer(); torch.cuda.synchronize(); t2 = time.time()
    dec.device_stage(which=3); torch.cuda.synchronize(); t3 = time.time()
    dec.device_stage(); torch.cuda.synchronize(); t4 = time.time()
    print("host stage %.1f ms (%.0f img/s)  h2d %.2f ms  entropy %.2f ms (%.0f img/s)  idct+color %.2f ms" % ((t1 - t0) * 1e3, B / (t1 - t0), (t2 - t1) * 1e3, (t3 - t2) * 1e3, B / (t3 - t2), (t4 - t3) * 1e3), flush=True)
    print(dec.stats(), flush=True)
for rep in range(3):
    t0 = time.time()
    for _ in range(3):
        dec.decode(jpegs, outs=outs, gpu_huffman=True)
    torch.cuda.synchronize(); t1 = time.time()
    print("end-to-end decode(gpu_huffman): %.1f ms/batch %.0f img/s" % ((t1 - t0) / 3 * 1e3, 3 * B / (t1 - t0)), flush=True)
# pipelined: two batches in flight, two output sets
outs2 = dec.allocate_outputs(jpegs)
outs3 = dec.allocate_outputs(jpegs)
ring = [outs, outs2, outs3]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    K = 30
    ts = tw = 0.0
    for i in range(K):
        a = time.time(); dec.submit(jpegs, ring[i % 3]); b = time.time(); ts += b - a
        if i > 1:
            dec.wait(); tw += time.time() - b
    dec.wait(); dec.wait()
    print("   host: submit %.2f ms, wait %.2f ms per batch" % (ts / K * 1e3, tw / (K - 2) * 1e3))
    torch.cuda.synchronize(); t1 = time.time()
    print("pipelined submit/wait: %.1f ms/batch %.0f img/s" % ((t1 - t0) / K * 1e3, K * B / (t1 - t0)), flush=True)
ref = oracle.decode(src[1]); print("parity2", np.array_equal(outs2[1].cpu().numpy(), ref)); print("parity", np.array_equal(outs[1].cpu().numpy(), ref))
