"""Ad-hoc timing of the encode API route (Python surface -> host harness -> encoder plugin).  Dev tool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from nvimagecodec_amd import api
from nvimagecodec_amd.synth import synth_image
src = [torch.from_numpy(synth_image(1920, 1080, seed=s)).cuda() for s in range(4)]
imgs = [api.as_image(src[i % 4]) for i in range(256)]
params = api.EncodeParams(quality=90, chroma_subsampling=api.ChromaSubsampling.CSS_420)
with api.Encoder(max_num_cpu_threads=16) as enc:
    for rep in range(4):
        t0 = time.time()
        out = enc.encode(imgs, "jpeg", params)
        t1 = time.time()
        print("encode API: %.1f ms/batch %.0f img/s" % ((t1 - t0) * 1e3, 256 / (t1 - t0)), flush=True)
    print("identical to oracle:", bytes(out[1]) == oracle.encode(synth_image(1920, 1080, seed=1), "420", 90))
