import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, oracle
from conftest import GOLDEN
from nvimagecodec_amd.lowlevel import BatchDecoder
M = json.load(open(os.path.join(GOLDEN, "manifest_cmyk.json")))["cmyk"]
e = M[0]
j = open(os.path.join(GOLDEN, "cmyk", e["name"] + ".jpg"), "rb").read()
dec = BatchDecoder(0, 2)
outs, st = dec.decode([j], fmt="y")
torch.cuda.synchronize()
got = outs[0].cpu().numpy(); ref = oracle.decode(j, oracle.FMT_GRAY); rgb = oracle.decode(j)
d = np.argwhere(got != ref)
print(len(d), [(tuple(x), int(got[tuple(x)]), int(ref[tuple(x)]), rgb[tuple(x)].tolist()) for x in d[:8]])
from nvimagecodec_amd import api
with api.Decoder(max_num_cpu_threads=2) as d2:
    os.environ["X"] = "1"
    im = d2.decode(j)
    print("api:", im)
