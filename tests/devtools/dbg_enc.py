import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from conftest import GOLDEN, load_encode_case
from nvimagecodec_amd.lowlevel import BatchEncoder
M = json.load(open(os.path.join(GOLDEN, "manifest.json")))
cases = [(e, *load_encode_case(e)) for e in M["encode"] if e["sub"] != "gray"]
enc = BatchEncoder(0, 4, gpu_huffman=True)
for n in (1, 2, len(cases)):
    sub = cases[:n]
    dev = [torch.from_numpy(np.ascontiguousarray(c[1])).cuda() for c in sub]
    out = enc.encode(dev, subsampling=[c[0]["sub"] for c in sub], quality=[c[0]["quality"] for c in sub])
    for i, (c, s) in enumerate(zip(sub, out)):
        ref = oracle.encode(c[1], c[0]["sub"], c[0]["quality"])
        if s != ref:
            d = [k for k in range(min(len(s), len(ref))) if s[k] != ref[k]]
            print("batch", n, "image", i, c[0]["name"], "len", len(s), len(ref), "ndiff", len(d), "first", d[:6], [(hex(s[k]), hex(ref[k])) for k in d[:6]], "scan starts", ref.rfind(b"\xff\xda") + 14)
print("done")
