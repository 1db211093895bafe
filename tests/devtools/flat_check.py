"""Dev tool: flat and striped pictures (periodic bitstreams: a decoder started in the wrong state can stay on a stable wrong trajectory, so
corrections travel group by group, one launch each) through the GPU entropy stage; prints the number of sync launches."""
import os
os.environ.setdefault("HIPJPEG_ENABLE_TEST_HOOKS", "1")
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle
from nvimagecodec_amd.lowlevel import BatchDecoder

dec = BatchDecoder(0, 8)
for (w, h) in ((1024, 1024), (2048, 2048), (4096, 2048)):
    for sub in ("444", "420", "422"):
        for kind in ("flat", "stripes"):
            img = np.full((h, w, 3), 137, np.uint8)
            if kind == "stripes":
                img[:, ::16] = 30
            j = oracle.encode(img, sub, 90)
            outs, st = dec.decode([j], gpu_huffman=True, check=False)
            torch.cuda.synchronize()
            ok = st[0] == 0 and np.array_equal(outs[0].cpu().numpy(), oracle.decode(j))
            s = dec.stats()
            print(w, h, sub, kind, "bytes", len(j), "status", st[0], "ok", ok, "gpu images", s["gpu_entropy_images"], "sync launches", s["sync_launches"], "host fallbacks", dec.host_fallbacks())

import time
for (w, h) in ((2048, 2048), (4096, 2048)):
    img = np.full((h, w, 3), 137, np.uint8)
    img[:, ::16] = 30
    j = oracle.encode(img, "420", 90)
    for gh in (True, False):
        dec.decode([j], gpu_huffman=gh, check=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dec.decode([j], gpu_huffman=gh, check=False)
        torch.cuda.synchronize()
        print(w, h, "stripes 420 gpu_huffman", gh, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), "launches", dec.stats()["sync_launches"], "host fallbacks", dec.host_fallbacks())
