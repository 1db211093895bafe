"""Test helper (run as a child process): decodes every golden bitstream in one batch through the GPU entropy stage and the host one and
compares with the manifest's hashes -- so that a test can run it under the library's measurement switches, which are read once per
process (HIPJPEG_DEVICE_DESTUFF_COUNT, HIPJPEG_NO_PK16, HIPJPEG_SINGLE_STREAM ...)."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    import numpy as np
    import torch
    from conftest import GOLDEN, load_decode_case
    from nvimagecodec_amd.lowlevel import BatchDecoder
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        entries = json.load(f)["decode"]
    jpegs = [load_decode_case(e)[0] for e in entries]
    dec = BatchDecoder(device=0, num_threads=4)
    for gh in (True, False):
        outs, st = dec.decode(jpegs, fmt="rgb", gpu_huffman=gh)
        torch.cuda.synchronize()
        assert all(s == 0 for s in st), st
        for e, o in zip(entries, outs):
            got = hashlib.sha256(np.ascontiguousarray(o.cpu().numpy()).tobytes()).hexdigest()
            assert got == e["rgb_sha256"], (e["name"], gh)
    plane, luma = dec.kernel_flavours()
    print("goldens ok", len(jpegs), "flavours", plane, luma)


if __name__ == "__main__":
    main()
