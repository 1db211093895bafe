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
    if os.environ.get("HIPJPEG_FUSED_DECODE"):
        # the pixel kernels decode the blocks themselves: damaged streams are flagged THERE, the host decoder takes them over and the
        # plain kernels finish them (DecodeBatch::launch_taken_pixels) -- same statuses and pixels as with Huffman decoding on the host
        small = [j for j in jpegs if len(j) > 3000][:12]
        outs, st = dec.decode(small, fmt="rgb", gpu_huffman=True)
        torch.cuda.synchronize()
        assert dec.fused_units() > 0
        import random
        rng = random.Random(7)
        damaged = []
        for j in small:
            b = bytearray(j)
            sos = bytes(b).rfind(b"\xff\xda")
            for _ in range(3):
                k = rng.randrange(sos + 20, len(b) - 2)
                b[k] ^= 1 << rng.randrange(8)
                if b[k] == 0xFF:
                    b[k] = 0xFE
            damaged.append(bytes(b))
        batch = [x for pair in zip(small, damaged) for x in pair]
        outs_h, st_h = dec.decode(batch, fmt="rgb", gpu_huffman=False, check=False)
        torch.cuda.synchronize()
        ref = [o.cpu().numpy().copy() if o is not None else None for o in outs_h]
        outs_g, st_g = dec.decode(batch, fmt="rgb", gpu_huffman=True, check=False)
        torch.cuda.synchronize()
        assert list(st_g) == list(st_h), (list(st_g), list(st_h))
        for i, (a, b) in enumerate(zip(outs_g, ref)):
            if st_h[i] == 0 and b is not None:
                assert np.array_equal(a.cpu().numpy(), b), i
        # found by tests/campaigns/fuzz_damage.py under this switch (round 3): damage that sits in a block of the MCU padding, which the FUSED
        # luma kernel did not decode at first -- the stream's check must cover every block, visible or not
        pad = open(os.path.join(GOLDEN, "damaged", "padding_block_damage.jpg"), "rb").read()
        trio = [small[0], pad, small[1]]
        _, st_h3 = dec.decode(trio, fmt="rgb", gpu_huffman=False, check=False)
        _, st_g3 = dec.decode(trio, fmt="rgb", gpu_huffman=True, check=False)
        torch.cuda.synchronize()
        assert list(st_g3) == list(st_h3) and st_h3[1] != 0, (list(st_g3), list(st_h3))
        print("fused: damaged neighbours ok,", sum(1 for s in st_h if s), "refused,", dec.host_fallbacks(), "taken over by the host decoder")
    print("goldens ok", len(jpegs), "flavours", plane, luma)


if __name__ == "__main__":
    main()
