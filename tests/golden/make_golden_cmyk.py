#!/usr/bin/env python3
"""Golden vectors for four-component (CMYK / YCCK) JPEGs, from the REAL libjpeg-turbo (via Pillow).  Dev-container only.

The reference's CPU path sets out_color_space = JCS_CMYK for such frames and converts to RGB itself
(extensions/libjpeg_turbo/jpeg_mem.cpp:168-172, 292-337).  What libjpeg-turbo hands it -- the CMYK samples -- is what these
vectors pin: Pillow reads every 4-layer JPEG with raw mode "CMYK;I" (inverted), so  libjpeg's output = 255 - Pillow's pixels.
Inputs: Pillow-encoded CMYK files (Adobe marker, transform 0), the same with the transform byte patched to 2 (YCCK: libjpeg
then converts YCC->RGB and complements), and with the Adobe segment removed (plain CMYK without marker).
Outputs:  cmyk/<name>.jpg, cmyk/<name>.cmyk (H x W x 4 bytes), manifest_cmyk.json."""
import hashlib
import io
import json
import os
import sys

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from nvimagecodec_amd.synth import synth_image  # noqa: E402


def main():
    assert features.check_feature("libjpeg_turbo")
    out = os.path.join(HERE, "cmyk")
    os.makedirs(out, exist_ok=True)
    entries = []
    for (w, h, seed) in ((50, 37, 1), (17, 13, 2), (64, 48, 3), (130, 70, 4)):
        rgb = synth_image(w, h, seed=7000 + seed)
        k = (255 - rgb.max(axis=2)).astype(np.uint8)
        cmyk = np.dstack([255 - rgb[:, :, 0], 255 - rgb[:, :, 1], 255 - rgb[:, :, 2], k]).astype(np.uint8)
        for sub in (0, 2):
            for prog in (False, True):
                if prog and w != 64:
                    continue
                b = io.BytesIO()
                Image.fromarray(cmyk, "CMYK").save(b, "JPEG", quality=90, subsampling=sub, progressive=prog)
                base = b.getvalue()
                i = base.find(b"Adobe")
                assert i > 0 and base[i + 11] == 0
                s = base.find(b"\xff\xee")
                seg = (base[s + 2] << 8) | base[s + 3]
                variants = {"adobe0": base, "adobe2": base[: i + 11] + b"\x02" + base[i + 12:], "plain": base[:s] + base[s + 2 + seg:]}
                for kind, jpeg in variants.items():
                    name = f"k{w}x{h}_{'s211' if sub else 's111'}_{'prog' if prog else 'base'}_{kind}"
                    lib_out = 255 - np.asarray(Image.open(io.BytesIO(jpeg)))   # what libjpeg-turbo's JCS_CMYK output holds
                    assert lib_out.shape == (h, w, 4)
                    with open(os.path.join(out, name + ".jpg"), "wb") as f:
                        f.write(jpeg)
                    with open(os.path.join(out, name + ".cmyk"), "wb") as f:
                        f.write(np.ascontiguousarray(lib_out).tobytes())
                    entries.append(dict(name=name, width=w, height=h, kind=kind, progressive=prog, subsampled=bool(sub),
                                        cmyk_sha256=hashlib.sha256(np.ascontiguousarray(lib_out).tobytes()).hexdigest()))
    with open(os.path.join(HERE, "manifest_cmyk.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden_cmyk.py", "pillow": Image.__version__, "libjpeg_turbo": features.version("libjpeg_turbo"),
                   "cmyk": entries}, f, indent=1)
    print(len(entries), "CMYK vectors")


if __name__ == "__main__":
    main()
