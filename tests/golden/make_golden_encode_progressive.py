#!/usr/bin/env python3
"""Progressive (SOF2) encode goldens: the REAL libjpeg-turbo's progressive files (via Pillow, `progressive=True`, i.e.
jpeg_simple_progression + per-scan optimal Huffman tables) for the input pixels already under tests/golden/encode/, plus a few
larger / restart-interval cases with inputs of their own.

Dev-container only (needs Pillow built against libjpeg-turbo).  Outputs (all data, no code):
  encode_prog/<name>.jpg          libjpeg-turbo's progressive encoding
  encode_prog/<name>.rgb          input pixels, only for the cases that are not in encode/
  manifest_encode_prog.json       parameters + sha256
Run:  python tests/golden/make_golden_encode_progressive.py
"""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import pil_encode, sha  # noqa: E402
from nvimagecodec_amd.synth import synth_image  # noqa: E402


def main():
    assert features.check_feature("libjpeg_turbo"), "Pillow must be built against libjpeg-turbo"
    out = os.path.join(HERE, "encode_prog")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(HERE, "manifest.json")) as f:
        base = json.load(f)
    entries = []
    for e in base["encode"]:
        rgb = np.fromfile(os.path.join(HERE, "encode", e["name"] + ".rgb"), dtype=np.uint8).reshape(e["height"], e["width"], 3)
        jpeg = pil_encode(rgb, e["quality"], e["sub"], True)
        with open(os.path.join(out, e["name"] + "_prog.jpg"), "wb") as f:
            f.write(jpeg)
        entries.append(dict(name=e["name"] + "_prog", input="encode/" + e["name"] + ".rgb", width=e["width"], height=e["height"], sub=e["sub"],
                            quality=e["quality"], restart=0, rgb_sha256=e["rgb_sha256"], jpeg_sha256=hashlib.sha256(jpeg).hexdigest()))
    # larger pictures (long EOB runs, many correction bits per run) and restart intervals
    extra = [(320, 200, "420", 90, 0, 51), (320, 200, "444", 30, 0, 52), (200, 150, "422", 98, 0, 53), (130, 70, "gray", 75, 0, 54),
             (130, 70, "420", 90, 7, 55), (96, 64, "444", 60, 1, 56), (96, 64, "gray", 90, 5, 57)]
    for (w, h, sub, q, rst, seed) in extra:
        img = synth_image(w, h, seed=seed)
        src = img if sub != "gray" else np.repeat(np.asarray(Image.fromarray(img).convert("L"))[:, :, None], 3, axis=2)
        kw = dict(restart_marker_blocks=rst) if rst else {}
        jpeg = pil_encode(img, q, sub, True, **kw)
        name = f"p{w}x{h}_{sub}_q{q}_rst{rst}_prog"
        with open(os.path.join(out, name + ".rgb"), "wb") as f:
            f.write(np.ascontiguousarray(src).tobytes())
        with open(os.path.join(out, name + ".jpg"), "wb") as f:
            f.write(jpeg)
        entries.append(dict(name=name, input="encode_prog/" + name + ".rgb", width=w, height=h, sub=sub, quality=q, restart=rst,
                            rgb_sha256=sha(src), jpeg_sha256=hashlib.sha256(jpeg).hexdigest()))
    with open(os.path.join(HERE, "manifest_encode_prog.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_golden_encode_progressive.py", libjpeg_turbo=features.version_feature("libjpeg_turbo"),
                       encode_progressive=entries), f, indent=1)
    print(len(entries), "progressive encode vectors")


if __name__ == "__main__":
    main()
