#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ with the REAL libjpeg-turbo (via Pillow).

Dev-container only (needs Pillow built against libjpeg-turbo; records the version).  The reference's CPU JPEG
path (extensions/libjpeg_turbo/jpeg_mem.cpp:143-450) calls libjpeg-turbo with JDCT_ISLOW + fancy upsampling +
JCS_RGB; Pillow's JPEG plugin drives the same library with the same settings, so `Image.open(...).convert()`
outputs are that path's pixels.  The reference's own fixtures (resources/ref/jpeg/*.ppm) are not replayable here:
their inputs are git-LFS stubs.

Outputs (all data, no code):
  decode/<name>.jpg           input bitstream (Pillow-encoded, or oracle-encoded for samplings Pillow cannot write)
  decode/<name>.rgb           libjpeg-turbo's decoded interleaved RGB (small cases only)
  encode/<name>.rgb           seeded input pixels
  encode/<name>.jpg           Pillow/libjpeg-turbo's baseline encoding of them (std Huffman tables)
  manifest.json               parameters + sha256 of every decoded image
Run:  python tests/golden/make_golden.py
"""
import hashlib
import io
import json
import os
import sys

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (only its ENCODER is used here, to write 4:4:0 / 4:1:1 / 4:1:0 inputs)
from nvimagecodec_amd.synth import synth_image  # noqa: E402

SUB = {"444": 0, "422": 1, "420": 2}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pil_encode(img, quality, sub, progressive=False, **kw):
    b = io.BytesIO()
    im = Image.fromarray(img)
    if sub == "gray":
        im.convert("L").save(b, "JPEG", quality=quality, progressive=progressive, **kw)
    else:
        im.save(b, "JPEG", quality=quality, subsampling=SUB[sub], progressive=progressive, **kw)
    return b.getvalue()


def pil_decode(jpeg, mode=None):
    im = Image.open(io.BytesIO(jpeg))
    if mode:
        im = im.convert(mode)
    return np.asarray(im)


def main():
    assert features.check_feature("libjpeg_turbo"), "Pillow must be built against libjpeg-turbo"
    os.makedirs(os.path.join(HERE, "decode"), exist_ok=True)
    os.makedirs(os.path.join(HERE, "encode"), exist_ok=True)
    manifest = {"generator": "tests/golden/make_golden.py", "pillow": Image.__version__,
                "libjpeg_turbo": features.version("libjpeg_turbo"),
                "reference_pin": "libjpeg-turbo 3.0.1 (external/README.rst:98-104)", "decode": [], "encode": []}

    def add_decode(name, jpeg, keep_pixels, **meta):
        rgb = pil_decode(jpeg, "RGB")
        gray = pil_decode(jpeg, "L") if meta.get("sub") == "gray" else None
        with open(os.path.join(HERE, "decode", name + ".jpg"), "wb") as f:
            f.write(jpeg)
        if keep_pixels:
            with open(os.path.join(HERE, "decode", name + ".rgb"), "wb") as f:
                f.write(rgb.tobytes())
        e = dict(name=name, width=int(rgb.shape[1]), height=int(rgb.shape[0]), rgb_sha256=sha(rgb), pixels=bool(keep_pixels), **meta)
        if gray is not None:
            e["gray_sha256"] = sha(gray)
        manifest["decode"].append(e)

    # --- small matrix: full pixels kept
    sizes = [(8, 8), (17, 13), (33, 65), (64, 48), (50, 37), (1, 1), (3, 5)]
    for (w, h) in sizes:
        img = synth_image(w, h, seed=1000 + w * 100 + h)
        for sub in ("444", "422", "420", "gray"):
            for prog in (False, True):
                for q in (50, 90):
                    if (w, h) in ((1, 1), (3, 5)) and (q == 50 or prog):
                        continue
                    name = f"s{w}x{h}_{sub}_{'prog' if prog else 'base'}_q{q}"
                    add_decode(name, pil_encode(img, q, sub, prog), True, sub=sub, progressive=prog, quality=q, restart=0,
                               encoder="pillow")
    # --- restart intervals
    img = synth_image(130, 70, seed=77)
    for sub in ("444", "420"):
        for prog in (False, True):
            for rb in (1, 7):
                name = f"r130x70_{sub}_{'prog' if prog else 'base'}_rst{rb}"
                add_decode(name, pil_encode(img, 90, sub, prog, restart_marker_blocks=rb), True, sub=sub, progressive=prog, quality=90,
                           restart=rb, encoder="pillow")
    # --- optimized Huffman tables (image-specific code lengths: long codes, second-level lookups of the GPU entropy stage)
    for (w, h, seed) in ((64, 48, 41), (130, 70, 42), (320, 200, 43)):
        img = synth_image(w, h, seed=seed)
        for sub in ("444", "420", "gray"):
            for q in (75, 98):
                name = f"h{w}x{h}_{sub}_opt_q{q}"
                add_decode(name, pil_encode(img, q, sub, False, optimize=True), w <= 130, sub=sub, progressive=False, quality=q, restart=0,
                           encoder="pillow", optimize=True)
    # --- samplings Pillow cannot write: inputs from the oracle's encoder, goldens still from libjpeg-turbo's decoder
    for sub in ("440", "411", "410"):
        for (w, h) in ((50, 37), (64, 48), (17, 13)):
            img = synth_image(w, h, seed=2000 + w)
            name = f"o{w}x{h}_{sub}_base_q90"
            add_decode(name, oracle.encode(img, sub, 90), True, sub=sub, progressive=False, quality=90, restart=0, encoder="oracle")
    # --- the BASELINE.json configs: C1 (640x480 4:4:4), C2-shaped (1920x1080 4:2:0), C4/C5-shaped; hashes only
    big = [("c1_640x480_444_base_q90", 640, 480, "444", False, 1),
           ("c2_1920x1080_420_base_q90", 1920, 1080, "420", False, 2),
           ("c4_1280x720_422_base_q90", 1280, 720, "422", False, 3),
           ("c5_640x360_444_prog_q90", 640, 360, "444", True, 4)]
    for name, w, h, sub, prog, seed in big:
        img = synth_image(w, h, seed=seed)
        add_decode(name, pil_encode(img, 90, sub, prog), False, sub=sub, progressive=prog, quality=90, restart=0, encoder="pillow")

    # --- encode goldens: input pixels + libjpeg-turbo's bitstream
    for (w, h) in ((48, 32), (50, 37), (17, 13), (64, 48), (8, 8)):
        img = synth_image(w, h, seed=3000 + w)
        for sub in ("444", "422", "420", "gray"):
            for q in (50, 90):
                name = f"e{w}x{h}_{sub}_q{q}"
                jpeg = pil_encode(img, q, sub)
                src = img if sub != "gray" else np.repeat(np.asarray(Image.fromarray(img).convert("L"))[:, :, None], 3, axis=2)
                with open(os.path.join(HERE, "encode", name + ".rgb"), "wb") as f:
                    f.write(np.ascontiguousarray(src).tobytes())
                with open(os.path.join(HERE, "encode", name + ".jpg"), "wb") as f:
                    f.write(jpeg)
                manifest["encode"].append(dict(name=name, width=w, height=h, sub=sub, quality=q, rgb_sha256=sha(src),
                                               jpeg_sha256=hashlib.sha256(jpeg).hexdigest()))
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    total = sum(os.path.getsize(os.path.join(dp, fn)) for dp, _, fns in os.walk(HERE) for fn in fns)
    print(f"{len(manifest['decode'])} decode + {len(manifest['encode'])} encode vectors, {total/1e6:.2f} MB")


if __name__ == "__main__":
    main()
