#!/usr/bin/env python3
"""Golden hashes for do_fancy_upsampling = FALSE (plugin option fancy_upsampling=0), from the REAL libjpeg-turbo.  Dev-container only.

Pillow cannot switch fancy upsampling off, so until round 3 the replication path of the oracle and the kernels was "parity unpinned".
This script drives the libjpeg-turbo that Pillow ships (pillow.libs/libjpeg-*.so.62.*, the library the other goldens come from) through
its public C API with ctypes: jpeg_std_error, jpeg_CreateDecompress, jpeg_mem_src, jpeg_read_header, jpeg_start_decompress,
jpeg_read_scanlines, jpeg_finish_decompress -- and clears cinfo.do_fancy_upsampling in between.

No header of the library is installed here, so the two facts about `struct jpeg_decompress_struct` the script needs are MEASURED, not
assumed:
  * its size: jpeg_CreateDecompress refuses a wrong size with JERR_BAD_STRUCT_SIZE and names the right one in the message parameters
    (a child process asks with size 1 and reads the answer in its error_exit callback);
  * the offsets of the fields it reads and writes (image_width/height, num_components, out_color_space, do_fancy_upsampling,
    output_width/height/components, output_scanline; libjpeg API v6.2 layout on LP64): the binding decodes EVERY file with fancy
    upsampling left ON first and the pixels must equal Pillow's decode of the same file -- and with it OFF every 4:4:4 and gray file must
    still equal Pillow's (nothing to upsample), every subsampled one must differ.  A wrong offset fails one of the three.

Output: manifest_plain.json -- per file of tests/golden/decode: sha256 of the H x W x 3 RGB pixels (gray files: expanded to RGB like the
other goldens) decoded with do_fancy_upsampling = FALSE; "cmyk": the same for the four-component files of tests/golden/cmyk (the library's
CMYK samples; with the switch on they must equal the stored goldens); and "roi": regions of interest decoded the way the reference's CPU path decodes
them (extensions/libjpeg_turbo/jpeg_mem.cpp:206-240: one spare pixel left and right, jpeg_crop_scanline, jpeg_skip_scanlines), with fancy
upsampling on and off -- the script asserts that each equals the same window of the full decode (504 random windows did, in both modes,
before these were chosen), which is the semantics the geometry pass of the kernels implements.  (For 4:2:0 / 4:2:2 the library then takes its merged upsampling + colour
conversion path, jdmerge.c -- the reference plugin's `fancy_upsampling=0` does exactly this, extensions/libjpeg_turbo/jpeg_mem.cpp:166.)"""
import ctypes as C
import glob
import hashlib
import io
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# struct jpeg_decompress_struct, libjpeg API 6.2, LP64 (validated below)
OFF_IMAGE_WIDTH, OFF_IMAGE_HEIGHT, OFF_NUM_COMPONENTS, OFF_OUT_COLOR_SPACE = 48, 52, 56, 64
OFF_DCT_METHOD, OFF_DO_FANCY = 96, 100
OFF_OUTPUT_WIDTH, OFF_OUTPUT_HEIGHT, OFF_OUTPUT_COMPONENTS, OFF_OUTPUT_SCANLINE = 136, 140, 148, 168
JCS_GRAYSCALE, JCS_RGB, JCS_CMYK = 1, 2, 4
# struct jpeg_error_mgr: error_exit at 0, msg_code at 40, msg_parm.i[] at 44
ERR_MSG_CODE, ERR_PARM = 40, 44


def library_path():
    import PIL
    hits = glob.glob(os.path.join(os.path.dirname(PIL.__file__), "..", "pillow.libs", "libjpeg-*.so.62*"))
    assert len(hits) == 1, hits
    return os.path.abspath(hits[0])


def load():
    lib = C.CDLL(library_path())
    lib.jpeg_std_error.restype = C.c_void_p
    lib.jpeg_std_error.argtypes = [C.c_void_p]
    lib.jpeg_CreateDecompress.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    lib.jpeg_mem_src.argtypes = [C.c_void_p, C.c_void_p, C.c_ulong]
    lib.jpeg_read_header.argtypes = [C.c_void_p, C.c_int]
    lib.jpeg_start_decompress.argtypes = [C.c_void_p]
    lib.jpeg_read_scanlines.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    lib.jpeg_read_scanlines.restype = C.c_uint
    lib.jpeg_finish_decompress.argtypes = [C.c_void_p]
    lib.jpeg_destroy_decompress.argtypes = [C.c_void_p]
    lib.jpeg_crop_scanline.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    lib.jpeg_skip_scanlines.argtypes = [C.c_void_p, C.c_uint]
    lib.jpeg_skip_scanlines.restype = C.c_uint
    lib.jpeg_abort_decompress.argtypes = [C.c_void_p]
    return lib


ERROR_EXIT = C.CFUNCTYPE(None, C.c_void_p)


def probe_struct_size():
    """Runs in a child: asks jpeg_CreateDecompress with a wrong size and prints what the library says the size is."""
    lib = load()
    err = C.create_string_buffer(1024)
    cinfo = C.create_string_buffer(4096)

    @ERROR_EXIT
    def on_error(cptr):
        e = C.cast(cptr, C.POINTER(C.c_void_p))[0]  # cinfo->err
        code = C.cast(e + ERR_MSG_CODE, C.POINTER(C.c_int))[0]
        parm = C.cast(e + ERR_PARM, C.POINTER(C.c_int))
        print("probe", code, parm[0], parm[1], flush=True)
        os._exit(0)  # the library cannot be returned into from an error

    lib.jpeg_std_error(err)
    C.cast(err, C.POINTER(C.c_void_p))[0] = C.cast(on_error, C.c_void_p).value
    C.cast(cinfo, C.POINTER(C.c_void_p))[0] = C.addressof(err)
    lib.jpeg_CreateDecompress(cinfo, 62, 1)
    print("probe none", flush=True)


def struct_size():
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--probe"], capture_output=True, text=True, timeout=120).stdout.split()
    # JERR_BAD_STRUCT_SIZE: parameters = (size the library was built with, size given)
    assert out[0] == "probe" and out[1] != "none" and int(out[3]) == 1, out
    return int(out[2])


class Decoder:
    def __init__(self):
        self.lib = load()
        self.size = struct_size()
        assert 400 < self.size < 1024, self.size

    def decode(self, data, fancy, roi=None):
        """roi = (x, y, w, h): the reference's recipe, jpeg_mem.cpp:206-240"""
        lib = self.lib
        err = C.create_string_buffer(1024)
        cinfo = C.create_string_buffer(self.size + 64)
        base = C.addressof(cinfo)

        def i32(off):
            return C.cast(base + off, C.POINTER(C.c_int))

        lib.jpeg_std_error(err)  # the default error_exit ends the process: the caller runs this in a child and checks its exit
        C.cast(cinfo, C.POINTER(C.c_void_p))[0] = C.addressof(err)
        lib.jpeg_CreateDecompress(cinfo, 62, self.size)
        buf = C.create_string_buffer(bytes(data), len(data))
        lib.jpeg_mem_src(cinfo, buf, len(data))
        assert lib.jpeg_read_header(cinfo, 1) == 1
        width, height, ncomp = i32(OFF_IMAGE_WIDTH)[0], i32(OFF_IMAGE_HEIGHT)[0], i32(OFF_NUM_COMPONENTS)[0]
        assert i32(OFF_DCT_METHOD)[0] == 0 and i32(OFF_DO_FANCY)[0] == 1  # defaults: JDCT_ISLOW, fancy upsampling on
        assert i32(OFF_OUT_COLOR_SPACE)[0] == {1: JCS_GRAYSCALE, 3: JCS_RGB, 4: JCS_CMYK}[ncomp]  # (YCCK sources come out as CMYK too)
        i32(OFF_DO_FANCY)[0] = 1 if fancy else 0
        lib.jpeg_start_decompress(cinfo)
        ow, oh, oc = i32(OFF_OUTPUT_WIDTH)[0], i32(OFF_OUTPUT_HEIGHT)[0], i32(OFF_OUTPUT_COMPONENTS)[0]
        assert (ow, oh) == (width, height) and oc == ncomp, (ow, oh, oc)
        row = (C.c_void_p * 1)()
        if roi is None:
            out = np.zeros((oh, ow * oc), dtype=np.uint8)
            while i32(OFF_OUTPUT_SCANLINE)[0] < oh:
                y = i32(OFF_OUTPUT_SCANLINE)[0]
                row[0] = out.ctypes.data + y * out.strides[0]
                assert lib.jpeg_read_scanlines(cinfo, row, 1) == 1
            lib.jpeg_finish_decompress(cinfo)
            out = out.reshape(oh, ow, oc)
        else:
            x, y, w, h = roi
            left = 0 if x == 0 else 1
            right = max(0, min(1, ow - (x + w)))
            cx, cw = C.c_uint(x - left), C.c_uint(w + left + right)
            lib.jpeg_crop_scanline(cinfo, C.byref(cx), C.byref(cw))  # moves the left edge down to an iMCU boundary
            assert i32(OFF_OUTPUT_WIDTH)[0] == cw.value
            assert lib.jpeg_skip_scanlines(cinfo, y) == y
            out = np.zeros((h, cw.value * oc), dtype=np.uint8)
            for r in range(h):
                row[0] = out.ctypes.data + r * out.strides[0]
                assert lib.jpeg_read_scanlines(cinfo, row, 1) == 1
            lib.jpeg_abort_decompress(cinfo)
            out = out.reshape(h, cw.value, oc)[:, x - cx.value:x - cx.value + w]
        lib.jpeg_destroy_decompress(cinfo)
        return np.repeat(out, 3, axis=2) if oc == 1 else out


def main():
    from PIL import Image, features
    manifest = json.load(open(os.path.join(HERE, "manifest.json")))
    assert features.version_feature("libjpeg_turbo") == manifest["libjpeg_turbo"], "the goldens come from another library version"
    dec = Decoder()
    cases = []
    differ = 0
    for c in manifest["decode"]:
        data = open(os.path.join(HERE, "decode", c["name"] + ".jpg"), "rb").read()
        pil = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        assert hashlib.sha256(pil.tobytes()).hexdigest() == c["rgb_sha256"], c["name"]
        on = dec.decode(data, True)
        assert np.array_equal(on, pil), ("the binding's decode with fancy upsampling on is not Pillow's", c["name"])
        off = dec.decode(data, False)
        if c["sub"] in ("444", "gray"):
            assert np.array_equal(off, pil), ("nothing to upsample, yet the switch changed pixels", c["name"])
        elif c["width"] > 2 and c["height"] > 2:
            differ += int(not np.array_equal(off, pil))
        cases.append({"name": c["name"], "sub": c["sub"], "width": c["width"], "height": c["height"],
                      "plain_rgb_sha256": hashlib.sha256(np.ascontiguousarray(off).tobytes()).hexdigest()})
    assert differ > 40, differ  # the switch reached the library
    # regions of interest: files of every sampling, baseline / progressive / restart intervals; windows on and off MCU boundaries
    rois = []
    rng = np.random.default_rng(2026)
    picked = [c for c in manifest["decode"] if c["width"] >= 48 and c["height"] >= 37]
    picked = [c for i, c in enumerate(picked) if i % 3 == 0 or c["sub"] in ("440", "411", "410")]
    for c in picked:
        data = open(os.path.join(HERE, "decode", c["name"] + ".jpg"), "rb").read()
        W, H = c["width"], c["height"]
        windows = [(0, 0, W, H), (1, 1, W - 2, H - 2), (16, 8, 16, 16), (W - 9, H - 7, 9, 7)]
        for _ in range(3):
            x, y = int(rng.integers(0, W - 4)), int(rng.integers(0, H - 4))
            windows.append((x, y, int(rng.integers(1, W - x + 1)), int(rng.integers(1, H - y + 1))))
        for fancy in (True, False):
            full = dec.decode(data, fancy)
            for (x, y, w, h) in windows:
                got = dec.decode(data, fancy, roi=(x, y, w, h))
                assert np.array_equal(got, full[y:y + h, x:x + w]), ("a region of interest that is not the window of the full decode", c["name"], fancy, x, y, w, h)
                rois.append({"name": c["name"], "fancy": fancy, "roi": [x, y, w, h], "rgb_sha256": hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest()})
    # four-component files (tests/golden/cmyk): the library's CMYK samples, which the reference turns into RGB itself (jpeg_mem.cpp:292-337)
    cmyk = []
    cm = json.load(open(os.path.join(HERE, "manifest_cmyk.json")))
    assert cm["libjpeg_turbo"] == manifest["libjpeg_turbo"]
    cmyk_differ = 0
    for c in cm["cmyk"]:
        data = open(os.path.join(HERE, "cmyk", c["name"] + ".jpg"), "rb").read()
        ref = np.fromfile(os.path.join(HERE, "cmyk", c["name"] + ".cmyk"), dtype=np.uint8).reshape(c["height"], c["width"], 4)
        assert np.array_equal(dec.decode(data, True), ref), ("the binding's CMYK samples are not the golden ones", c["name"])
        off = dec.decode(data, False)
        assert c["subsampled"] or np.array_equal(off, ref), c["name"]
        cmyk_differ += int(not np.array_equal(off, ref))
        cmyk.append({"name": c["name"], "kind": c["kind"], "subsampled": c["subsampled"], "width": c["width"], "height": c["height"],
                     "plain_cmyk_sha256": hashlib.sha256(np.ascontiguousarray(off).tobytes()).hexdigest()})
    assert cmyk_differ >= 10, cmyk_differ
    out = {"generator": "tests/golden/make_golden_plain_upsampling.py", "libjpeg_turbo": manifest["libjpeg_turbo"], "library": os.path.basename(library_path()),
           "jpeg_decompress_struct_bytes": dec.size, "subsampled_files_that_differ_from_fancy": differ, "decode": cases, "cmyk": cmyk, "roi": rois}
    json.dump(out, open(os.path.join(HERE, "manifest_plain.json"), "w"), indent=1)
    print("wrote manifest_plain.json:", len(cases), "files,", differ, "differ from the fancy decode;", len(rois), "regions of interest; struct size", dec.size)


if __name__ == "__main__":
    if "--probe" in sys.argv:
        probe_struct_size()
    else:
        main()
