"""The nvimtrans-style command line tool (example/hipimtrans.cpp; reference example/nvimtrans/main.cpp:561-690): transcodes a
directory of JPEGs through nvimgcodecDecoderDecode / nvimgcodecEncoderEncode and prints per-stage images-per-second figures.
The files it writes must be what the oracle's encoder makes of the oracle's decode of the inputs."""
import os
import re
import subprocess

import numpy as np
import pytest

import oracle
from nvimagecodec_amd import _native as N
from nvimagecodec_amd.synth import synth_image

pytestmark = pytest.mark.gpu

TOOL = os.path.join(os.path.dirname(N.LIB_PATH), "hipimtrans")


def test_transcode_a_directory_and_report_stage_rates(tmp_path):
    assert os.path.exists(TOOL), "build the tool: make -C nvimagecodec_amd/csrc"
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    dst.mkdir()
    inputs = {}
    for i, (w, h, sub) in enumerate([(640, 480, "420"), (333, 217, "444"), (1280, 720, "422"), (64, 64, "420"), (800, 600, "420")]):
        j = oracle.encode(synth_image(w, h, seed=40 + i), sub, 85)
        inputs["img%02d.jpg" % i] = j
        (src / ("img%02d.jpg" % i)).write_bytes(j)
    p = subprocess.run([TOOL, "-i", str(src), "-o", str(dst), "-b", "3", "-w", "1", "-q", "90", "-s", "420"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    for stage in ("transcoding", "reading", "parsing", "decoding", "encoding"):
        m = re.search(r"Avg %s speed  \(in images per sec\): ([0-9.]+)" % stage, p.stdout)
        assert m and float(m.group(1)) > 0, (stage, p.stdout)
    assert "Total images: 5 (failed: 0)" in p.stdout
    for name, j in inputs.items():
        out = (dst / name).read_bytes()
        assert out == oracle.encode(oracle.decode(j), "420", 90), name


def test_decode_only_mode_and_a_bad_file(tmp_path):
    src = tmp_path / "in"
    src.mkdir()
    (src / "a.jpg").write_bytes(oracle.encode(synth_image(320, 240, seed=1), "420", 90))
    (src / "b.jpg").write_bytes(b"\xff\xd8 this is not a jpeg")
    p = subprocess.run([TOOL, "-i", str(src), "-b", "2", "-w", "0"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0      # the parser refuses b.jpg: the tool stops like the reference's CHECK_NVIMGCODEC
    p = subprocess.run([TOOL, "-i", str(src / "a.jpg"), "-b", "4", "-r", "8", "-w", "1"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "Total images: 8 (failed: 0)" in p.stdout and "Avg decoding speed" in p.stdout
