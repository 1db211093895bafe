import json
import os
import sys

import numpy as np
import pytest

# the library's hipjpegTest* entry points (fault injection, counters) answer only in processes started with this (read once at first use;
# child processes of the tests inherit it)
os.environ.setdefault("HIPJPEG_ENABLE_TEST_HOOKS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def load_decode_case(entry):
    """-> (jpeg bytes, golden RGB array or None)"""
    with open(os.path.join(GOLDEN, "decode", entry["name"] + ".jpg"), "rb") as f:
        jpeg = f.read()
    rgb = None
    if entry["pixels"]:
        rgb = np.fromfile(os.path.join(GOLDEN, "decode", entry["name"] + ".rgb"), dtype=np.uint8).reshape(entry["height"], entry["width"], 3)
    return jpeg, rgb


def load_encode_case(entry):
    rgb = np.fromfile(os.path.join(GOLDEN, "encode", entry["name"] + ".rgb"), dtype=np.uint8).reshape(entry["height"], entry["width"], 3)
    with open(os.path.join(GOLDEN, "encode", entry["name"] + ".jpg"), "rb") as f:
        jpeg = f.read()
    return rgb, jpeg
