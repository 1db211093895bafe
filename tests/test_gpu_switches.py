"""The library's measurement switches change HOW a batch is decoded, never WHAT comes out: every golden vector, in a child process per
switch (they are read once per process), bit-exact through both entropy stages."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HELPER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "decode_goldens.py")


@pytest.mark.parametrize("switches", [{"HIPJPEG_DEVICE_DESTUFF_COUNT": "1"}, {"HIPJPEG_TAIL_AFTER": "1"}, {"HIPJPEG_FUSED_DECODE": "1"}, {"HIPJPEG_DENSE_STAGING": "1"},
                                      {"HIPJPEG_DEVICE_DESTUFF_COUNT": "1", "HIPJPEG_TAIL_AFTER": "3", "HIPJPEG_FUSED_DECODE": "1"}],
                         ids=["device_destuff_count", "tail_after_1", "fused_decode", "dense_staging", "all"])
def test_goldens_under_switch(switches):
    env = dict(os.environ)
    env.update(switches)
    r = subprocess.run([sys.executable, HELPER], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "goldens ok" in r.stdout
