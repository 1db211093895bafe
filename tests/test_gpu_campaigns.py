"""Short runs of the randomized cross-check campaigns (tests/campaigns/*.py: the long versions are run by hand on the GPU box).
Every campaign compares independent routes of the product with each other and a sample with the CPU oracle, bit for bit."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("script,seed,rounds", [("fuzz_gpu.py", 2, 2), ("fuzz_damage.py", 3, 4), ("fuzz_geometry.py", 4, 2), ("fuzz_outputs.py", 5, 2),
                                                ("fuzz_encode.py", 6, 4), ("fuzz_plugin.py", 7, 3), ("fuzz_pass1.py", 8, 2)])
def test_campaign(script, seed, rounds, monkeypatch, capsys):
    monkeypatch.setattr(sys, "argv", [script, str(seed), str(rounds)])
    runpy.run_path(os.path.join(HERE, "campaigns", script), run_name="__main__")
    assert "ok" in capsys.readouterr().out.splitlines()[-1]
