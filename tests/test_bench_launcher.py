"""bench.py's launcher logic, without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=e)


def test_more_gpus_than_visible_is_refused():
    import torch
    if torch.cuda.device_count() >= 2:
        return
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0 and "refusing to measure fewer than asked for" in p.stderr


def test_world_size_must_match_gpus():
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr


def test_without_a_gpu_the_bench_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    p = _run(["--steps", "1", "--warmup", "0"])
    assert p.returncode != 0 and "needs a GPU" in p.stderr
