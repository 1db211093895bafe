"""The N>1 path with a real decode behind it: two processes, launched exactly as the driver launches bench.py
(python -m torch.distributed.run, 127.0.0.1), share GPU 0 of the one-GPU box; each decodes its queue of a mixed-shape batch
through the HIP path and checks it against the oracle; the union of the queues must be the batch.  No collective on the data
path -- the process group carries only the verdicts."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_decode_their_queues_on_one_gpu():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "helpers", "two_rank_decode.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = next(l for l in p.stdout.splitlines() if l.startswith("TWO_RANK_RESULT "))
    res = json.loads(line[len("TWO_RANK_RESULT "):])
    ranks = sorted(res["ranks"], key=lambda r: r["rank"])
    assert len(ranks) == 2 and all(r["ok"] for r in ranks)
    assert sorted(ranks[0]["indices"] + ranks[1]["indices"]) == list(range(res["n"]))   # complete and disjoint
    assert ranks[0]["indices"] and ranks[1]["indices"]
    assert res["max_over_ranks"] == 2.0


def test_bench_refuses_to_measure_fewer_gpus_than_asked_for():
    """`python bench.py --gpus 8` on a box with fewer GPUs must fail loudly instead of silently timing one (round-1 verdict)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300)
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this box really has 8 GPUs")
    assert p.returncode != 0 and "refusing" in (p.stderr + p.stdout)
