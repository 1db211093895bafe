"""The N>1 path on CPU: two gloo ranks partition one mixed-shape batch (independent per-GPU queues, no
collective on the data path) and reduce their timings the way bench.py does.  The decode itself needs a GPU and is covered by
the -m gpu tests; what is checked here is everything that differs between N=1 and N>1."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from nvimagecodec_amd import sharding  # noqa: E402
from nvimagecodec_amd.synth import synth_image  # noqa: E402

SHAPES = [(640, 480, "420"), (1280, 720, "422"), (320, 200, "444"), (1920, 1080, "420"), (97, 61, "420"), (800, 600, "422"), (64, 64, "gray"),
          (1024, 768, "420"), (333, 777, "422")]


def _batch():
    return [oracle.encode(synth_image(w, h, seed=i), sub, 85) for i, (w, h, sub) in enumerate(SHAPES)]


def test_partition_is_complete_disjoint_and_balanced():
    jpegs = _batch()
    costs = [sharding.image_cost(j) for j in jpegs]
    for world in (1, 2, 3, 8):
        queues = sharding.shard_indices(costs, world)
        flat = sorted(i for q in queues for i in q)
        assert flat == list(range(len(jpegs)))
        loads = [sum(costs[i] for i in q) for q in queues]
        # LPT guarantee: no queue exceeds the mean by more than the largest single item
        assert max(loads) <= sum(costs) / world + max(costs)
    assert sharding.shard_indices([], 4) == [[], [], [], []]
    with pytest.raises(ValueError):
        sharding.shard_indices(costs, 0)


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        jpegs = _batch()  # every rank sees the same batch in the same order
        mine = sharding.shard_batch(jpegs, world, rank)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        # the job's time is the slowest rank's
        t = sharding.max_over_ranks(1.0 + rank, dist)
        single = sharding.max_over_ranks(0.25, None)
        ret[rank] = (mine, gathered, t, single)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        r0, r1 = ret[0], ret[1]
    n = len(SHAPES)
    assert sorted(r0[0] + r1[0]) == list(range(n))          # complete, disjoint
    assert r0[1] == r1[1] == [r0[0], r1[0]]                   # both ranks derive the same partition
    assert r0[2] == r1[2] == 2.0                              # MAX over ranks
    assert r0[3] == 0.25
