"""Multi-GPU work partitioning for batched decode: every image is an independent unit, so a batch is split
into per-device queues on the host and each rank decodes its own queue -- there is no collective on the data path.

The reference has no multi-GPU path of its own (one decoder instance per device_id, src/default_executor.cpp:51-52); the
partitioning rule follows what its generic decoder already does inside one device: sort samples by decreasing size
(src/image_generic_decoder.cpp:134-178) and hand them out greedily.
"""
from . import lowlevel


def image_cost(jpeg):
    """Work estimate of one image: bytes of MCU-padded coefficient blocks the device stage touches plus the bitstream bytes
    the entropy stage walks.  Header parse only (no GPU needed)."""
    info = lowlevel.get_image_info(jpeg)
    return int(info["coef_bytes"]) + len(jpeg)


def shard_indices(costs, world_size):
    """Greedy longest-processing-time partition of items (given by their costs) into world_size queues.
    Deterministic: ties go to the lowest rank, items are visited in (cost desc, index asc) order.
    Returns a list of index lists, one per rank; every index appears exactly once."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    queues = [[] for _ in range(world_size)]
    load = [0] * world_size
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        queues[r].append(i)
        load[r] += costs[i]
    return queues


def shard_batch(jpegs, world_size, rank):
    """Indices of the images rank `rank` decodes, out of a batch every rank sees in the same order."""
    return shard_indices([image_cost(j) for j in jpegs], world_size)[rank]


def max_over_ranks(value, dist=None, device=None):
    """The job's wall time is the slowest rank's: MAX-reduce a float over the process group (bench.py contract).
    dist = torch.distributed (initialised) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
