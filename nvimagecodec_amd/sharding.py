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


# ---------------------------------------------------------------------------------------------- host placement
def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def device_numa_cpus(device):
    """(numa node, CPUs of that node) of HIP device `device`, from its PCI address in sysfs; (None, None) when the platform
    does not say (containers often hide it)."""
    try:
        import torch
        p = torch.cuda.get_device_properties(int(device))
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return None, None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            return node, _parse_cpulist(f.read())
    except Exception:
        return None, None


def split_evenly(cpus, parts, index):
    """The index-th of `parts` contiguous slices of the sorted CPU list (every slice non-empty when len(cpus) >= parts)."""
    cpus = sorted(cpus)
    lo, hi = len(cpus) * index // parts, len(cpus) * (index + 1) // parts
    return cpus[lo:hi] if hi > lo else cpus


def pin_to_device_numa(local_rank, world_size):
    """Restrict this process (and the threads it creates from now on) to host CPUs near its GPU: the CPUs of the GPU's NUMA
    node that the process is allowed to use, shared evenly between the ranks whose GPUs sit on the same node; without NUMA
    information the allowed CPUs are simply split evenly between the ranks.  Each device's thread pool then competes with
    nobody (the reference keys its pools by device, src/default_executor.cpp:45-58, but leaves placement to the OS).
    Returns a small description for the bench line; never raises."""
    import os
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except Exception:
        return {"pinned": None, "why": "no affinity interface"}
    if world_size <= 1:
        return {"pinned": None, "why": "single rank: all %d allowed CPUs" % len(allowed)}
    node, cpus = device_numa_cpus(local_rank)
    try:
        if cpus:
            near = sorted(set(allowed) & cpus)
            # ranks whose GPUs share this node split its CPUs
            peers = [r for r in range(world_size) if device_numa_cpus(r)[0] == node]
            if near and len(near) >= len(peers):
                mine = split_evenly(near, len(peers), peers.index(local_rank))
                os.sched_setaffinity(0, mine)
                return {"pinned": len(mine), "numa_node": node, "ranks_on_node": len(peers)}
        mine = split_evenly(allowed, world_size, local_rank)
        os.sched_setaffinity(0, mine)
        return {"pinned": len(mine), "numa_node": None, "why": "no NUMA information for the device: allowed CPUs split evenly"}
    except Exception as e:
        return {"pinned": None, "why": repr(e)}
