"""Worker of tests/test_gpu_two_ranks.py: started twice by torch.distributed.run (gloo rendezvous on 127.0.0.1), both ranks on
GPU 0 of the one-GPU box.  Every rank takes ITS queue of one mixed-shape batch (sharding.shard_batch, the partition
bench.py uses for BASELINE configs[3]), decodes it with its own decoder instance and checks its outputs against the oracle;
rank 0 then checks that the two queues together cover the batch exactly once and prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import oracle  # noqa: E402
from nvimagecodec_amd import sharding  # noqa: E402
from nvimagecodec_amd.lowlevel import BatchDecoder  # noqa: E402
from nvimagecodec_amd.synth import synth_image  # noqa: E402

SHAPES = [(640, 480, "420"), (1280, 720, "422"), (320, 200, "444"), (1920, 1080, "420"), (97, 61, "420"), (800, 600, "422"), (64, 64, "gray"),
          (1024, 768, "420"), (333, 777, "422"), (2560, 1440, "420"), (1920, 1080, "422"), (641, 479, "420")]


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    pin = sharding.pin_to_device_numa(0, world)
    jpegs = [oracle.encode(synth_image(w, h, seed=i), sub, 85) for i, (w, h, sub) in enumerate(SHAPES)]  # same on every rank
    mine = sharding.shard_batch(jpegs, world, rank)
    dec = BatchDecoder(device=0, num_threads=2)
    batch = [jpegs[i] for i in mine]
    outs = dec.allocate_outputs(batch, "rgb")
    dec.submit(batch, outs, gpu_huffman=True)
    statuses = dec.wait()
    torch.cuda.synchronize()
    ok = all(s == 0 for s in statuses) and all(np.array_equal(o.cpu().numpy(), oracle.decode(j)) for j, o in zip(batch, outs))
    elapsed = sharding.max_over_ranks(1.0 + rank, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, {"rank": rank, "indices": mine, "ok": bool(ok), "pinned": pin})
    dec.close()
    dist.barrier()
    if rank == 0:
        print("TWO_RANK_RESULT " + json.dumps({"ranks": gathered, "n": len(jpegs), "max_over_ranks": elapsed}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
