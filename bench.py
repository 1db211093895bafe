#!/usr/bin/env python3
"""Contract benchmark: batched 1920x1080 4:2:0 baseline JPEG decode -> interleaved RGB u8 (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  N > 1 without WORLD_SIZE in the environment: bench.py starts the N ranks itself (torch.distributed.run, 127.0.0.1)
  before anything touches a GPU; under the driver's own torch.distributed.run it just reads RANK / LOCAL_RANK / WORLD_SIZE.

One step = ONE FULL DECODE of a batch of 256 images whose JPEG bitstreams are already resident in HBM: byte-stuffing removal
and Huffman decoding on the GPU (the GPU entropy stage), dequantize + ISLOW IDCT + fancy chroma upsampling + YCbCr->RGB +
interleaved store (the device stage).  Nothing of the decode is outside the timed region; nothing is cached between steps.
`value` = images through that path per second, W warm-up steps then exactly K timed steps between barriers, max over ranks.
The boundary hands over HOST bytes, so the PCIe-inclusive rates ride on the same JSON line under "end_to_end" (pipelined
GPU-entropy path, and the north-star split with Huffman on the host cores) -- they are never `value`.

Also on the line:
  roofline       SURVEY.md 8(d): algorithmic bytes of the device stage (12,487,680 B per image = int16 coefficient blocks read
                 once + RGB written once) / HIP-event time of its two kernels inside the timed steps, vs the 8 TB/s HBM3E peak;
                 `traffic` = HBM bytes from a PMC pass of THIS source tree (profiles/r02_hbm_traffic.json carries a hash of the
                 kernel sources; any other tree prints null)
  entropy_stage  the same for the GPU entropy stage (bitstream bytes in + coefficient bytes out per step / event time)
  steady_state   the timed protocol once more after a long untimed run (GPU clocks settled) -- reported next to `value`,
                 never instead of it
  cpu_baseline   real libjpeg-turbo (through Pillow, when the box has it) on the host cores, T=1 and T=all, including the
                 reference extension's extra row copy; the oracle port beside it
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 256
WIDTH, HEIGHT = 1920, 1080
ALG_BYTES_PER_IMAGE = 12_487_680  # 48,960 blocks * 128 B + 1920*1080*3 B
COEF_BYTES_PER_IMAGE = 6_266_880  # 48,960 blocks * 128 B
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0             # ... and what a float4 device-to-device copy reaches there ("report fraction of both")
NUM_SOURCES = 8                   # distinct synthetic images cycled through the batch
STEADY_PREWARM_STEPS = 40
# BENCH_REHEARSE_ON_ONE_GPU=1: every rank on GPU 0 with a gloo process group -- a rehearsal of the N > 1 code path (spawning, sharding
# of configs[3], NUMA split, max over ranks, the JSON line) on a box with one card.  Its figures mean nothing and the line says so.
REHEARSAL = os.environ.get("BENCH_REHEARSE_ON_ONE_GPU") == "1"
REDUCE_DEVICE = "cpu" if REHEARSAL else "cuda"  # where the max-over-ranks tensor lives (gloo reduces host tensors)
PROGRESSIVE_DEPTH = 6             # batches in flight for configs[4] (hipjpegSetPipelineDepth): a progressive batch is one wave per scan
# (every batch in flight runs its entropy stage on a stream of its own; the HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES
# hardware queues, default 4.  The LIBRARY asks for twelve when it is loaded -- csrc/hipjpeg_api.cpp hipjpeg_runtime_defaults -- which
# works as long as it is loaded before the process's first HIP call: main() loads it before touching torch.cuda.)


def make_inputs():
    """8 distinct seeded 1080p photo-like images, baseline 4:2:0 q90, standard Huffman tables, no restart markers."""
    from nvimagecodec_amd.synth import synth_image
    imgs = [synth_image(WIDTH, HEIGHT, seed=1234 + s) for s in range(NUM_SOURCES)]
    try:
        return [_pil_encode(im, 90, "420") for im in imgs], "synthetic (seeded images, libjpeg-turbo/Pillow-encoded q90 4:2:0)"
    except ImportError:
        return _product_encode(imgs, "420", 90), "synthetic (seeded images, encoded by this package's GPU encoder q90 4:2:0)"


def _pil_encode(im, quality, sub, progressive=False):
    import io
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(im).save(b, "JPEG", quality=quality, subsampling={"444": 0, "422": 1, "420": 2}[sub], progressive=progressive)
    return b.getvalue()


def _product_encode(imgs, sub, quality):
    import torch
    from nvimagecodec_amd.lowlevel import BatchEncoder
    enc = BatchEncoder(device=torch.cuda.current_device(), num_threads=2, gpu_huffman=True)
    out = enc.encode([torch.from_numpy(im).cuda() for im in imgs], sub, quality, "rgb")
    enc.close()
    return out


def usable_cpus():
    """CPU threads this process may really use: affinity mask and cgroup quota, not the host's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def kernel_source_hash():
    """Identifies the tree a PMC traffic file was measured on: sha256 over the kernel and host sources of the extension."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "nvimagecodec_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def measured_traffic():
    """HBM bytes per step of the roofline kernels from a PMC pass (tools/round_profile.sh) -- only if it was taken on this tree."""
    import glob
    sha = kernel_source_hash()
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            with open(p) as f:
                t = json.load(f)
            if t.get("source_sha") == sha:
                return t.get("roofline_kernels_bytes_per_step")
        except Exception:
            pass
    return None


# ------------------------------------------------------------------------------------------------------------ launcher
def spawn_ranks(args):
    """`python bench.py --gpus N` on its own: start N ranks, one per GPU, before any GPU call is made in this process."""
    import torch  # device_count() does not initialise the GPU
    have = torch.cuda.device_count()
    if have < args.gpus and not REHEARSAL:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node -- refusing to measure fewer than asked for")
    port = 29000 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


# ------------------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(sources):
    """The reference's CPU path for the same bitstreams on the host cores: libjpeg-turbo (JDCT_ISLOW, fancy upsampling, JCS_RGB)
    to interleaved RGB in host memory plus the extension's extra copy into the user buffer, row by row
    (ref extensions/libjpeg_turbo/libjpeg_turbo_decoder.cpp:414-436).  Bounded sample, ~10-30 s of CPU work in all."""
    import concurrent.futures as cf
    import numpy as np
    import oracle
    threads = min(usable_cpus(), 64)

    def run(fn, n, t):
        jobs = [sources[i % len(sources)] for i in range(n)]
        fn(jobs[0])
        t0 = time.perf_counter()
        if t == 1:
            for j in jobs:
                fn(j)
        else:
            with cf.ThreadPoolExecutor(t) as ex:  # Pillow's decoder and the oracle's ctypes call release the GIL
                list(ex.map(fn, jobs))
        return n / (time.perf_counter() - t0)

    def port(j):
        return oracle.decode(j)

    res = None
    try:
        import io
        from PIL import Image, features
        if features.check_feature("libjpeg_turbo"):
            user = np.empty((threads + 1, HEIGHT, WIDTH, 3), dtype=np.uint8)
            import threading
            slot = threading.local()
            counter = [0]
            lock = threading.Lock()

            def turbo(j):
                if not hasattr(slot, "i"):
                    with lock:
                        slot.i = counter[0] % (threads + 1)
                        counter[0] += 1
                a = np.asarray(Image.open(io.BytesIO(j)).convert("RGB"))
                dst = user[slot.i]
                np.copyto(dst, a)  # the extension's copy of the decoded picture into the user's buffer (:431-436)
                return dst

            n_all = 16 * threads
            v_all = run(turbo, n_all, threads)
            v_one = run(turbo, 48, 1)
            res = {"value": round(v_all, 2), "unit": "images/s", "cores": threads, "kind": "reference",
                   "library": "libjpeg-turbo %s through Pillow: the library ref:extensions/libjpeg_turbo calls (JDCT_ISLOW, fancy "
                              "upsampling, JCS_RGB) + that extension's row copy into the user buffer; the extension's own wrapper "
                              "(oracle/_ref) is unbuildable here" % features.version("libjpeg_turbo"),
                   "sample": f"{n_all} decodes of the bench bitstreams (1920x1080 4:2:0 q90) on {threads} threads; T=1: 48 decodes",
                   "threads_1": {"value": round(v_one, 2), "cores": 1}}
    except Exception:
        res = None
    n_port = 12 * threads
    v_port = run(port, n_port, threads)
    port_res = {"value": round(v_port, 2), "unit": "images/s", "cores": threads, "kind": "port",
                "sample": f"{n_port} decodes of the same bitstreams by oracle/jpeg_oracle.c on {threads} threads"}
    if res is None:
        return port_res
    res["port"] = port_res
    return res


# ------------------------------------------------------------------------------------------------------------ secondary figures
def config3_sharded(dec, rank, world, dist, host_threads):
    """BASELINE configs[3]: 2048 mixed-shape images (480p..4K, 50/50 4:2:0 / 4:2:2) partitioned over the ranks by
    sharding.shard_batch (LPT over coefficient + bitstream bytes); every rank decodes ITS queue, pipelined in pieces of 256;
    the job's time is the slowest rank's."""
    import torch
    from nvimagecodec_amd import sharding
    from nvimagecodec_amd.synth import synth_image
    shapes = [(640, 480), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160)]
    srcs = []
    for k in range(10):
        w, h = shapes[k % 5]
        im = synth_image(w, h, seed=500 + k)
        try:
            srcs.append(_pil_encode(im, 90, "420" if k < 5 else "422"))
        except ImportError:
            srcs.append(_product_encode([im], "420" if k < 5 else "422", 90)[0])
    total = 2048
    order = [(7 * i + 3) % 10 for i in range(total)]  # fixed pseudo-random draw, identical on every rank
    costs = [sharding.image_cost(s) for s in srcs]
    mine = sharding.shard_indices([costs[k] for k in order], world)[rank]
    # pieces of 256 with the same mix of sizes each (every npieces-th image of the size-sorted queue), so that the three
    # staging pages of the decoder see similar batches and never have to grow in the middle of the run
    npieces = max(1, (len(mine) + BATCH - 1) // BATCH)
    pieces = [mine[k::npieces] for k in range(npieces)]
    batches = [[srcs[order[i]] for i in piece] for piece in pieces]
    ring = {}

    def outs_for(b, slot):
        key = (slot, tuple(order[i] for i in pieces[b]))
        if key not in ring:
            ring[key] = dec.allocate_outputs(batches[b], "rgb")
        return ring[key]

    def one_pass():
        inflight = 0
        for b in range(len(batches)):
            dec.submit(batches[b], outs_for(b, b % 3))
            inflight += 1
            if inflight == 3:
                dec.wait()
                inflight -= 1
        while inflight:
            dec.wait()
            inflight -= 1
        torch.cuda.synchronize()

    one_pass()  # warm: arenas and output buffers
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    one_pass()
    t = sharding.max_over_ranks(time.perf_counter() - t0, dist, REDUCE_DEVICE)
    mp = sum(shapes[k % 5][0] * shapes[k % 5][1] for k in order) / 1e6
    # parity of what the timed pass wrote (after the timed region, never inside): every output of this rank against the oracle's decode of
    # its source -- ten distinct sources, one reference each
    import oracle
    refs = {}
    checked, same = 0, True
    for b in range(len(batches)):
        outs_b = outs_for(b, b % 3)
        for i, o in zip(pieces[b], outs_b):
            k = order[i]
            if k not in refs:
                refs[k] = torch.from_numpy(oracle.decode(srcs[k])).to(o.device)
            same = same and bool(torch.equal(o, refs[k]))
            checked += 1
    return {"workload": "configs[3]: batch=2048 mixed 480p-4K, 4:2:0/4:2:2 -> I_RGB, sharded over %d GPU(s) by sharding.shard_batch" % world,
            "images_per_s": round(total / t, 1), "mp_per_s": round(mp / t, 1), "images_this_rank": len(mine),
            "parity_config3": same, "parity_images_checked": checked,
            "path": "GPU entropy stage, pieces of 256, three in flight per rank; max over ranks"}


def config4_progressive(dec, host_threads):
    import torch
    from nvimagecodec_amd.synth import synth_image
    try:
        prog = [_pil_encode(synth_image(WIDTH, HEIGHT, seed=900 + k), 90, "444", progressive=True) for k in range(4)]
    except ImportError:
        return {"skipped": "no progressive encoder available on this box to make inputs"}
    batch = [prog[i % 4] for i in range(128)]
    outs = dec.allocate_outputs(batch, "rgb_planar")
    res = {"workload": "configs[4]: batch=128 1920x1080 progressive 4:4:4 -> P_RGB"}
    for label, gh in (("host_entropy", False), ("gpu_entropy", True)):
        dec.decode(batch, fmt="rgb_planar", outs=outs, gpu_huffman=gh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 1 if not gh else 3
        for _ in range(reps):
            dec.decode(batch, fmt="rgb_planar", outs=outs, gpu_huffman=gh)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps
        res[label + "_images_per_s"] = round(128 / t, 1)
        res[label + "_gpu_decoded_images"] = dec.stats()["gpu_entropy_images"]
    # the GPU path with three batches in flight (hipjpegDecodeBatchSubmit/Wait), as configs[1]'s end-to-end figure is taken: the
    # walk of a progressive scan is a sequential chain per scan, so a batch's time is the longest chain's -- batches in flight
    # fill the rest of the chip
    depth = PROGRESSIVE_DEPTH
    dec.set_pipeline_depth(depth)
    ring = [outs] + [dec.allocate_outputs(batch, "rgb_planar") for _ in range(depth - 1)]
    for k in range(depth):  # warm-up: every page in flight sizes its arenas on first use
        dec.submit(batch, ring[k], fmt="rgb_planar")
    for k in range(depth):
        dec.wait()
    torch.cuda.synchronize()
    nb = 3 * depth
    t0 = time.perf_counter()
    for i in range(nb):
        dec.submit(batch, ring[i % depth], fmt="rgb_planar")
        if i >= depth - 1:
            dec.wait()
    for _ in range(depth - 1):
        dec.wait()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / nb
    dec.set_pipeline_depth(3)
    res["gpu_entropy_pipelined_images_per_s"] = round(128 / t, 1)
    res["pipeline_depth"] = depth
    res["hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES")
    res["images_per_s"] = max(res["host_entropy_images_per_s"], res["gpu_entropy_images_per_s"], res["gpu_entropy_pipelined_images_per_s"])
    res["path"] = ("host JPEG bytes -> P_RGB in HBM; best of: host entropy stage, GPU walk + replay one batch at a time, the same with %d batches "
                   "in flight (hipjpegSetPipelineDepth)" % depth)
    res["host_threads"] = host_threads
    return res


def device_stage_flavours(dec, jpegs):
    """The pixel kernels' other flavours on the same batch (configs[1] times the everyday one: interleaved RGB, fancy upsampling):
    K1 + K2 per step by HIP events, coefficients resident in HBM.  The generic flavour of K2 runs at the same five-waves bound with
    24-40 bytes of scratch per lane -- measured faster than four waves without."""
    import torch
    res = {}
    for label, fmt, fancy in (("rgb_planar", "rgb_planar", True), ("bgr_interleaved", "bgr", True), ("rgb_plain_upsampling", "rgb", False),
                              ("gray", "y", True)):
        outs = dec.allocate_outputs(jpegs, fmt)
        dec.host_stage(jpegs, outs, fmt, fancy=fancy, gpu_huffman=True)
        dec.transfer()
        dec.device_stage(which=3)
        for _ in range(10):
            dec.device_stage(which=0)
            dec.device_stage(which=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            dec.device_stage(which=0)
            dec.device_stage(which=1)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        px = 1 if fmt == "y" else 3
        res[label] = {"k1_k2_ms": round(ms, 4), "images_per_s": round(BATCH / ms * 1e3, 1),
                      "GBps_algorithmic": round((COEF_BYTES_PER_IMAGE + WIDTH * HEIGHT * px) * BATCH / (ms * 1e-3) / 1e9, 1)}
        del outs
    return res


def encode_figures(outs, local_rank, host_threads):
    import torch
    from nvimagecodec_amd.lowlevel import BatchEncoder
    enc = BatchEncoder(device=local_rank, num_threads=host_threads)
    enc.device_stage(outs, "420", 90, "rgb")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    for _ in range(40):
        enc.relaunch()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        enc.relaunch()
    e1.record()
    torch.cuda.synchronize()
    enc_ms = e0.elapsed_time(e1) / reps
    t0 = time.perf_counter()
    enc.host_stage(gpu_huffman=False)
    t_ench = time.perf_counter() - t0
    enc.host_stage(gpu_huffman=True)
    t0 = time.perf_counter()
    enc.host_stage(gpu_huffman=True)
    t_encg = time.perf_counter() - t0
    for _ in range(3):
        enc.submit(outs, "420", 90, "rgb", gpu_huffman=True)
    for _ in range(3):
        enc.wait(fetch=False)
    enc_batches = 18
    t0 = time.perf_counter()
    for i in range(enc_batches):
        enc.submit(outs, "420", 90, "rgb", gpu_huffman=True)
        if i > 1:
            enc.wait(fetch=False)
    enc.wait(fetch=False)
    enc.wait(fetch=False)
    t_enc_e2e = (time.perf_counter() - t0) / enc_batches
    files = enc.bitstreams()   # the files of the last timed batch, fetched after the timed region
    # per-image optimized Huffman tables (nvimgcodecJpegEncodeParams_t::optimized_huffman) through the same pipeline: statistics on the
    # device, jpeg_gen_optimal_table on the host, coding on the device (round 2: the host coder, ~1.4 k images/s)
    for _ in range(3):
        enc.submit(outs, "420", 90, "rgb", gpu_huffman=True, optimized_huffman=True)
    for _ in range(3):
        enc.wait(fetch=False)
    opt_gpu_images = enc.stats()["gpu_entropy_images"]
    opt_batches = 12
    t0 = time.perf_counter()
    for i in range(opt_batches):
        enc.submit(outs, "420", 90, "rgb", gpu_huffman=True, optimized_huffman=True)
        if i > 1:
            enc.wait(fetch=False)
    enc.wait(fetch=False)
    enc.wait(fetch=False)
    t_enc_opt = (time.perf_counter() - t0) / opt_batches
    opt_files = enc.bitstreams()
    est = enc.stats()
    # parity: every file of that batch against the oracle's encoder on the same pixels (byte for byte; the inputs are the decoded
    # pictures of the decode leg, eight distinct ones)
    import oracle
    want, enc_same, enc_checked = {}, True, 0
    for i, (o, f) in enumerate(zip(outs, files)):
        k = i % 8
        if k not in want:
            want[k] = oracle.encode(o.cpu().numpy(), "420", 90)
        enc_same = enc_same and f is not None and bytes(f) == want[k]
        enc_checked += 1
    info = {"workload": "configs[2]: batch=256 1920x1080 RGB -> JPEG q90 4:2:0", "device_stage_ms": round(enc_ms, 4),
            "device_stage_images_per_s": round(BATCH / enc_ms * 1e3, 1),
            "device_stage_GBps_algorithmic": round((est["pixel_bytes"] + est["coef_bytes"]) / enc_ms / 1e6, 1),
            "host_huffman_images_per_s": round(BATCH / t_ench, 1), "host_threads": host_threads,
            "gpu_huffman_stage_ms": round(t_encg * 1e3, 3), "gpu_huffman_images_per_s": round(BATCH / t_encg, 1),
            "end_to_end_images_per_s": round(BATCH / t_enc_e2e, 1),
            "parity_encode": enc_same, "parity_files_checked": enc_checked,
            "optimized_huffman_end_to_end_images_per_s": round(BATCH / t_enc_opt, 1), "optimized_huffman_gpu_coded_images": opt_gpu_images,
            "optimized_huffman_bytes_vs_standard": round(sum(len(f) for f in opt_files) / max(1, sum(len(f) for f in files)), 4),
            "end_to_end_includes": "RGB in HBM -> JPEG files in host memory: forward kernel + GPU entropy coder, files written to pinned host "
                                   "memory by the last kernel; three batches in flight (hipjpegEncodeBatchSubmit/Wait)"}
    enc.close()
    return info


# ------------------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernels-only", action="store_true",
                    help="setup + warm-up + timed steps and nothing else: the run to put under rocprofv3 --kernel-trace --stats / --pmc")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if REHEARSAL else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    distributed = world > 1

    # torch's libraries, then ours, then the first HIP call: the library must be in the process before the runtime initialises (it asks
    # for more hardware queues, see the top) and behind torch's own copy of the HIP runtime (nvimagecodec_amd/_native.py load)
    import torch
    from nvimagecodec_amd import _native
    _native.load()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decode path has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    from nvimagecodec_amd import sharding
    # each rank's host threads next to its GPU: the threads created from here on inherit the mask
    numa = sharding.pin_to_device_numa(local_rank, world)
    dist = None
    if distributed:
        import torch.distributed as dist
        if REHEARSAL:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from nvimagecodec_amd.lowlevel import BatchDecoder
    from nvimagecodec_amd.sharding import max_over_ranks
    sources, data_desc = make_inputs()
    jpegs = [sources[i % len(sources)] for i in range(BATCH)]
    host_threads = max(1, usable_cpus() // max(world if numa.get("pinned") is None else 1, 1))
    dec = BatchDecoder(device=local_rank, num_threads=host_threads)
    outs = dec.allocate_outputs(jpegs, "rgb")

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- setup: bitstreams into HBM (header parse + staging + one H2D copy).  From here on a step touches no host data.
    dec.host_stage(jpegs, outs, "rgb", fancy=True, gpu_huffman=True)
    dec.transfer()
    torch.cuda.synchronize()
    gst = dec.stats()
    # configs[1] = YCbCr 4:2:0 -> interleaved RGB, fancy upsampling: the everyday interleaved build of the fused luma kernel (layout 1)
    k1_name = "idct_plane_kernel"
    k2_name = "luma_color_kernel<2, 2, 1>"
    assert gst["gpu_entropy_images"] == BATCH, "the bench batch must take the GPU entropy stage"

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        dec.device_stage(which=6)   # GPU entropy stage: destuff, synchronise, scan, positions, blocks, DC
        if ev is not None:
            ev[1].record()
        dec.device_stage(which=0)   # idct_plane_kernel   (chroma blocks -> planes)
        if ev is not None:
            ev[2].record()
        dec.device_stage(which=1)   # luma_color_kernel   (luma IDCT + upsample + colour + store)
        if ev is not None:
            ev[3].record()

    def timed(steps):
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            step(ev[k])
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0, dist, REDUCE_DEVICE)
        ms = [sum(ev[k][i].elapsed_time(ev[k][i + 1]) for k in range(steps)) / steps for i in range(3)]
        return elapsed, ms

    # ---- the contract's protocol: W untimed warm-up steps, then exactly K timed steps
    for _ in range(args.warmup):
        step()
    elapsed, (ent_ms, k1_ms, k2_ms) = timed(args.steps)
    statuses = dec.statuses(BATCH)   # settles the GPU entropy stage's verdicts of the last step
    assert all(s == 0 for s in statuses), statuses

    # ---- the same protocol again with the clocks settled (reported beside `value`)
    for _ in range(STEADY_PREWARM_STEPS):
        step()
    elapsed_s, (ent_s, k1_s, k2_s) = timed(args.steps)
    assert all(s == 0 for s in dec.statuses(BATCH))

    # ---- parity of what the timed steps wrote: EVERY output against the oracle (the checker, never the thing measured)
    parity, parity_n = None, 0
    if rank == 0:
        import oracle
        refs = [torch.from_numpy(oracle.decode(s)).cuda() for s in sources]
        parity = all(torch.equal(o, refs[i % len(sources)]) for i, o in enumerate(outs))
        parity_n = len(outs)
        del refs

    extras = {}
    if not args.kernels_only:
        # ---- end to end from HOST bytes (PCIe inclusive; never `value`)
        # (a) pipelined GPU-entropy path: header parse + staging + H2D of the bitstreams + every kernel, three batches in flight
        ring = [outs, dec.allocate_outputs(jpegs, "rgb"), dec.allocate_outputs(jpegs, "rgb")]
        pipe_batches = 96  # (the first Wait holds the whole latency of batch 0, 9 ms: 3 % of the loop)
        for k in range(3):  # warm-up: every one of the three pages sizes its pinned and device arenas on first use (11-15 ms each)
            dec.submit(jpegs, ring[k])
        for k in range(3):
            dec.wait()
        barrier()
        t0 = time.perf_counter()
        for i in range(pipe_batches):
            dec.submit(jpegs, ring[i % 3])
            if i > 1:
                dec.wait()
        dec.wait()
        dec.wait()
        torch.cuda.synchronize()
        t_e2e_gpu = max_over_ranks((time.perf_counter() - t0) / pipe_batches, dist, REDUCE_DEVICE)
        # (a') the same loop with the files in page-locked memory: the copy engine reads each scan from the caller's buffer, the
        # staging copy on the host is gone (zero-copy input; one pass over host DRAM per byte instead of three)
        pinned = {id(s): torch.frombuffer(bytearray(s), dtype=torch.uint8).pin_memory() for s in sources}
        jpegs_pinned = [pinned[id(j)] for j in jpegs]
        for k in range(3):
            dec.submit(jpegs_pinned, ring[k])
        for k in range(3):
            dec.wait()
        zero_copy_images = dec.stats()["zero_copy_images"]
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for i in range(pipe_batches):
            dec.submit(jpegs_pinned, ring[i % 3])
            if i > 1:
                dec.wait()
        dec.wait()
        dec.wait()
        torch.cuda.synchronize()
        t_e2e_zc = max_over_ranks((time.perf_counter() - t0) / pipe_batches, dist, REDUCE_DEVICE)
        stream_bytes_per_batch = dec.stats()["stream_bytes"]
        del ring, jpegs_pinned, pinned
        # (b) the north-star split: Huffman on the host cores, coefficients over PCIe, device stage -- through the same Submit/Wait
        # pipeline, three batches in flight (the host stage of batch k+1 runs while batch k's coefficients cross PCIe and batch k-1's
        # kernels run); every page warmed first; per-iteration times kept so that the line carries the median and the spread
        ring = [dec.allocate_outputs(jpegs, "rgb") for _ in range(3)]
        for i in range(3):
            dec.submit(jpegs, ring[i], gpu_huffman=False)
        for _ in range(3):
            dec.wait()
        torch.cuda.synchronize()
        cpu_batches = 12
        marks = [time.perf_counter()]
        for i in range(cpu_batches):
            dec.submit(jpegs, ring[i % 3], gpu_huffman=False)
            if i > 1:
                dec.wait()
                marks.append(time.perf_counter())
        dec.wait()
        marks.append(time.perf_counter())
        dec.wait()
        torch.cuda.synchronize()
        marks.append(time.perf_counter())
        t_e2e_cpu = max_over_ranks((marks[-1] - marks[0]) / cpu_batches, dist, REDUCE_DEVICE)
        cpu_iters = sorted(b - a for a, b in zip(marks[1:-1], marks[2:]))  # steady-state iterations (the first holds the pipeline's fill)
        del ring
        # host entropy stage and H2D alone
        t0 = time.perf_counter()
        dec.host_stage(jpegs, outs, "rgb", fancy=True)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dec.transfer()
        torch.cuda.synchronize()
        t_h2d = time.perf_counter() - t0
        hst = dec.stats()
        extras["end_to_end"] = {
            "images_per_s": round(BATCH * world / t_e2e_gpu, 1), "mp_per_s": round(BATCH * world / t_e2e_gpu * WIDTH * HEIGHT / 1e6, 1),
            "includes": "host JPEG bytes -> RGB in HBM: header parse + H2D of the bitstreams + GPU entropy stage + device stage, "
                        "three batches in flight (hipjpegDecodeBatchSubmit/Wait)",
            "zero_copy_input": {"images_per_s": round(BATCH * world / t_e2e_zc, 1), "zero_copy_images_per_batch": zero_copy_images,
                                "note": "the same loop with the files in pinned host memory: no staging copy, the copy engine reads the caller's buffers"},
            # what each rank asks of the host's memory system (eight ranks share it): the bitstream bytes it consumes per
            # second, times the passes over DRAM per byte -- read + non-temporal store into the staging area + the copy engine's read = 3
            # with pageable inputs, 1 with pinned inputs
            "host_bitstream_GBps_per_rank": round(stream_bytes_per_batch / t_e2e_gpu / 1e9, 2),
            "host_dram_GBps_per_rank": {"pageable_inputs_3_passes": round(3 * stream_bytes_per_batch / t_e2e_gpu / 1e9, 2),
                                        "pinned_inputs_1_pass": round(stream_bytes_per_batch / t_e2e_zc / 1e9, 2)},
            "cpu_huffman_images_per_s": round(BATCH * world / t_e2e_cpu, 1),
            "cpu_huffman_includes": "the north-star split: Huffman on the host cores + H2D of the coefficients + device stage, through "
                                    "hipjpegDecodeBatchSubmit/Wait with three batches in flight, %d timed batches after every page was warmed" % cpu_batches,
            "cpu_huffman_iteration_ms": {"median": round(cpu_iters[len(cpu_iters) // 2] * 1e3, 2), "p10": round(cpu_iters[len(cpu_iters) // 10] * 1e3, 2),
                                         "max": round(cpu_iters[-1] * 1e3, 2), "n": len(cpu_iters)},
            "host_threads_per_gpu": host_threads}
        extras["host_stage"] = {"huffman_images_per_s": round(BATCH / t_host, 1), "threads": host_threads,
                                "h2d_GBps": round(hst["h2d_bytes"] / t_h2d / 1e9, 1), "h2d_ms": round(t_h2d * 1e3, 2),
                                # zero-run-compressed staging: what the host entropy stage puts on PCIe against the dense int16 blocks
                                "h2d_bytes_per_batch": hst["h2d_bytes"], "dense_coefficient_bytes_per_batch": hst["coef_bytes"],
                                "sparse_images": hst["sparse_images"]}
        if rank == 0:
            try:
                extras["device_stage_flavours"] = device_stage_flavours(dec, jpegs)
            except Exception as e:
                extras["device_stage_flavours"] = {"error": repr(e)}
        try:
            extras["encode"] = encode_figures(outs, local_rank, host_threads)
        except Exception as e:  # the decode line must not be lost because an extra failed
            extras["encode"] = {"error": repr(e)}
        others = {}
        try:
            others["config3_mixed_shapes"] = config3_sharded(dec, rank, world, dist, host_threads)
        except Exception as e:
            others["config3_mixed_shapes"] = {"error": repr(e)}
        if rank == 0:
            try:
                others["config4_progressive_444"] = config4_progressive(dec, host_threads)
            except Exception as e:
                others["config4_progressive_444"] = {"error": repr(e)}
        extras["other_configs"] = others

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = BATCH * world * args.steps / elapsed
        alg_bytes = ALG_BYTES_PER_IMAGE * BATCH
        ent_bytes = gst["stream_bytes"] + COEF_BYTES_PER_IMAGE * BATCH

        def roof(bytes_, ms):
            a = bytes_ / (ms * 1e-3) / 1e9
            return round(a, 1), round(a / HBM_PEAK_GBS, 4)

        ach, frac = roof(alg_bytes, k1_ms + k2_ms)
        ach_s, frac_s = roof(alg_bytes, k1_s + k2_s)
        each, efrac = roof(ent_bytes, ent_ms)
        line = {
            "metric": "images/sec, batched 1920x1080 4:2:0 baseline JPEG decode to interleaved RGB u8",
            "value": round(value, 1), "unit": "images/s", "mp_per_s": round(value * WIDTH * HEIGHT / 1e6, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": data_desc,
            "timed_region": "full decode, bitstreams resident in HBM -> RGB in HBM: GPU entropy stage (byte-stuffing removal + Huffman decode) "
                            "+ device stage (dequantize, IDCT, upsample, colour, store); host bytes -> HBM is reported under end_to_end",
            "config": {"workload": "configs[1]: batch=256 1920x1080 4:2:0 baseline JPEG -> I_RGB u8, fancy upsampling, ISLOW IDCT",
                       "batch_per_gpu": BATCH,
                       "parallelism": f"{world} independent per-GPU replicas, no collective" +
                                      (" -- REHEARSAL: all ranks share GPU 0 (BENCH_REHEARSE_ON_ONE_GPU), figures are not measurements" if REHEARSAL else ""),
                       "host_threads_per_gpu": host_threads, "numa": numa},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac, "traffic": measured_traffic(),
                         "scope": "device stage (SURVEY 8d): idct_plane_kernel + luma_color_kernel, HIP events inside the timed steps",
                         "algorithmic_bytes_per_step": alg_bytes,
                         "copy_rate": HBM_COPY_GBS, "frac_of_copy_rate": round(ach / HBM_COPY_GBS, 4),
                         "kernels": [{"name": k1_name, "avg_ms": round(k1_ms, 4), "workgroups": gst["units"][0]},
                                     {"name": k2_name, "avg_ms": round(k2_ms, 4), "workgroups": gst["units"][1]}]},
            "entropy_stage": {"avg_ms": round(ent_ms, 4), "algorithmic_bytes_per_step": ent_bytes, "achieved_GBps": each, "frac_of_hbm_peak": efrac,
                              "bitstream_bytes_per_batch": gst["stream_bytes"],
                              "note": "destuff + self-synchronising Huffman decode + block write + DC scan; latency bound, not bandwidth bound"},
            "steady_state": {"prewarm_steps": STEADY_PREWARM_STEPS, "images_per_s": round(BATCH * world * args.steps / elapsed_s, 1),
                             "ms_per_step": round(elapsed_s / args.steps * 1e3, 4), "entropy_ms": round(ent_s, 4),
                             "roofline_achieved": ach_s, "roofline_frac": frac_s,
                             "kernels_ms": [round(k1_s, 4), round(k2_s, 4)],
                             "note": "the same K timed steps after a further untimed run: GPU clocks settled"},
            "parity_vs_oracle": parity, "parity_images_checked": parity_n,
        }
        line.update(extras)
        if world == 1 and not args.no_cpu_baseline and not args.kernels_only:
            line["cpu_baseline"] = cpu_baseline(sources)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
