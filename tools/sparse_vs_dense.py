"""Dev tool: K1 / K2 of configs[1] on host-decoded coefficients staged DENSE (HIPJPEG_DENSE_STAGING=1) or SPARSE (default: zero-run-compressed
records expanded by the kernels' LDS fetch) -- what the pixel kernels would gain if the GPU entropy stage wrote the sparse format too.
usage (GPU box): python tools/sparse_vs_dense.py   (runs itself twice, once per staging)"""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1:
    import torch
    import bench
    from nvimagecodec_amd.lowlevel import BatchDecoder
    src, _ = bench.make_inputs()
    jpegs = [src[i % len(src)] for i in range(bench.BATCH)]
    dec = BatchDecoder(0, bench.usable_cpus())
    outs = dec.allocate_outputs(jpegs)
    dec.host_stage(jpegs, outs, "rgb", fancy=True, gpu_huffman=False)
    dec.transfer()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for _ in range(5):
        dec.device_stage(which=0)
        dec.device_stage(which=1)
    torch.cuda.synchronize()
    k1 = k2 = 0.0
    n = 30
    for _ in range(n):
        ev[0].record()
        dec.device_stage(which=0)
        ev[1].record()
        dec.device_stage(which=1)
        ev[2].record()
        torch.cuda.synchronize()
        k1 += ev[0].elapsed_time(ev[1])
        k2 += ev[1].elapsed_time(ev[2])
    st = dec.stats()
    print("%s staging: K1 %.3f ms, K2 %.3f ms; sparse images %s, H2D bytes %s" % (sys.argv[1], k1 / n, k2 / n, st.get("sparse_images"), st.get("h2d_bytes")))
else:
    for name, env in (("dense", {"HIPJPEG_DENSE_STAGING": "1"}), ("sparse", {})):
        subprocess.run([sys.executable, os.path.abspath(__file__), name], env=dict(os.environ, **env), check=True)
