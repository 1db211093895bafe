import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from nvimagecodec_amd import api
src, _ = bench.make_inputs()
jpegs = [src[i % len(src)] for i in range(256)]
with api.Decoder(max_num_cpu_threads=bench.usable_cpus()) as dec:
    dec.decode(jpegs); torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3):
        dec.decode(jpegs); torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
