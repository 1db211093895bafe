import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from nvimagecodec_amd.lowlevel import BatchDecoder
dec = BatchDecoder(0, bench.usable_cpus())
t0 = time.perf_counter()
print(bench.config3_sharded(dec, 0, 1, None, bench.usable_cpus()))
print("total", time.perf_counter() - t0, dec.stats())
