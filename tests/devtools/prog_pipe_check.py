"""Dev tool: configs[4] through the pipelined entry points (three batches in flight)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from nvimagecodec_amd.lowlevel import BatchDecoder
dec = BatchDecoder(0, bench.usable_cpus())
print(bench.config4_progressive(dec, bench.usable_cpus()))
