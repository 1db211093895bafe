"""Profiling driver (dev tool): BASELINE configs[3] alone (2048 mixed 480p-4K images, 4:2:0 / 4:2:2, pieces of 256, three in flight).
Run under rocprofv3 --kernel-trace --stats for the per-kernel split, or plainly for the rate."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nvimagecodec_amd.lowlevel import BatchDecoder
dec = BatchDecoder(0, bench.usable_cpus())
print(bench.config3_sharded(dec, 0, 1, None, bench.usable_cpus()))
