#!/usr/bin/env python3
"""Registers, scratch and static VALU instruction counts per kernel of a device-only assembly listing
(hipcc -O3 --offload-arch=gfx950 -S --cuda-device-only file.hip -o file.s).  Usage: kernel_resources.py file.s"""
import re
import sys

s = open(sys.argv[1]).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    v = re.search(r'\.amdhsa_next_free_vgpr (\d+)', body).group(1)
    sg = re.search(r'\.amdhsa_next_free_sgpr (\d+)', body).group(1)
    sc = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', body).group(1)
    fm = re.search('^' + re.escape(name) + r':[^\n]*\n(.*?)\.Lfunc_end', s, re.S | re.M)
    n = len(re.findall(r'^\s+v_', fm.group(1), re.M)) if fm else -1
    print(f"{name[:90]:90s} vgpr {v:>4s} sgpr {sg:>4s} scratch {sc:>4s} valu_static {n}")
