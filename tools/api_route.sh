#!/bin/bash
# usage (on the GPU box): tools/api_route.sh   -- the nvimgcodecDecoderDecode route (public C API -> priority chain -> hipjpeg_decoder
# plugin) on 256 x 1080p 4:2:0 q90 files, no Python in the loop: example/hipimtrans.cpp in decode-only mode, per-stage rates.
set -e
R=$GRAFT_REPO_ROOT
D=/tmp/api_route_inputs
mkdir -p $D
python3 - <<PY
import sys
sys.path.insert(0, "$R")
import bench
src, _ = bench.make_inputs()
for i in range(256):
    open("$D/img%03d.jpg" % i, "wb").write(src[i % len(src)])
PY
$R/nvimagecodec_amd/hipimtrans -i $D -b 256 -w 2 -r 8 | grep -E "Total images|speed|per batch"
# several batches in flight (nvimgcodecDecoderDecode returns a future; the caller submits the next batch before it waits)
for p in 2 3; do
  $R/nvimagecodec_amd/hipimtrans -i $D -b 256 -w 3 -r 16 -p $p | grep -E "Total images|speed"
done
