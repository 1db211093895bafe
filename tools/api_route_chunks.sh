#!/bin/bash
# usage (on the GPU box): tools/api_route_chunks.sh  -- the nvimgcodecDecoderDecode route with 1..3 calls outstanding, for the default
# (adaptive) and forced values of the plugin option pipeline_chunks
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
for opt in ""; do
  for p in 1 2 3 4 6; do
    echo "== options '$opt' -p $p: $($R/nvimagecodec_amd/hipimtrans -i $D -b 256 -w 8 -r 24 -p $p --skip_encode --options "$opt" | grep -E "decoding speed")"
  done
done
