#!/bin/bash
# dev tool: rebuild the decode kernels with different occupancy targets and time the device stage
cd $GRAFT_REPO_ROOT
for w in 5 6; do
  rm -f nvimagecodec_amd/csrc/build/decode_kernels.o
  make -C nvimagecodec_amd/csrc -j8 EXTRA_FLAGS="-DHJ_MIN_WAVES=$w" > /dev/null 2>&1
  echo "== min waves/SIMD $w"
  timeout -k 10 200 python tests/devtools/quick_time.py 256 2>&1 | grep -E "device stage|parity" | tail -2
done
