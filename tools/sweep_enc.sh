#!/bin/bash
# dev tool: rebuild the encode kernels with different occupancy targets and time the forward kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in ${SWEEP_WAVES:-3 4}; do
  rm -f nvimagecodec_amd/csrc/build/encode_kernels.o
  make -C nvimagecodec_amd/csrc -j8 EXTRA_FLAGS="-DHJ_PAIR_WAVES=$w" > /dev/null 2>&1
  echo "== min waves/SIMD $w"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_enc_w$w -o enc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof_enc.py 4 > /dev/null 2>&1)
  grep forward gpurun_out/prof_enc_w$w/enc_kernel_stats.csv | awk -F, '{print $(NF-5), $(NF-4), "avg ns", $(NF-4)}' | head -2
done
