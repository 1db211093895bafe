#!/bin/bash
# usage (on the GPU box): tools/run_campaigns.sh <tag> [scale]  -- the randomized cross-check campaigns of tests/campaigns at length, one after the
# other, summary lines into gpurun_out/<tag>_campaigns.txt (copy to profiles/).  scale multiplies the round counts (default 1 ~ 6 minutes).
R=$GRAFT_REPO_ROOT
TAG=$1
S=${2:-1}
cd $R
export HIPJPEG_ENABLE_TEST_HOOKS=1
OUT=gpurun_out/${TAG}_campaigns.txt
: > $OUT
run() { echo "== $*" | tee -a $OUT; timeout -k 10 900 python3 "$@" 2>&1 | tail -2 | tee -a $OUT; }
run tests/campaigns/fuzz_pass1.py 31 $((40 * S))
run tests/campaigns/fuzz_gpu.py 32 $((40 * S))
run tests/campaigns/fuzz_damage.py 33 $((60 * S))
run tests/campaigns/fuzz_progressive.py 41 $((40 * S))
run tests/campaigns/fuzz_encode.py 34 $((60 * S))
run tests/campaigns/fuzz_outputs.py 35 $((20 * S))
run tests/campaigns/fuzz_geometry.py 36 $((20 * S))
run tests/campaigns/fuzz_plugin.py 37 $((30 * S))
HIPJPEG_FUSED_DECODE=1 run tests/campaigns/fuzz_gpu.py 38 $((20 * S))
HIPJPEG_FUSED_DECODE=1 run tests/campaigns/fuzz_damage.py 39 $((30 * S))
HIPJPEG_DENSE_STAGING=1 run tests/campaigns/fuzz_gpu.py 40 $((10 * S))
echo "campaigns done" | tee -a $OUT
