#!/bin/bash
# usage (on the GPU box, after tools/build_variant.sh walkprof "-DHJ_WALK_PROFILE=1" progressive_gpu.hip here): tools/walk_laps.sh [variant name]
# (a variant built with -DHJ_WALK_PROFILE=2 books the fast loops alone: visits that took 0 / 1 / 2 / more symbols, shown as fast / event / window / block)
# Where a walking wave's time goes (BASELINE configs[4], one batch at a time): lap timers around the fast loops, the event handling, the
# window switches and the block starts of prog_walk_scan, per scan of image 0.  The stamps wait for whatever LDS traffic is in flight, so the
# shares are an upper bound for the parts that issue it (window, block).
R=${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
HIPJPEG_LIB_PATH=$R/nvimagecodec_amd/variants/lib_${1:-walkprof}.so HIPJPEG_DEBUG_TIMING=1 HIPJPEG_WALK_LAPS=1 python3 $R/tools/prof_prog_pipe.py 2 1 2>&1 | grep -i "scan [0-9]\|laps\|pipelined" | tail -21 | cut -c1-170
