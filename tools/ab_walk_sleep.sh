#!/bin/bash
R=$GRAFT_REPO_ROOT
for lib in "" sleep8 sleep32 sleep100; do
  if [ -z "$lib" ]; then unset HIPJPEG_LIB_PATH; else export HIPJPEG_LIB_PATH=$R/nvimagecodec_amd/variants/lib_$lib.so; fi
  echo "== ${lib:-shipped (sleep 2)}: $(python3 $R/tools/prof_prog_pipe.py 6 1 | tail -1) | $(python3 $R/tools/prof_prog_pipe.py 24 6 | tail -1)"
done
