#!/bin/bash
# dev tool: timing-only builds with one memory stream removed each (outputs are wrong by construction)
cd $GRAFT_REPO_ROOT
for f in "" "-DHJ_ABLATE_STORE" "-DHJ_ABLATE_CHROMA" "-DHJ_ABLATE_COEF" "-DHJ_ABLATE_STORE -DHJ_ABLATE_CHROMA -DHJ_ABLATE_COEF"; do
  rm -f nvimagecodec_amd/csrc/build/decode_kernels.o
  make -C nvimagecodec_amd/csrc -j8 EXTRA_FLAGS="$f" > /dev/null 2>&1
  echo "== flags: $f"
  timeout -k 10 200 python tests/devtools/quick_time.py 256 2>&1 | grep -E "device stage" | tail -1
done
