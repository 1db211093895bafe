#!/bin/bash
# usage (on the GPU box): tools/ab_full_flag.sh "<flags A>" "<flags B>" ...  -- rebuild the WHOLE library with each set of -D flags (for
# constants that host and device code share, e.g. -DHJ_MCUS_PER_WG=256) and print the bench line's step time, roofline kernels and
# entropy stage; the shipped build is restored at the end
cd $GRAFT_REPO_ROOT
for f in "$@" ""; do
  rm -f nvimagecodec_amd/csrc/build/*.o
  make -C nvimagecodec_amd/csrc -j16 EXTRA_FLAGS="$f" > /dev/null 2>&1
  [ -z "$f" ] && echo "== shipped build" || echo "== $f"
  for i in 1 2; do
  python3 bench.py --kernels-only 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  K1 %.4f  K2 %.4f  entropy %.4f  parity %s' % (d['ms_per_step'], d['roofline']['kernels'][0]['avg_ms'], d['roofline']['kernels'][1]['avg_ms'], d['entropy_stage']['avg_ms'], d['parity_vs_oracle']))"
  done
done
