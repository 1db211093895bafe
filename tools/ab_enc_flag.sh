#!/bin/bash
# usage (on the GPU box): tools/ab_enc_flag.sh "<flags A>" "<flags B>" ...  -- rebuild the encode kernels with each set of -D flags and time
# the forward kernel (rocprofv3 kernel trace of tools/prof_enc.py); the shipped build is restored at the end
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for f in "$@" ""; do
  rm -f nvimagecodec_amd/csrc/build/encode_kernels.o
  make -C nvimagecodec_amd/csrc -j8 EXTRA_FLAGS="$f" > /dev/null 2>&1
  [ -z "$f" ] && [ $# -gt 0 ] && { echo "== shipped build"; }
  [ -n "$f" ] && echo "== $f"
  rm -rf gpurun_out/prof_enc_ab
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_enc_ab -o enc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/prof_enc.py 6 > /dev/null 2>&1)
  python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/prof_enc_ab/enc_kernel_stats.csv')):
    if 'forward' in r['Name']: print('forward kernel: avg %.1f us, min %.1f us over %s calls' % (float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, r['Calls']))"
done
