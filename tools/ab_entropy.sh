#!/bin/bash
# usage (on the GPU box): tools/ab_entropy.sh <other .so> ...   -- per-kernel times of the GPU entropy stage, shipped build vs other builds
# (tools/build_variant.sh makes them)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "" "$@"; do
  export HIPJPEG_LIB_PATH=$lib
  [ -z "$lib" ] && unset HIPJPEG_LIB_PATH
  echo "== ${lib:-shipped build}"
  rm -rf $R/gpurun_out/prof_ab
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ab -o gh --output-format csv -- python3 $R/tools/prof_gh.py 6 > $R/gpurun_out/prof_ab.log 2>&1) || { tail -5 $R/gpurun_out/prof_ab.log; exit 1; }
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof_ab/gh_kernel_trace.csv")))
seq = []
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].split('::')[-1][:28]
    seq.append((n, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
last = [i for i, (n, _) in enumerate(seq) if n.startswith('destuff_c')][-1]
tot = 0
for n, us in seq[last:]:
    print("%-30s %9.1f us" % (n, us)); tot += us
print("total %.1f us" % tot)
PY
done
