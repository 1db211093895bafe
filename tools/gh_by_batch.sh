#!/bin/bash
# usage (on the GPU box): tools/gh_by_batch.sh [batch sizes...]   -- per-kernel times of the GPU entropy stage for several batch sizes
# (which kernels are bound by residency -- time proportional to the batch -- and which by the latency of their longest chain)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in ${@:-256 128 64}; do
  echo "== batch $B"
  rm -rf $R/gpurun_out/prof_gb
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gb -o gh --output-format csv -- python3 $R/tools/prof_gh.py 4 $B > $R/gpurun_out/prof_gb.log 2>&1) || { tail -5 $R/gpurun_out/prof_gb.log; exit 1; }
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof_gb/gh_kernel_trace.csv")))
seq = []
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].split('::')[-1][:28]
    seq.append((n, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
last = [i for i, (n, _) in enumerate(seq) if n.startswith('destuff_count')][-1]
print("  ".join("%s %.0f" % (n.replace('_kernel', ''), us) for n, us in seq[last:] if not n.startswith('__amd')))
PY
done
