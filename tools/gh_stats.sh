#!/bin/bash
# usage (on the GPU box): tools/gh_stats.sh  -- per-kernel times of the GPU entropy stage (rocprofv3 kernel trace)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gh -o gh --output-format csv -- python3 $R/tools/prof_gh.py 3 > $R/gpurun_out/prof_gh.log 2>&1 || { tail -5 $R/gpurun_out/prof_gh.log; exit 1; }
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof_gh/gh_kernel_trace.csv")))
seq = []
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].split('::')[-1][:28]
    seq.append((n, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
# last repetition
last = [i for i, (n, _) in enumerate(seq) if n.startswith('destuff_count')][-1]
tot = 0
for n, us in seq[last - 1:]:
    print("%-30s %9.1f us" % (n, us)); tot += us
print("total %.1f us" % tot)
PY
