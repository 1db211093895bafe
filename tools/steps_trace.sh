#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_two
(cd /tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/prof_two -o two --output-format csv -- python3 $R/tools/steps_in_flight.py 12 2 > $R/gpurun_out/prof_two.log 2>&1) || { tail -5 $R/gpurun_out/prof_two.log; exit 1; }
tail -1 $R/gpurun_out/prof_two.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof_two/two_kernel_trace.csv")))
ev = []
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].split('::')[-1][:22]
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n, r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
ev.sort()
# the last 3 steps' worth of kernels
syncs = [e for e in ev if e[2].startswith('huff_sync')]
t0 = syncs[-5][0]
for s, e, n, q, st in ev:
    if s < t0: continue
    print("%8.3f .. %8.3f ms %7.3f q%-2s s%-2s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, st, n))
PY
