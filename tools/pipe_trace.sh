#!/bin/bash
# usage (on the GPU box): tools/pipe_trace.sh   -- timeline of the pipelined end-to-end decode (configs[1], three batches in flight):
# kernels and host-to-device copies of the last batches, so that what waits for what can be read off
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_pl
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/prof_pl -o pl --output-format csv -- python3 $R/tools/prof_pipe.py > $R/gpurun_out/prof_pl.log 2>&1) || { tail -5 $R/gpurun_out/prof_pl.log; exit 1; }
grep pipelined $R/gpurun_out/prof_pl.log
python3 - <<PY
import csv, glob
ev = []
for r in csv.DictReader(open("$R/gpurun_out/prof_pl/pl_kernel_trace.csv")):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].split('::')[-1][:22]
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n, 'q' + r.get('Queue_Id', '?')))
for f in glob.glob("$R/gpurun_out/prof_pl/pl_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')[-12:], ''))
ev.sort()
d = [e for e in ev if e[2].startswith('destuff_count')]
t0 = d[-3][0]
for s, e, n, q in ev:
    if s < t0 or (e - s) < 20000: continue
    print("%8.3f .. %8.3f ms  %7.3f  %-4s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n))
PY
