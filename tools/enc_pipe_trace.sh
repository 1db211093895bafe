#!/bin/bash
# usage (on the GPU box): tools/enc_pipe_trace.sh -- timeline of the pipelined encode (kernels + copies)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/prof_encpipe -o ep --output-format csv -- python3 $R/tools/prof_enc_pipe.py > $R/gpurun_out/prof_encpipe.log 2>&1 || { tail -5 $R/gpurun_out/prof_encpipe.log; exit 1; }
tail -2 $R/gpurun_out/prof_encpipe.log
python3 - <<PY
import csv
ev = []
for r in csv.DictReader(open("$R/gpurun_out/prof_encpipe/ep_kernel_trace.csv")):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].split('::')[-1][:22]
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n, r.get('Stream_Id', r.get('Queue_Id', '?'))))
for r in csv.DictReader(open("$R/gpurun_out/prof_encpipe/ep_memory_copy_trace.csv")):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')[-12:], r.get('Stream_Id', '?')))
ev.sort()
t0 = ev[0][0]
# the last ~3 batches
last = [e for e in ev if e[0] > ev[-1][1] - 16e6]
for s, e, n, q in last:
    if e - s > 20000: print("%9.3f ms  +%7.3f ms  %-24s q=%s" % ((s - t0) / 1e6, (e - s) / 1e6, n, q))
PY
