#!/bin/bash
# usage (on the GPU box): tools/prog_pipe_trace.sh [batches [depth]]   -- kernel timeline of the pipelined progressive decode (who overlaps whom)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_pp
(cd /tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/prof_pp -o pp --output-format csv -- python3 $R/tools/prof_prog_pipe.py ${1:-6} ${2:-3} > $R/gpurun_out/prof_pp.log 2>&1) || { tail -5 $R/gpurun_out/prof_pp.log; exit 1; }
grep -v rocprof $R/gpurun_out/prof_pp.log | tail -${1:-6} ; tail -2 $R/gpurun_out/prof_pp.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof_pp/pp_kernel_trace.csv")))
ev = []
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].split('::')[-1][:24]
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n, r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
ev.sort()
walks = [e for e in ev if e[2].startswith('prog_walk')]
t0 = walks[1][0] if len(walks) > 1 else ev[0][0]
for s, e, n, q, st in ev:
    if s < t0 or (e - s) < 200000: continue  # from the second batch on; kernels of 0.2 ms and more
    print("%9.2f .. %9.2f ms  %7.2f ms  q%-3s s%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, st, n))
PY
