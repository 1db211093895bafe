#!/bin/bash
# usage (on the GPU box): tools/enc_stats.sh [subsampling] -- encode kernel durations (rocprofv3 kernel trace) and VALU instruction counts (PMC, separate run)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SUB=${1:-420}
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_enc -o enc --output-format csv -- python3 $R/tools/prof_enc.py 6 $SUB > $R/gpurun_out/prof_enc.log 2>&1 || { tail -5 $R/gpurun_out/prof_enc.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/prof_enc_pmc -o enc -- python3 $R/tools/prof_enc.py 2 $SUB > $R/gpurun_out/prof_enc_pmc.log 2>&1 || { tail -5 $R/gpurun_out/prof_enc_pmc.log; exit 1; }
python3 - <<PY
import csv, glob, collections
for r in csv.DictReader(open("$R/gpurun_out/prof_enc/enc_kernel_stats.csv")):
    n = r['Name'].replace('(anonymous namespace)::', '')
    if 'kernel' in n: print("%-50s calls %s avg %.1f us min %.1f" % (n[:50], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/prof_enc_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].replace("hipjpeg::(anonymous namespace)::", "")[:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if 'SQ_WAVES' in d and 'forward' in k:
        w = sum(d['SQ_WAVES']) / len(d['SQ_WAVES'])
        print(k, "waves %.0f" % w, " ".join("%s/wave=%.0f" % (c, sum(v) / len(v) / w) for c, v in sorted(d.items()) if c != 'SQ_WAVES'))
PY
