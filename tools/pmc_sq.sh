#!/bin/bash
# usage: tools/pmc_sq.sh <outdir-under-gpurun_out> <script> [args...]   -- SQ/LDS counter groups only (two passes)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/"$@" > $OUT/g$i.log 2>&1 || { echo "group $i failed"; tail -5 $OUT/g$i.log; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].replace("hipjpeg::(anonymous namespace)::", "")[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s n=%d mean=%.4g  [%s]" % (c, len(v), sum(v) / len(v), " ".join("%.3g" % x for x in v[:4])))
PY
