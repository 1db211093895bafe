#!/bin/bash
# usage (on the GPU box): tools/d2h_engine_probe.sh  -- which engine carries the device-to-host copy of the staged encode output
# (HIPJPEG_ENCODE_STAGED_OUTPUT=1) under the runtime's copy knobs: pipelined encode rate untraced, then the number and duration of the
# runtime's blit-copy kernels and of the SDMA copies in a traced run
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export HIPJPEG_ENCODE_STAGED_OUTPUT=1
for v in "X=0" "GPU_BLIT_ENGINE_TYPE=1" "GPU_BLIT_ENGINE_TYPE=2" "GPU_FORCE_BLIT_COPY_SIZE=0" "DEBUG_CLR_LIMIT_BLIT_WG=4" "HSA_ENABLE_SDMA=1" "ROC_ENABLE_LARGE_BAR=0"; do
  echo "== $v"
  env $v python3 $R/tools/prof_enc_pipe.py 20 2>&1 | tail -1
  rm -rf $R/gpurun_out/prof_probe
  (cd /tmp && env $v rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/prof_probe -o ep --output-format csv -- python3 $R/tools/prof_enc_pipe.py 6 > /dev/null 2>&1)
  python3 - <<PY
import csv, os
k = [r for r in csv.DictReader(open("$R/gpurun_out/prof_probe/ep_kernel_trace.csv")) if 'copyBuffer' in r['Kernel_Name']]
big = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in k if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 500000]
print("  blit-copy kernels: %d (of them %d longer than 0.5 ms, avg %.2f ms)" % (len(k), len(big), sum(big) / max(1, len(big)) / 1e6))
p = "$R/gpurun_out/prof_probe/ep_memory_copy_trace.csv"
if os.path.exists(p):
    c = [(r['Direction'], int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in csv.DictReader(open(p))]
    for d in sorted(set(x[0] for x in c)):
        v = [x[1] for x in c if x[0] == d]
        print("  copies %s: %d, longest %.2f ms" % (d, len(v), max(v) / 1e6))
PY
done
