#!/bin/bash
# usage (on the GPU box): tools/round_profile.sh <tag>  -- bench line, rocprofv3 kernel stats of the same command, HBM traffic (PMC passes)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
cd $R
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench --output-format csv -- python3 $R/bench.py > $R/gpurun_out/${TAG}_prof_bench.json 2> $R/gpurun_out/${TAG}_prof.err
cp $R/gpurun_out/${TAG}_prof/bench_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
head -8 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
cd $R
bash tools/pmc.sh ${TAG}_pmc tools/prof_decode.py 3 > gpurun_out/${TAG}_pmc.txt 2>&1
python3 tools/traffic_from_pmc.py gpurun_out/${TAG}_pmc gpurun_out/${TAG}_hbm_traffic.json | tail -5
