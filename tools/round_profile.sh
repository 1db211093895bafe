#!/bin/bash
# usage (on the GPU box): tools/round_profile.sh <tag>
#   1. the contract bench line                                  -> gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the CLEAN steady-state run (bench.py --kernels-only: setup once, warm-up, timed
#      steps, nothing else) so that the average durations of the CSV are the timed steps'  -> gpurun_out/<tag>_kernel_stats.csv
#   3. HBM traffic: separate --pmc passes of the same clean run  -> gpurun_out/<tag>_hbm_traffic.json (with the source hash)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
head -c 700 gpurun_out/${TAG}_bench.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench --output-format csv -- python3 $R/bench.py --kernels-only --steps 100 --warmup 5 > $R/gpurun_out/${TAG}_prof_bench.json 2> $R/gpurun_out/${TAG}_prof.err
cp $R/gpurun_out/${TAG}_prof/bench_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
head -14 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
cd $R
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmc/g$i -- python3 $R/bench.py --kernels-only --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_pmc_g$i.log 2>&1) || { echo "pmc group $i failed"; tail -5 gpurun_out/${TAG}_pmc_g$i.log; }
done
python3 tools/traffic_from_pmc.py gpurun_out/${TAG}_pmc gpurun_out/${TAG}_hbm_traffic.json | tail -12
