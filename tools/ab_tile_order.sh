#!/bin/bash
# usage (on the GPU box): tools/ab_tile_order.sh  -- K1 / K2 and their HBM traffic with the luma tiles dealt to the XCDs by rows (shipped) and in plain row-major order
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in "" 1; do
  if [ -z "$mode" ]; then unset HIPJPEG_ROW_MAJOR_TILES; echo "== rows dealt to XCDs (shipped)"; else export HIPJPEG_ROW_MAJOR_TILES=1; echo "== plain row-major order"; fi
  python3 $R/bench.py --kernels-only --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   value %.0f images/s, %.4f ms per step; K1 %.4f K2 %.4f ms; roofline.frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernels'][0]['avg_ms'], d['roofline']['kernels'][1]['avg_ms'], d['roofline']['frac']))"
  i=0
  for grp in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1)); rm -rf $R/gpurun_out/ab_tiles_pmc/g$i
    (cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/ab_tiles_pmc/g$i -- python3 $R/bench.py --kernels-only --steps 3 --warmup 1 > /dev/null 2>&1)
  done
  python3 $R/tools/traffic_from_pmc.py $R/gpurun_out/ab_tiles_pmc $R/gpurun_out/ab_tiles_traffic.json | grep -A3 "luma_color" | head -5
done
