#!/bin/bash
# sweep of the fused ICP kernel's launch geometry (development aid)
root=${GRAFT_REPO_ROOT:-$(pwd)}
for blk in 256 512; do for grid in 64 128 150 192 256 300 512; do
  echo -n "blk=$blk grid=$grid  "
  RPE_ICP_BLOCK=$blk RPE_ICP_GRID=$grid RPE_MAX_BLOCKS=1024 timeout 120 python3 $root/scripts/frontend_times.py 2>&1 | grep "device-resident, fused" | sed 's/.*us_per_round": \([0-9.]*\).*/\1/'
done; done
