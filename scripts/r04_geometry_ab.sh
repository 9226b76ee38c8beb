#!/bin/bash
# round 4: the streaming normal-equation kernels after the instruction cuts -- time per launch by geometry (workgroup size x cap) and
# with / without the fp32 transform, 1 M and 10 M; then the SQ counters of the default geometry at 1 M (as scripts/sq_pmc.sh)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r04_geometry_ab.jsonl
K=K1_p2p,K2_p2plane,K3_bearing,K1p_moments
for f32 in 1 0; do
  for geo in "256 256" "256 512" "512 256" "512 512"; do
    set -- $geo
    RPE_F32_TRANSFORM=$f32 RPE_BLOCK=$1 RPE_MAX_BLOCKS=$2 python3 $root/scripts/kernel_roofline.py --kernels $K --sizes 1000000,10000000 --launches 30 --out $out --tag "f32=$f32,block=$1,cap=$2" > /dev/null 2>&1
  done
done
python3 - <<PY
import json
rows = [json.loads(l) for l in open("$out")]
for r in rows:
    if r["state"] == "steady":
        print(r["tag"], r["kernel"], r["n"], r["avg_us"], r["frac_of_peak"])
PY
bash $root/scripts/sq_pmc.sh r04sq > /dev/null 2>&1
cat $root/gpurun_out/r04sq/counters_*.csv | cut -c1-400 | head -60
