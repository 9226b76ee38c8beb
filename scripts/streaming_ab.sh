#!/bin/bash
# round 4: the streaming normal-equation kernels -- CLEAN flavour (default, clean-first protocol) vs guarded (RPE_GUARD_ALWAYS=1),
# event-timed, 307 200 / 1 M / 10 M, steady + cold
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-r04_streaming_ab}.jsonl
K=K1_p2p,K1_p2p_mask,K2_p2plane,K3_bearing
for v in "RPE_GUARD_ALWAYS=0" "RPE_GUARD_ALWAYS=1"; do
  env $v python3 $root/scripts/kernel_roofline.py --kernels $K --sizes 307200,1000000,10000000 --launches 40 --out $out --tag "$v" > /dev/null 2>&1
done
python3 - <<PY
import json
rows = [json.loads(l) for l in open("$out")]
for r in rows:
    print(r["tag"], r["kernel"], r["n"], r["state"], r["avg_us"], r["frac_of_peak"])
PY
