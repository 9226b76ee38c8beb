#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r06k}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC"; do
  name=$(echo $set | tr ' ' '+')
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/p_$name -- python3 $root/scripts/score_pmc_probe.py 307200 512 4 > $out/run_$name.txt 2>&1
  f=$(ls $out/p_$name/*/*_counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && (head -1 $f; grep "score_kernel" $f) > $out/counters_$name.csv
  rm -rf $out/p_$name
done
python3 - <<EOF
import csv,glob,collections,re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/counters_*.csv"):
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(.*","",r["Kernel_Name"]); agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k, {c: round(sum(v)/len(v)) for c,v in sorted(agg[k].items())})
EOF
