#!/bin/bash
tag=${1:-r03d}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
./scripts/ubench/valu_rates > $out/valu_rates.jsonl 2>&1; cat $out/valu_rates.jsonl
timeout 900 python3 scripts/resident_timeline.py > $out/resident_timeline.jsonl 2> $out/resident_timeline.err; cat $out/resident_timeline.jsonl; tail -3 $out/resident_timeline.err
