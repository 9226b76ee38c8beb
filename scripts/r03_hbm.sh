#!/bin/bash
tag=${1:-r03hbm}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 python3 $root/scripts/hbm_stream_probe.py > $out/hbm_stream_plain.json 2>/dev/null; cat $out/hbm_stream_plain.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $root/scripts/hbm_stream_probe.py > $out/hbm_stream_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/scripts/hbm_stream_probe.py > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/scripts/hbm_stream_probe.py > /dev/null 2>&1
f=$(ls $out/prof/*/*_kernel_stats.csv | head -1); cp $f $out/hbm_stream_kernel_stats.csv
for d in pmc_fetch pmc_write; do f=$(ls $out/$d/*/*_counter_collection.csv | head -1); (head -1 $f; grep "normal_eq_kernel" $f) > $out/${d}_counters.csv; done
rm -rf $out/prof $out/pmc_fetch $out/pmc_write
head -4 $out/hbm_stream_kernel_stats.csv | cut -c1-250
