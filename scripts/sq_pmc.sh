#!/bin/bash
# SQ counters of the three normal-equation kernels at 1 M (one launch per call): separate --pmc passes, counters only.
tag=${1:-r03sq}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $out/avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*" $out/avail.txt | sort -u | tr '\n' ' ' | cut -c1-3000 > $out/avail_sq.txt
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_THREAD_CYCLES_VALU"; do
  name=$(echo $set | tr ' ' '+')
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $out/p_$name -- python3 $root/scripts/kernel_pmc_probe.py > $out/run_$name.txt 2>&1
  f=$(ls $out/p_$name/*/*_counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && (head -1 $f; grep "normal_eq_kernel" $f) > $out/counters_$name.csv
  rm -rf $out/p_$name
done
ls -la $out
