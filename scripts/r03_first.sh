#!/bin/bash
# round 3, first GPU session: the new resident paths, the whole GPU suite, and the kernels this round changed under the clock
tag=${1:-r03a}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 900 python3 -m pytest tests/test_gpu_resident_paths.py -x -q > $out/pytest_resident.txt 2>&1
tail -15 $out/pytest_resident.txt
timeout 600 python3 scripts/roofline_runs.py 307200 1000000 > $out/roofline_runs.jsonl 2>&1
grep -E '"name": "(p2p|p2plane|bearing)"' $out/roofline_runs.jsonl | cut -c1-260
timeout 300 python3 scripts/joint_probe.py > $out/joint_probe.txt 2>&1; tail -12 $out/joint_probe.txt
timeout 1500 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1
tail -5 $out/pytest_gpu.txt
