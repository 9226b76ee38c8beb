#!/bin/bash
tag=${1:-r03c}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout 300 python3 -m pytest tests/test_gpu_resident_paths.py tests/test_gpu_device_loop.py tests/test_gpu_joint.py tests/test_gpu_examples.py -q -x > $out/pytest_sel.txt 2>&1; tail -5 $out/pytest_sel.txt
timeout 300 python3 scripts/device_loop_ab.py > $out/device_loop_ab.jsonl 2>&1; cat $out/device_loop_ab.jsonl | cut -c1-330
timeout 600 python3 scripts/geometry_sweep_r03.py > $out/geometry_sweep.jsonl 2>&1; cat $out/geometry_sweep.jsonl
python3 -c "
import sys, json
sys.path.insert(0, 'tests')
import config3_case
print(json.dumps(config3_case.run(with_cpu=True)))" > $out/config3.json 2>&1; tail -c 900 $out/config3.json
