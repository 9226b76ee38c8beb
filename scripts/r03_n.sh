#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_resident_paths.py tests/test_gpu_device_loop.py tests/test_gpu_joint.py tests/test_gpu_pipelines.py tests/test_gpu_frontend.py -q -x 2>&1 | tail -3
timeout 300 python3 scripts/device_loop_ab.py 2>/dev/null | cut -c1-200
timeout 600 python3 scripts/resident_timeline.py 2>/dev/null | cut -c1-900
for i in 1 2; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-hbm --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver', j['ms_per_step'], j['value'], j['roofline']['avg_launch_us_per_step'])"; python3 bench.py --no-cpu-baseline --no-hbm --no-extras 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', j['ms_per_step'], j['value'], j['roofline']['avg_launch_us_per_step'])"; done
timeout 300 python3 scripts/roofline_runs.py 307200 1000000 2>/dev/null | grep -E '"name": "(p2p|p2plane|bearing)"' | cut -c1-150
