#!/bin/bash
# full GPU suite + the world=1 RCCL bench line + the default bench line (round-5 check-in)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r05_gpu_suite.txt 2>&1
tail -5 gpurun_out/r05_gpu_suite.txt
RPE_BENCH_FORCE_DIST=1 RPE_BENCH_EXTRAS=gpurun_out/r05_bench_rccl_world1_extras.json timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_rccl_world1.json 2> gpurun_out/r05_bench_rccl_world1.err
tail -c 1800 gpurun_out/r05_bench_rccl_world1.json
RPE_BENCH_EXTRAS=gpurun_out/r05_bench_default_extras.json timeout 900 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
tail -c 1800 gpurun_out/r05_bench_default.json
