# the collective path of bench.py --gpus N on the real RCCL communicator with the one rank a one-GPU box allows; valid JSON out
mkdir -p gpurun_out/r06
export HSA_ENABLE_IPC_MODE_LEGACY=0 RPE_BENCH_FORCE_DIST=1 RPE_BENCH_PREWARM_STEPS=300 MASTER_ADDR=127.0.0.1 MASTER_PORT=29547 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
python bench.py --gpus 1 --steps 20 --warmup 5 --repeats 20 --no-cpu-baseline --no-extras --no-hbm 2> gpurun_out/r06/bench_rccl_world1.err | tail -1 > gpurun_out/r06/bench_rccl_world1.json
python - <<'PY'
import json
j = json.load(open('gpurun_out/r06/bench_rccl_world1.json'))
x = json.load(open('bench_extras.json'))
print(json.dumps({"ms_per_step": j["ms_per_step"], "collective": x["config"]["collective"], "collective_step_us": x["config"]["collective_step_us"]}))
PY
