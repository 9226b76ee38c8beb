# A/B of ENVIRONMENT settings of one build on ONE box: usage ab_env_r06.sh tagA=VAR=value tagB= ...  (an empty setting = the default);
# alternates four times with the driver's bench command, prints ms_per_step | p10 | p90 | average launch of the resident kernel
mkdir -p gpurun_out/r06
for rep in 1 2 3 4; do
 for spec in "$@"; do
  tag=${spec%%=*}; setting=${spec#*=}
  env $setting python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-hbm 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$tag', j['ms_per_step'], j['config']['ms_per_step_p10'], j['config']['ms_per_step_p90'], j['roofline']['avg_launch_us'])"
 done
done | tee gpurun_out/r06/ab_env.txt
