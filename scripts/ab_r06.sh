# A/B of library builds on ONE box (boxes of the pool differ by +-5 %): usage ab_r06.sh tagA=libA.so tagB=libB.so ; alternates three times
mkdir -p gpurun_out/r06
for rep in 1 2 3; do
 for spec in "$@"; do
  tag=${spec%%=*}; lib=$PWD/${spec#*=}
  RPE_LIBRARY=$lib python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-hbm 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$tag', j['ms_per_step'], j['config']['ms_per_step_p10'], j['config']['ms_per_step_p90'], j['roofline']['avg_launch_us'])"
 done
done | tee gpurun_out/r06/ab_resident.txt
