#!/bin/bash
# Round-6 evidence on one MI355X box (through gpurun):   gpurun --timeout 4500 -- 'bash scripts/collect_evidence_r06.sh r06e'
# GPU tests, the bench line (default and the DRIVER'S command), rocprofv3 --kernel-trace --stats and the two --pmc passes (separate
# runs, counters only) on the driver's command / the 2000-step default / the bandwidth-bound launches / the reference-API kernels, the
# step budget from the stamps build, the SQ counters of the streaming kernels, a fuzz campaign.  Summarised by profile_summary_r06.py.
tag=${1:-r06e}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
drv="--gpus 1 --steps 20 --warmup 5"
timeout 1800 python3 -m pytest $root/tests -m gpu -q > $out/pytest_gpu.txt 2>&1   # (the suite as the driver runs it: multi-process cases on, soak off)
tail -4 $out/pytest_gpu.txt
timeout 900 python3 $root/bench.py $drv > $out/bench_driver_cmd.json 2> $out/bench_stderr.txt
tail -c 300 $out/bench_driver_cmd.json; echo
timeout 900 python3 $root/bench.py > $out/bench_default_2000steps.json 2>> $out/bench_stderr.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_driver -- python3 $root/bench.py $drv > $out/bench_driver_under_rocprof.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_driver_noextras -- python3 $root/bench.py $drv --no-extras --no-cpu-baseline --no-hbm > $out/bench_driver_noextras_under_rocprof.txt 2>&1
RPE_BENCH_PREWARM_S=0.1 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_20 -- python3 $root/bench.py $drv --repeats 20 --no-extras --no-cpu-baseline --no-hbm > /dev/null 2>&1
RPE_BENCH_PREWARM_S=0.1 timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_20 -- python3 $root/bench.py $drv --repeats 20 --no-extras --no-cpu-baseline --no-hbm > /dev/null 2>&1
RPE_BENCH_PREWARM_S=0.1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_2000 -- python3 $root/bench.py --steps 2000 --warmup 2000 --no-extras --no-cpu-baseline --no-hbm > $out/bench_2000_under_rocprof.txt 2>&1
RPE_BENCH_PREWARM_S=0.1 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_2000 -- python3 $root/bench.py --steps 2000 --warmup 2000 --repeats 5 --no-extras --no-cpu-baseline --no-hbm > /dev/null 2>&1
RPE_BENCH_PREWARM_S=0.1 timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_2000 -- python3 $root/bench.py --steps 2000 --warmup 2000 --repeats 5 --no-extras --no-cpu-baseline --no-hbm > /dev/null 2>&1
for d in prof_driver prof_driver_noextras prof_2000; do f=$(ls $out/$d/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/${d}_kernel_stats.csv; done
for d in pmc_fetch_20 pmc_write_20 pmc_fetch_2000 pmc_write_2000; do f=$(ls $out/$d/*/*_counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && (head -1 $f; grep normal_eq_resident $f) > $out/${d}_counters.csv; done
rm -rf $out/prof_driver $out/prof_driver_noextras $out/prof_2000 $out/pmc_fetch_20 $out/pmc_write_20 $out/pmc_fetch_2000 $out/pmc_write_2000
# the bandwidth-bound launches of roofline_hbm (normal_eq_kernel: 20 M / 10 M point-to-point, 1 M point-to-plane)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_hbm -- python3 $root/scripts/hbm_stream_probe.py > $out/hbm_stream_probe.json 2>/dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_hbm -- python3 $root/scripts/hbm_stream_probe.py > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_hbm -- python3 $root/scripts/hbm_stream_probe.py > /dev/null 2>&1
f=$(ls $out/prof_hbm/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/prof_hbm_kernel_stats.csv
f=$(ls $out/prof_hbm/*/*_kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && (head -1 $f; grep normal_eq_kernel $f) > $out/prof_hbm_kernel_trace.csv
for d in pmc_fetch_hbm pmc_write_hbm; do f=$(ls $out/$d/*/*_counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && (head -1 $f; grep normal_eq_kernel $f) > $out/${d}_counters.csv; done
rm -rf $out/prof_hbm $out/pmc_fetch_hbm $out/pmc_write_hbm
# the reference-API kernels (K1' moments, K5 nl_round, K4b mask), one size per process
for n in 307200 1000000 10000000; do
  RPE_PROBE_N=$n timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_api_$n -- python3 $root/scripts/reference_api_probe.py > $out/reference_api_probe_$n.json 2>/dev/null
  RPE_PROBE_N=$n timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_api_$n -- python3 $root/scripts/reference_api_probe.py > /dev/null 2>&1
  RPE_PROBE_N=$n timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_api_$n -- python3 $root/scripts/reference_api_probe.py > /dev/null 2>&1
  f=$(ls $out/prof_api_$n/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/prof_api_${n}_kernel_stats.csv
  for d in pmc_fetch_api_$n pmc_write_api_$n; do f=$(ls $out/$d/*/*_counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && (head -1 $f; grep -E "moments_kernel|nl_round|mask_kernel" $f) > $out/${d}_counters.csv; done
  rm -rf $out/prof_api_$n $out/pmc_fetch_api_$n $out/pmc_write_api_$n
done
# where a step's time goes on this tree (stamps build of the resident kernels; built here, never shipped)
timeout 900 python3 $root/scripts/resident_timeline.py > $out/resident_timeline.jsonl 2> $out/resident_timeline.err
rm -f $root/rgbd_pose_estimation_amd/lib/*stamps*
# SQ counters of the streaming kernels at 1 M
bash $root/scripts/sq_pmc.sh $tag/sq > /dev/null 2>&1
# streaming kernels by flavour, and the examples' tuned / untuned loop
$root/examples/gn_refine_main 20 50 > $out/gn_refine_main_20.txt 2>&1
$root/examples/gn_refine_main 2000 10 > $out/gn_refine_main_2000.txt 2>&1
# ---- this round's additions
# the world = 1 RCCL step (exp-map on the host, and chained on the device) beside the resident / host-exchange step: ONE JSON line (the
# RCCL banner goes to stderr), its extras, and the rocprofv3 kernel list of that command
(cd $root && HSA_ENABLE_IPC_MODE_LEGACY=0 RPE_BENCH_FORCE_DIST=1 RPE_BENCH_PREWARM_STEPS=300 MASTER_ADDR=127.0.0.1 MASTER_PORT=29547 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 RPE_BENCH_EXTRAS=$out/bench_rccl_world1_extras.json \
  timeout 600 python3 bench.py $drv --repeats 20 --no-cpu-baseline --no-extras --no-hbm 2>> $out/bench_stderr.txt | tail -1 > $out/bench_rccl_world1.json)
(cd $root && HSA_ENABLE_IPC_MODE_LEGACY=0 RPE_BENCH_FORCE_DIST=1 RPE_BENCH_PREWARM_STEPS=300 MASTER_ADDR=127.0.0.1 MASTER_PORT=29548 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 RPE_BENCH_EXTRAS=$out/tmp_extras.json \
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_rccl -- python3 bench.py $drv --repeats 10 --no-extras --no-cpu-baseline --no-hbm > /dev/null 2>&1)
f=$(ls $out/prof_rccl/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/prof_rccl_kernel_stats.csv; rm -rf $out/prof_rccl $out/tmp_extras.json
timeout 120 $root/scripts/ubench/stream_signal > $out/stream_signal.jsonl 2>/dev/null
# one-launch kernels, event-timed by size
timeout 600 python3 $root/scripts/kernel_roofline.py --sizes 307200,1000000,10000000 --kernels K1_p2p,K2_p2plane,K3_bearing,K1p_moments,K5_nl_round,J_p2p_bearing,J_p2plane_bearing,J_p2p_bearing_normal,K4b_mask_33,K4b_mask_33_23,K4b_mask_nn_33_23 --launches 40 --out $out/kernel_roofline.jsonl --tag r06 > /dev/null 2>&1
# autonomous loops (solving workgroup on / off)
timeout 600 python3 $root/scripts/device_loop_time.py > $out/device_loop_solver.jsonl 2>/dev/null
RPE_AUTO_SOLVER=0 timeout 600 python3 $root/scripts/device_loop_time.py > $out/device_loop_nosolver.jsonl 2>/dev/null
# scoring: kernel stats of 512 x 307 200 passes, the band filter's fall-through counter, the session
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_score -- python3 $root/scripts/score_pmc_probe.py > $out/score_probe.json 2>/dev/null
f=$(ls $out/prof_score/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/prof_score_kernel_stats.csv; rm -rf $out/prof_score
(cd $root && python3 -c "from rgbd_pose_estimation_amd import build; print(build.build_score_stats())") > $out/build_score_stats.txt 2>&1
RPE_LIBRARY=$root/rgbd_pose_estimation_amd/lib/librgbdpose_hip_scorestats.so timeout 600 python3 $root/scripts/score_filter_stats.py > $out/score_filter_stats.jsonl 2>/dev/null
rm -f $root/rgbd_pose_estimation_amd/lib/*scorestats* $root/rgbd_pose_estimation_amd/lib/rpe_score_stats.o
timeout 300 python3 $root/scripts/session_time.py 307200 > $out/session_time.json 2>/dev/null
RPE_QUIET=1 timeout 120 $root/examples/engine_profile 307200 totals > $out/engine_session_on.txt 2>&1
RPE_SCORE_SESSION=0 RPE_QUIET=1 timeout 120 $root/examples/engine_profile 307200 totals > $out/engine_session_off.txt 2>&1
RPE_QUIET=1 timeout 120 $root/examples/engine_profile 307200 > $out/engine_session_phases.txt 2>&1
# soak of the cross-workgroup protocols, randomised campaign against the oracle on this tree (RPE_EVIDENCE_SHORT=1: a re-run of the
# measurements after a kernel change leaves them out)
if [ -z "$RPE_EVIDENCE_SHORT" ]; then
timeout 900 python3 $root/scripts/soak.py ${RPE_SOAK_S:-300} > $out/soak.json 2> $out/soak.err
RPE_FUZZ_SEEDS=${RPE_FUZZ_SEEDS:-1200} timeout 2400 python3 -m pytest $root/tests/test_gpu_fuzz.py -q > $out/fuzz_campaign.txt 2>&1
tail -3 $out/fuzz_campaign.txt
fi
ls $out | head -80
