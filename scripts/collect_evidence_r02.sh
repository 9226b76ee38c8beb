#!/bin/bash
# Round-2 evidence on one MI355X box (through gpurun):   gpurun --timeout 2400 -- 'bash scripts/collect_evidence_r02.sh r02x'
# bench line (default and the driver's command), rocprofv3 kernel stats of the bench command, the two PMC passes (separate runs,
# counters only), config-4 cold / steady, roofline sweep, engine profile, pipeline times, the resident-loop and collecting-stage A/Bs,
# the phase timelines (diagnostic builds: build_stamps(1), build_stamps(2) must have been built before the call), then the GPU tests
# (multi-process cases included, last).
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $root/bench.py > $out/bench_n1.json 2> $out/bench_stderr.txt
tail -c 600 $out/bench_n1.json; echo
timeout 300 python3 $root/bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2>> $out/bench_stderr.txt
# every launch of the resident kernel the same length as the timed ones (no pre-warm, warmup = steps): the CSV's average IS the average launch
RPE_BENCH_PREWARM_S=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 $root/bench.py --steps 2000 --warmup 2000 --no-cpu-baseline --no-extras > $out/bench_under_rocprof.txt 2>&1
# counters: separate passes, counters only
RPE_BENCH_PREWARM_S=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 2000 --warmup 2000 --repeats 5 --no-cpu-baseline --no-extras > /dev/null 2>&1
RPE_BENCH_PREWARM_S=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 2000 --warmup 2000 --repeats 5 --no-cpu-baseline --no-extras > /dev/null 2>&1
timeout 300 python3 $root/scripts/config4_p2plane.py > $out/config4_cold_steady.jsonl 2>&1
timeout 300 python3 $root/scripts/config4_p2plane.py 1250000 p2p >> $out/config4_cold_steady.jsonl 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_config4 -- python3 $root/scripts/config4_p2plane.py > /dev/null 2>&1
timeout 600 python3 $root/scripts/roofline_runs.py > $out/roofline_runs.jsonl 2>&1
timeout 600 python3 $root/scripts/resident_ab.py > $out/resident_ab.jsonl 2>/dev/null
timeout 600 python3 $root/scripts/collect_ab.py > $out/collect_ab.jsonl 2>/dev/null
timeout 300 python3 $root/scripts/resident_timeline.py > $out/resident_timeline.jsonl 2>/dev/null
timeout 900 python3 $root/scripts/tail_timeline.py > $out/tail_timeline.jsonl 2>/dev/null
timeout 300 python3 $root/scripts/refine_intercept.py > $out/refine_intercept.jsonl 2>/dev/null
timeout 600 python3 $root/tests/perf/pipeline_times.py > $out/pipeline_times.jsonl 2>&1
RPE_QUIET=1 timeout 300 $root/examples/engine_profile > $out/engine_profile.txt 2>&1
timeout 300 python3 $root/scripts/frontend_times.py > $out/frontend_times.jsonl 2>&1
RPE_TEST_MULTIPROC=1 timeout 1500 python3 -m pytest $root/tests -m gpu -q > $out/pytest_gpu.txt 2>&1
tail -3 $out/pytest_gpu.txt
ls -R $out | head -80
