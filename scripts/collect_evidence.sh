#!/bin/bash
# Round evidence on one MI355X box: GPU tests, the bench line, rocprofv3 kernel stats of the same command, the two PMC
# passes (separate runs, counters only), and the roofline / mode probes.  Usage (through gpurun):
#   gpurun --timeout 2400 -- 'bash scripts/collect_evidence.sh r01c'
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $root/bench.py > $out/bench_n1.json 2> $out/bench_stderr.txt
tail -1 $out/bench_n1.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 $root/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extras > $out/bench_under_rocprof.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > /dev/null 2>&1
timeout 600 python3 $root/scripts/roofline_runs.py > $out/roofline_runs.jsonl 2>&1
timeout 300 python3 $root/scripts/gn_modes.py > $out/gn_modes.txt 2>&1
timeout 600 python3 $root/tests/perf/pipeline_times.py > $out/pipeline_times.jsonl 2>&1
timeout 300 python3 $root/scripts/frontend_times.py > $out/frontend_times.jsonl 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_frontend -- python3 $root/scripts/frontend_times.py > /dev/null 2>&1
RPE_QUIET=1 timeout 300 $root/examples/engine_profile > $out/engine_profile.txt 2>&1
# tests last (they include the multi-process-on-one-GPU cases, which are kept away from the measurements above)
RPE_TEST_MULTIPROC=1 timeout 1500 python3 -m pytest $root/tests -m gpu -q > $out/pytest_gpu.txt 2>&1
tail -3 $out/pytest_gpu.txt
ls -R $out | head -60
