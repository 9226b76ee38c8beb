#!/bin/bash
# Round-3 evidence on one MI355X box (through gpurun):   gpurun --timeout 3000 -- 'bash scripts/collect_evidence_r03.sh r03b'
# The bench line (default and the DRIVER'S command), rocprofv3 --kernel-trace --stats of the driver's command, the two --pmc passes
# (separate runs, counters only) on that command and on the 2000-step default, then the kernels this round changed, then the GPU tests.
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
drv="--gpus 1 --steps 20 --warmup 5"
timeout 300 python3 -m pytest $root/tests/test_gpu_resident_paths.py $root/tests/test_gpu_score_filter.py $root/tests/test_gpu_examples.py -q > $out/pytest_new.txt 2>&1
tail -8 $out/pytest_new.txt
timeout 600 python3 $root/bench.py $drv > $out/bench_driver_cmd.json 2> $out/bench_stderr.txt
tail -c 400 $out/bench_driver_cmd.json; echo
timeout 600 python3 $root/bench.py > $out/bench_default_2000steps.json 2>> $out/bench_stderr.txt
# the driver's own command under rocprofv3 (every launch of the resident kernel serves 20 steps, the --warmup launch 5)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_driver -- python3 $root/bench.py $drv > $out/bench_driver_under_rocprof.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_driver_noextras -- python3 $root/bench.py $drv --no-extras --no-cpu-baseline --no-hbm > $out/bench_driver_noextras_under_rocprof.txt 2>&1
# counters: separate passes, counters only; a short pre-warm (every launch is serialised under --pmc)
RPE_BENCH_PREWARM_S=0.1 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_20 -- python3 $root/bench.py $drv --repeats 20 --no-extras --no-cpu-baseline --no-hbm > /dev/null 2>&1
RPE_BENCH_PREWARM_S=0.1 timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_20 -- python3 $root/bench.py $drv --repeats 20 --no-extras --no-cpu-baseline --no-hbm > /dev/null 2>&1
RPE_BENCH_PREWARM_S=0.1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_2000 -- python3 $root/bench.py --steps 2000 --warmup 2000 --no-extras --no-cpu-baseline --no-hbm > $out/bench_2000_under_rocprof.txt 2>&1
RPE_BENCH_PREWARM_S=0.1 timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_2000 -- python3 $root/bench.py --steps 2000 --warmup 2000 --repeats 5 --no-extras --no-cpu-baseline --no-hbm > /dev/null 2>&1
RPE_BENCH_PREWARM_S=0.1 timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_2000 -- python3 $root/bench.py --steps 2000 --warmup 2000 --repeats 5 --no-extras --no-cpu-baseline --no-hbm > /dev/null 2>&1
# keep what is needed of the (large) traces: the stats files whole, the counter files per dispatch of the resident kernel
for d in prof_driver prof_driver_noextras prof_2000; do f=$(ls $out/$d/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/${d}_kernel_stats.csv; done
for d in pmc_fetch_20 pmc_write_20 pmc_fetch_2000 pmc_write_2000; do f=$(ls $out/$d/*/*_counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && (head -1 $f; grep normal_eq_resident $f) > $out/${d}_counters.csv; done
rm -rf $out/prof_driver $out/prof_driver_noextras $out/prof_2000 $out/pmc_fetch_20 $out/pmc_write_20 $out/pmc_fetch_2000 $out/pmc_write_2000
timeout 600 python3 $root/scripts/roofline_runs.py 307200 1000000 > $out/roofline_runs.jsonl 2>&1
grep -E '"name": "(p2p|p2plane|bearing)"|33_23 exact' $out/roofline_runs.jsonl | cut -c1-230
timeout 300 python3 $root/scripts/config4_p2plane.py > $out/config4_cold_steady.jsonl 2>&1
timeout 300 python3 $root/scripts/device_loop_ab.py > $out/device_loop_ab.jsonl 2>&1
RPE_TEST_MULTIPROC=1 timeout 1800 python3 -m pytest $root/tests -m gpu -q > $out/pytest_gpu.txt 2>&1
tail -5 $out/pytest_gpu.txt
ls -la $out
