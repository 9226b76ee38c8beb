#!/bin/bash
# last check-in of the round on the final tree: GPU suite, smoke, the two bench lines, scoring kernel stats, session timings
cd "$GRAFT_REPO_ROOT" || exit 1
out=$GRAFT_REPO_ROOT/gpurun_out/r05f; mkdir -p $out
RPE_TEST_MULTIPROC=1 timeout 2400 python3 -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
RPE_BENCH_EXTRAS=$out/bench_driver_cmd_extras.json timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_stderr.txt; tail -c 400 $out/bench_driver_cmd.json; echo
RPE_BENCH_EXTRAS=$out/bench_default_extras.json timeout 900 python3 bench.py > $out/bench_default_2000steps.json 2>> $out/bench_stderr.txt; tail -c 300 $out/bench_default_2000steps.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_score -- python3 $GRAFT_REPO_ROOT/scripts/score_pmc_probe.py > $out/score_probe.json 2>/dev/null
f=$(ls $out/prof_score/*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/prof_score_kernel_stats.csv; rm -rf $out/prof_score
cd $GRAFT_REPO_ROOT
timeout 300 python3 scripts/score_filter_stats.py > $out/score_times.jsonl 2>/dev/null
timeout 300 python3 scripts/dev/session_time.py 307200 > $out/session_time.json 2>/dev/null
RPE_QUIET=1 timeout 120 ./examples/engine_profile 307200 totals > $out/engine_session_on.txt 2>&1
RPE_SCORE_SESSION=0 RPE_QUIET=1 timeout 120 ./examples/engine_profile 307200 totals > $out/engine_session_off.txt 2>&1
timeout 600 python3 scripts/device_loop_time.py > $out/device_loop_solver.jsonl 2>/dev/null
timeout 900 python3 scripts/soak.py 300 > $out/soak.json 2> $out/soak.err
grep ransac2 $out/engine_session_on.txt | tail -3; tail -c 600 $out/soak.json
