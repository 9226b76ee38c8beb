"""-m gpu: bench.py refuses to describe a path that did not run.  If a resident grid is lost during the run (another process on the
GPU; injected here through the per-context test hook) the library finishes those refinements with one launch per iteration and
returns RPE_OK -- the timed repetitions then mix two paths, so the line says `valid: false` with the reason, carries the resident
state it read before and after the timed region, and the process exits non-zero (round-3 advisor finding)."""
import json
import os
import subprocess
import sys

import pytest

from bench_util import run_bench

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lost_grid_makes_the_line_invalid():
    env = dict(os.environ, RPE_BENCH_INJECT_LOST_GRID="3", RPE_BENCH_PREWARM_S="0.2", RPE_BENCH_NO_UNTUNED="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r, line, j = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--repeats", "4", "--no-extras",
                        "--no-cpu-baseline", "--no-hbm"], env, timeout=900, expect_rc=3)
    assert "INVALID RUN" in r.stderr
    assert j["valid"] is False and "lost" in j["invalid_reason"]
    st = j["config"]["resident_state"]
    assert st["lost_in_run"] >= 1 and st["resident_loop_ran"] is False and st["before"]["lost"] == 0
    assert "one launch per step" in j["config"]["host_loop"] and j["roofline"]["steps_per_launch"] == 1


def test_an_undisturbed_run_is_valid():
    env = dict(os.environ, RPE_BENCH_PREWARM_S="0.2", RPE_BENCH_NO_UNTUNED="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RPE_BENCH_INJECT_LOST_GRID"):
        env.pop(k, None)
    r, line, j = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--repeats", "4", "--no-extras",
                        "--no-cpu-baseline", "--no-hbm"], env, timeout=900)
    assert "valid" not in j and j["config"]["resident_state"]["resident_loop_ran"] is True and j["roofline"]["steps_per_launch"] == 20
    # the committed PMC profile of this command stands in the line while the kernel sources are the ones it was taken on; otherwise the
    # line carries null and says which hashes disagree (bench.py kernel_source_hash)
    roof = j["roofline"]
    if roof["traffic"] is not None:
        assert 0.03 < roof["traffic_over_algorithmic"] < 0.2 and roof["kernel_src_sha256"] in roof["traffic_source"]
    else:
        assert roof["traffic_source"].startswith("none:") and roof["kernel_src_sha256"] in roof["traffic_source"]
