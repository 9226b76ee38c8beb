"""-m gpu: bench.py's N > 1 path with the host-side exchange, from a bare shell, two ranks sharing the one GPU of the test box (gloo
carries the plumbing; RPE_BENCH_SHARE_GPU=1 maps both ranks to cuda:0, so rpe_gn_refine launches once per iteration instead of keeping
two resident grids that would wait for each other's hosts).  No kernel waits for another process on this path."""
import json
import os
import subprocess
import sys

import pytest

from bench_util import run_bench

pytestmark = [pytest.mark.gpu]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_host_exchange_from_a_bare_shell():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RPE_BENCH_COLLECTIVE="host", RPE_BENCH_PREWARM_STEPS="300", RPE_BENCH_BACKEND="gloo",
               RPE_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r, line, j = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--repeats", "10", "--n-total", "614400",
                        "--no-cpu-baseline"], env, timeout=900)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["steps"] == 20 and j["value"] > 1e9
    assert j["config"]["global_corr"] == 614400 and j["config"]["corr_rank0"] == 307200
    assert "configs[4]" in j["config"]["workload"] and "host-side exchange" in j["config"]["collective"]
    assert j["config"]["collective_step_us"]["host_us"] > 0 and j["config"]["rccl_ranks"] == 0
    assert j["pose_error_vs_truth"]["rot_rad"] < 1e-2
    w = j["weak_scaling"]                                    # a frame-sized shard per rank, same exchange
    assert w["corr_per_rank"] == 307200 and w["global_corr"] == 614400 and w["value"] > 1e9 and 0.8 * 614400 < w["valid_corr_per_step"] <= 614400


def test_bench_falls_back_together_when_one_rank_loses_the_exchange():
    """auto mode: rank 1 drops out of the exchange's trial refinement; rank 0 runs into the exchange's bounded wait, both agree on the
    failure and the run finishes on the other collective (torch.distributed here: two ranks on one GPU cannot form an RCCL communicator)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RPE_BENCH_COLLECTIVE="auto", RPE_BENCH_PREWARM_STEPS="100", RPE_BENCH_BACKEND="gloo",
               RPE_BENCH_SHARE_GPU="1", RPE_BENCH_INJECT_HOSTEX_FAIL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r, line, j = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--repeats", "5", "--n-total", "614400",
                        "--no-cpu-baseline", "--no-extras"], env, timeout=900)
    assert "host-side exchange dropped" in r.stderr
    assert j["n_gpus"] == 2 and j["value"] > 1e8 and "host-side exchange" not in j["config"]["collective"]
    assert j["pose_error_vs_truth"]["rot_rad"] < 1e-2
