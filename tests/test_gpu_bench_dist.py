"""-m gpu: bench.py's N > 1 code path from a bare shell -- the self-launcher, fixed collective-step counts, peer-to-peer record
verification, strong-scaling workload, JSON contract -- with two ranks sharing the one GPU of the test box (gloo carries the plumbing
there because RCCL refuses two ranks on one device; RPE_BENCH_SHARE_GPU=1 maps both ranks to cuda:0).  The driver's real multi-GPU
run uses one GPU per rank and the RCCL communicator."""
import json
import os
import subprocess
import sys

import pytest

from bench_util import run_bench

# Several processes share the ONE GPU of the test box here and wait for each other inside kernels (every wait bounded): part of the
# default suite, RPE_TEST_MULTIPROC=0 switches it off.
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("RPE_TEST_MULTIPROC") == "0",
                                                  reason="RPE_TEST_MULTIPROC=0: multi-process-on-one-GPU tests are switched off")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_gpu_from_a_bare_shell():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RPE_BENCH_COLLECTIVE="p2p", RPE_BENCH_PREWARM_STEPS="300", RPE_BENCH_BACKEND="gloo",
               RPE_BENCH_STRICT_COLLECTIVE="1", RPE_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r, line, j = run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--repeats", "10", "--n-total", "614400",
                        "--no-cpu-baseline"], env, timeout=900)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["steps"] == 20 and j["value"] > 1e9
    assert j["config"]["global_corr"] == 614400 and j["config"]["corr_rank0"] == 307200
    assert 0.8 * 614400 < j["config"]["valid_corr_per_step"] < 614400
    assert "configs[4]" in j["config"]["workload"] and "peer-to-peer" in j["config"]["collective"]
    assert j["timing"]["repeats"] == 10 and j["roofline"]["launches_timed"] >= 20
    assert j["pose_error_vs_truth"]["rot_rad"] < 1e-2
    assert j["device_resident_loop"]["iterations"] == 500
