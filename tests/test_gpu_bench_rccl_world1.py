"""-m gpu: bench.py's N > 1 code path on the REAL RCCL communicator, with the one rank a one-GPU box allows (RPE_BENCH_FORCE_DIST=1): the
collective plan, the communicator set up by the library, the verified all-reduce, rccl_ranks read back from the communicator, the
host-side exchange timed beside it, the headline carried by RCCL, the JSON fields the multi-GPU line promises."""
import json
import os
import subprocess
import sys

import pytest

from bench_util import run_bench

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_collective_path_on_rccl_with_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RPE_BENCH_FORCE_DIST="1", RPE_BENCH_PREWARM_STEPS="300", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547",
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r, line, j = run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--repeats", "10", "--no-cpu-baseline",
                        "--no-extras", "--no-hbm"], env, timeout=900)
    cfg = j["config"]
    assert j["n_gpus"] == 1 and j["value"] > 1e9
    assert cfg["collective"].startswith("rccl:") and "library-owned communicator" in cfg["collective"]
    assert cfg["rccl_ranks"] == 1 and cfg["rccl_verified"] is True and "ncclCommCount" in cfg["rccl_ranks_source"]
    assert cfg["collective_step_us"]["rccl_us"] > 0 and cfg["collective_step_us"]["host_us"] > 0 and cfg["collective_step_us"]["p2p_us"] is None
    assert cfg["collective_plan"]["headline"] == "rccl" and cfg["collective_plan"]["timed_beside"] == ["host"]
    assert isinstance(cfg["pci_bus_ids"], list) and len(cfg["pci_bus_ids"]) == 1 and ":" in cfg["pci_bus_ids"][0]
    assert "rpe_gn_steps_dist" in cfg["host_loop"]
