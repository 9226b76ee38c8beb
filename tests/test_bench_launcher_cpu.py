"""CPU: `python bench.py --gpus N` from a bare shell (no torch.distributed.run around it).  The parent must start N rank processes
with the rendezvous environment, touch no GPU itself (here: it must not even import torch or the HIP library), forward rank 0's JSON
line and fail if any rank fails.  The ranks run in dry-run mode: they leave before anything initialises HIP."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, extra_env, gpus=3):
    # a torch / package stub that explodes on import: neither the parent nor a dry-run rank may import them
    stub = tmp_path / "stubs"
    (stub / "torch").mkdir(parents=True, exist_ok=True)
    (stub / "torch" / "__init__.py").write_text("raise ImportError('torch imported by the launcher')\n")
    env = dict(os.environ, PYTHONPATH=str(stub), RPE_BENCH_DRY_RUN="1", RPE_LIBRARY="/nonexistent/librgbdpose_hip.so", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "20", "--warmup", "5"], env=env,
                          capture_output=True, text=True, timeout=120)


def test_bare_shell_launch_spawns_ranks_and_forwards_rank0(tmp_path):
    r = _run(tmp_path, {})
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["dry_run"] and j["rank"] == 0 and j["world"] == 3 and j["gpus"] == 3 and j["steps"] == 20
    assert j["master"].startswith("127.0.0.1:") and int(j["master"].split(":")[1]) > 0
    assert len([l for l in r.stdout.strip().splitlines() if l.startswith("{")]) == 1   # only rank 0's line reaches stdout
    # what the run will do about the per-iteration all-reduce, decided before a GPU is touched: the north star's collective (RCCL on the
    # library's communicator) carries the headline, the host-side exchange is timed beside it, every rank must sit on its own GPU
    plan = j["collective_plan"]
    assert plan["headline"] == "rccl" and plan["rccl_init"] and plan["rccl_ranks_expected"] == 3
    assert plan["timed_beside"] == ["host"] and plan["distinct_gpu_check"] and plan["scaling"] == "strong" and plan["weak_scaling_extra"]
    for field in ("config.collective", "config.rccl_ranks", "config.rccl_verified", "config.collective_step_us", "config.pci_bus_ids", "weak_scaling"):
        assert field in plan["json_fields"]


def test_collective_plan_modes(tmp_path):
    for want, headline, beside, ranks in (("rccl", "rccl", [], 8), ("host", "host", [], 0), ("auto_p2p", "rccl", ["p2p"], 8), ("p2p", "p2p", [], 8)):
        r = _run(tmp_path, {"RPE_BENCH_COLLECTIVE": want}, gpus=8)
        assert r.returncode == 0, r.stderr
        plan = json.loads(r.stdout.strip().splitlines()[-1])["collective_plan"]
        assert (plan["headline"], plan["timed_beside"], plan["rccl_ranks_expected"]) == (headline, beside, ranks), (want, plan)
    r = _run(tmp_path, {"RPE_BENCH_COLLECTIVE": "auto", "RPE_BENCH_SHARE_GPU": "1"}, gpus=2)   # the one-GPU test rig switches the check off
    assert not json.loads(r.stdout.strip().splitlines()[-1])["collective_plan"]["distinct_gpu_check"]
    r = _run(tmp_path, {"RPE_BENCH_COLLECTIVE": "nonsense"}, gpus=2)
    assert r.returncode != 0


def test_bare_shell_launch_fails_when_a_rank_fails(tmp_path):
    r = _run(tmp_path, {"RPE_BENCH_DRY_EXIT": "7", "RPE_BENCH_DRY_EXIT_RANK": "2"})
    assert r.returncode != 0 and "rank(s) failed" in r.stderr


def test_single_gpu_invocation_does_not_spawn(tmp_path):
    r = _run(tmp_path, {}, gpus=1)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["world"] == 1 and j["rank"] == 0
    assert j["collective_plan"]["headline"] == "none" and j["collective_plan"]["rccl_ranks_expected"] == 0
