"""-m gpu: the peer-to-peer all-reduce inside the normal-equation kernel (rpe_p2p_export / rpe_p2p_init / rpe_gn_step_dist),
exercised with 2, 3 and 8 ranks that share the one GPU of the test box (tests/p2p_worker.py)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

# Several processes share the ONE GPU of the test box here and wait for each other inside kernels.  That depends on the driver scheduling
# the processes' queues concurrently; every wait is bounded (a missing peer ends in RPE_ERR_HIP after 10 s, never in a hang).  These
# cases are part of the default -m gpu suite (world 2 / 3 / 8 on one GPU, the sharded device loop, sharded scoring, the missing-peer
# time-out: ~25 s together); RPE_TEST_MULTIPROC=0 leaves only the smallest case, and the 20 000-exchange soak runs on request only
# (RPE_TEST_MULTIPROC=1).
pytestmark = [pytest.mark.gpu]
multiproc = pytest.mark.skipif(os.environ.get("RPE_TEST_MULTIPROC") == "0", reason="RPE_TEST_MULTIPROC=0: the larger multi-process-on-one-GPU cases are switched off")
soak = pytest.mark.skipif(os.environ.get("RPE_TEST_MULTIPROC") != "1", reason="the exchange soak runs with RPE_TEST_MULTIPROC=1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(world, mode, n, steps, timeout=240):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RPE_QUIET="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py"), mode, str(n), str(steps)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("p2p workers timed out")
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    line = [l for l in outs[0].splitlines() if l.startswith("RESULT ")]
    assert line, outs[0][-2000:]
    return json.loads(line[0][7:])


@pytest.mark.parametrize("world,n", [(2, 20000), pytest.param(3, 100003, marks=multiproc), pytest.param(8, 307200, marks=multiproc)])
def test_p2p_sharded_steps_match_the_single_gpu_run(world, n):
    res = run_world(world, "steps", n, 6)
    ranks = res["ranks"]
    assert all(r["p2p"] for r in ranks), ranks
    assert all("error" not in r for r in ranks), ranks
    poses = [np.array(r["pose"]) for r in ranks]
    for p in poses[1:]:
        assert np.array_equal(p, poses[0])            # rank-ordered sums: bitwise the same record on every rank
    ref = np.array(res["reference"])
    assert np.abs(poses[0] - ref).max() < 1e-9         # shards add up to the whole (different summation order only)


def test_p2p_reinit_after_an_odd_session():
    """rpe_p2p_init a second time after ONE step: the restarted step 0 must not take the first session's records for delivered."""
    res = run_world(2, "reinit", 20000, 5)
    ranks = res["ranks"]
    assert all(r["p2p"] and "error" not in r for r in ranks), ranks
    poses = [np.array(r["pose"]) for r in ranks]
    assert np.array_equal(poses[0], poses[1])
    assert np.abs(poses[0] - np.array(res["reference"])).max() < 1e-9


@multiproc
@pytest.mark.parametrize("world,n", [(2, 20000), (8, 307200)])
def test_p2p_sharded_device_resident_loop(world, n):
    """One launch per iteration on every rank: exchange + sum + solve + exp-map in the kernel's last workgroup."""
    res = run_world(world, "device", n, 30)
    ranks = res["ranks"]
    assert all(r["p2p"] and "error" not in r for r in ranks), ranks
    poses = [np.array(r["pose"]) for r in ranks]
    for p, r in zip(poses, ranks):
        assert np.array_equal(p, poses[0]) and r["iters"] == ranks[0]["iters"]   # identical records -> identical decisions
    assert ranks[0]["iters"] == ranks[0]["ref_iters"] < 30
    assert np.abs(poses[0] - np.array(res["reference"])).max() < 1e-9


@multiproc
@pytest.mark.parametrize("world,n,H", [(2, 20000, 70), (8, 100003, 3000)])
def test_p2p_sharded_scoring_counts_are_exact(world, n, H):
    """rpe_score after rpe_p2p_init: every rank gets the votes of the WHOLE correspondence set (integer sums, exact)."""
    res = run_world(world, "score", n, H)
    ranks = res["ranks"]
    assert all(r["p2p"] and "error" not in r for r in ranks), ranks
    for r in ranks:
        assert r["votes"] == res["reference"]
        assert r["votes2"] == res["reference"][::-1]


@soak
@pytest.mark.parametrize("world", [2, 4])
def test_p2p_exchange_soak(world):
    """20 000 exchanges from a fixed pose: every rank sees bitwise the same record every time, and the same as its peers."""
    res = run_world(world, "soak", 40000, 20000)
    ranks = res["ranks"]
    assert all(r["p2p"] and "error" not in r for r in ranks), [{k: v for k, v in r.items() if k != "record"} for r in ranks]
    assert all(r["bad"] == 0 for r in ranks)
    for r in ranks[1:]:
        assert r["record"] == ranks[0]["record"]


@multiproc
def test_p2p_missing_peer_times_out_instead_of_hanging():
    res = run_world(2, "straggler", 20000, 3)
    r0, r1 = res["ranks"]
    assert r0["p2p"] and r1["p2p"]
    assert r1.get("slept") and "error" not in r1
    assert r0.get("code") == -2 and "timed out" in r0["error"]     # RPE_ERR_HIP after 10 s, no hang
