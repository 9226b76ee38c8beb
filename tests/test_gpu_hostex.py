"""-m gpu: sharded Gauss-Newton and scoring whose all-reduce is the host-side exchange (rpe_hostex_init), with 2 / 4 ranks sharing
the one GPU of the test box (tests/hostex_worker.py).  No kernel ever waits for another process here (the exchange happens between
the host threads), so these cases are part of the default -m gpu suite."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu]
# the cases in which kernels of different processes must run on the one GPU AT THE SAME TIME (two resident grids that wait for each
# other's hosts), and the 4-rank frame-sized case: default on, RPE_TEST_MULTIPROC=0 switches them off
multiproc = pytest.mark.skipif(os.environ.get("RPE_TEST_MULTIPROC") == "0", reason="RPE_TEST_MULTIPROC=0: the larger multi-process-on-one-GPU cases are switched off")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(world, mode, n, steps, env_extra=None, timeout=240):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RPE_QUIET="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "hostex_worker.py"), mode, str(n), str(steps)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("host-exchange workers timed out")
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    line = [l for l in outs[0].splitlines() if l.startswith("RESULT ")]
    assert line, outs[0][-2000:]
    return json.loads(line[0][7:])


def check_poses(res):
    ranks = res["ranks"]
    assert all(r["hostex"] and "error" not in r for r in ranks), ranks
    poses = [np.array(r["pose"]) for r in ranks]
    for p in poses[1:]:
        assert np.array_equal(p, poses[0])            # rank-ordered sums: bitwise the same record, hence pose, on every rank
    assert np.abs(poses[0] - np.array(res["reference"])).max() < 1e-9   # shards add up to the whole (different summation order only)
    return ranks


@pytest.mark.parametrize("world,n", [(2, 20000), pytest.param(4, 307200, marks=multiproc)])
def test_sharded_steps_match_the_single_gpu_run(world, n):
    check_poses(run_world(world, "steps", n, 6))


@pytest.mark.parametrize("resident", [False, pytest.param(True, marks=multiproc)])
def test_sharded_refine_resident_and_launch_per_step(resident):
    """rpe_gn_refine on a sharded context: ranks that share a GPU launch once per iteration; with one GPU per rank (simulated here:
    RPE_HOSTEX_ALLOW_SHARED=1 on a problem whose grids are all resident at once) every rank keeps its resident kernel."""
    res = run_world(2, "refine", 20000, 30, {"RPE_HOSTEX_ALLOW_SHARED": "1"} if resident else None)
    ranks = check_poses(res)
    assert ranks[0]["iters"] == ranks[1]["iters"] == ranks[0]["ref_iters"] < 30


def test_sharded_votes_add_up():
    res = run_world(2, "score", 50001, 40)
    ranks = res["ranks"]
    assert all(r["hostex"] and "error" not in r for r in ranks), ranks
    for r in ranks:
        assert r["votes"] == res["reference"]
        assert r["votes_short"] == res["reference"][:12]


def test_hostex_single_rank_and_misuse(gpu_ctx_factory):
    """world = 1: the exchange is the identity, every sharded entry point still works; a second init is refused; sharded contexts refuse
    the device generators (they sample the local shard)."""
    import ctypes as C
    import util
    from rgbd_pose_estimation_amd import _lib as L, api
    sc = util.scene33(4, 20000, np.float32, noise=0.02, outliers=0.0)
    ctx = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P)
    ref = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P)
    name = f"/rpe_hx_test_solo_{os.getpid()}"
    ctx.hostex_init(1, 0, name, True)
    assert not os.path.exists("/dev/shm" + name)          # the name is dropped as soon as every rank has the segment mapped
    with pytest.raises(L.RpeError) as e:
        ctx.hostex_init(1, 0, name + "b", True)
    assert e.value.code == L.RPE_ERR_STATE
    p0 = api.pose12(np.eye(3), np.zeros(3))
    a = ctx.gn_refine([L.RES_P2P], p0, max_iter=20, tol=1e-10)
    b = ref.gn_refine([L.RES_P2P], p0, max_iter=20, tol=1e-10)
    assert np.array_equal(a[0], b[0]) and a[1] == b[1]
    pa, pb = p0.copy(), p0.copy()
    ctx.gn_steps_dist(L.RES_P2P, pa, 3)
    for _ in range(3):
        ref.gn_step(L.RES_P2P, pb)
    assert np.abs(pa - pb).max() < 1e-12
    votes = np.zeros(8, np.int32); q7 = np.zeros((8, 7)); valid = np.zeros(8, np.uint8)
    rc = L.lib().rpe_ransac33_batch(ctx._h, C.c_uint64(1), C.c_uint64(109), 8, L.SCORE_EXACT, 0.2, votes.ctypes.data_as(C.c_void_p),
                                    q7.ctypes.data_as(C.c_void_p), valid.ctypes.data_as(C.c_void_p))
    assert rc == L.RPE_ERR_STATE
    ctx.hostex_destroy()
    ctx.hostex_destroy()                                   # idempotent
