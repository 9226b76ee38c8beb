"""The host-side all-reduce between the rank processes of one node (csrc/rpe_hostex.cpp), exercised WITHOUT a GPU: the protocol
(parity slots, release / acquire step numbers, rank-ordered sums, bounded waits) is plain host code."""
import json
import os
import secrets
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(world, steps, mode="plain", ranks=None, timeout=120):
    name = f"/rpe_hx_test_{os.getpid()}_{secrets.token_hex(4)}"
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "hostex_cpu_worker.py"), name, str(world), str(r), str(steps), mode],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in (ranks if ranks is not None else range(world))]
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=timeout)
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    res = [json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][0][7:]) for o in outs]
    assert not os.path.exists("/dev/shm" + name) or mode == "timeout"
    if os.path.exists("/dev/shm" + name):
        os.unlink("/dev/shm" + name)
    return res


@pytest.mark.parametrize("world", [1, 2, 5, 8])
def test_sums_are_rank_ordered_and_identical(world):
    res = run_world(world, 400)
    assert all("error" not in r for r in res), res
    assert all(r["bad"] == 0 and not r["collide"] for r in res), res
    assert res[0]["left_in_dev_shm"] is False


def test_shared_gpu_labels_are_noticed():
    res = run_world(2, 3, "collide")
    assert all(r["collide"] for r in res), res


def test_missing_peer_fails_the_call_instead_of_hanging():
    res = run_world(2, 3, "timeout", ranks=[0])   # rank 1 never shows up
    assert "did not deliver" in res[0].get("error", ""), res
    # the caller's counters are still its own (the sums live in a local buffer until every rank has delivered), and the handle refuses
    # further use at once instead of running every later step into the same wait
    assert res[0]["first_rc"] != 0 and res[0]["untouched"]
    assert res[0]["second_rc"] != 0 and "earlier step timed out" in res[0]["second_error"] and res[0]["second_s"] < 0.5


def test_bad_arguments():
    sys.path.insert(0, ROOT)
    from rgbd_pose_estimation_amd import _lib as L, api
    with pytest.raises(L.RpeError):
        api.HostExchange("no_leading_slash", 2, 0, True)
    with pytest.raises(L.RpeError):
        api.HostExchange("/rpe_hx_test_args", 9, 0, True)
    hx = api.HostExchange(f"/rpe_hx_test_{os.getpid()}_solo", 1, 0, True)
    assert np.array_equal(hx.allreduce_f64([1.5, 2.5]), [1.5, 2.5])
    with pytest.raises(L.RpeError):
        hx.allreduce_f64(np.zeros(65))
    hx.close()   # the creator's close removes the name
    assert not os.path.exists(f"/dev/shm/rpe_hx_test_{os.getpid()}_solo")
