"""-m gpu: the resident (one launch per refinement) forms added in round 3 -- bearing kind, joint kernel -- against the one-launch-per-
iteration loops and the oracle; and the failure handling of resident loops: a grid that loses a workgroup's sums is finished with one
launch per iteration (the call succeeds), an early stop followed at once by the next call does not stall, a smaller device (fewer
workgroups resident at once) gives the same answer."""
import os
import time

import numpy as np
import pytest

from rgbd_pose_estimation_amd import _lib as L, api
import util

pytestmark = pytest.mark.gpu


def _ctx(env=None):
    """a context created under a temporary environment (RPE_RESIDENT / RPE_RESIDENT_CAP are read at creation / first use)"""
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        return api.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _pose_close(a, b, rot=1e-8, rel=1e-8):
    """same objective, same iteration count, sums added in a different order (fp32 products): the poses agree to a few 1e-9"""
    assert util.rot_err(a[:9].reshape(3, 3), b[:9].reshape(3, 3)) < rot
    assert util.trans_rel_err(a[9:], b[9:]) < rel


@pytest.mark.parametrize("n", [2000, 307200, 1000000])
@pytest.mark.parametrize("f64", [False, True])
def test_bearing_refine_resident_equals_launch_per_iteration(oracle, n, f64):
    dt = np.float64 if f64 else np.float32
    sc = util.scene_full(900 + n, n, dt, n2d=2.0, n3d=0.03, nan_frac=0.02)
    p0 = api.pose12(*util.perturbed_pose(np.random.default_rng(n), sc.R, sc.t, 0.01, 0.03))
    res, per = _ctx(), _ctx({"RPE_RESIDENT": "0"})
    try:
        assert res.resident_state()["host_driven"] and not per.resident_state()["host_driven"]
        out = []
        for c in (res, per):
            c.load(L.F64 if f64 else L.F32, xw=sc.Q, bv=sc.U)
            out.append(c.gn_refine([L.RES_BEARING], p0, max_iter=12, tol=0.0))
        (pa, ia, sa, ca), (pb, ib, sb, cb) = out
        assert ia == ib == 12
        _pose_close(pa, pb)
        assert abs(ca - cb) <= 1e-9 * abs(cb)
        if n <= 307200:   # and the oracle's fp64 Gauss-Newton on the same residual (P3P.hpp:482-485)
            po, _, _, _ = oracle.gn_refine([dict(kind=L.RES_BEARING, a=sc.Q, b=sc.U)], n, p0, max_iter=12, tol=0.0)
            assert util.rot_err(pa[:9].reshape(3, 3), po[:9].reshape(3, 3)) < util.ROT_TOL_RAD
            assert util.trans_rel_err(pa[9:], po[9:]) < util.TRANS_REL_TOL
        assert res.resident_state()["lost"] == 0
    finally:
        res.close(); per.close()


COMBOS = [(L.RES_P2P, L.RES_BEARING), (L.RES_P2PLANE, L.RES_BEARING), (L.RES_P2P, L.RES_BEARING, L.RES_NORMAL), (L.RES_P2P, L.RES_NORMAL)]


@pytest.mark.parametrize("n", [3000, 307200, 1500000])
@pytest.mark.parametrize("combo", COMBOS, ids=["+".join(map(str, c)) for c in COMBOS])
def test_joint_refine_resident_equals_launch_per_iteration(n, combo):
    sc = util.scene_full(700 + n, n, np.float32, n2d=2.0, n3d=0.03, nan_frac=0.03)
    rng = np.random.default_rng(n)
    p0 = api.pose12(*util.perturbed_pose(rng, sc.R, sc.t, 0.01, 0.03))
    masks = {m: (rng.uniform(size=n) < 0.8).astype(np.int16) for m in (L.MOD_23, L.MOD_33, L.MOD_NN)}
    weights = {m: rng.uniform(0.2, 2.0, n).astype(np.float32) for m in (L.MOD_23, L.MOD_33, L.MOD_NN)}
    terms = [(k, [1.0, 4.0, 0.5][i], L.ROBUST_HUBER if i == 1 else 0, 0.01) for i, k in enumerate(combo)]
    res, per = _ctx(), _ctx({"RPE_RESIDENT": "0"})
    try:
        out = []
        for c in (res, per):
            c.load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
            for m in masks:
                c.upload_mask(m, masks[m]); c.upload_weight(m, weights[m])
            out.append(c.gn_refine_joint(terms, p0, flags=L.USE_MASK | L.USE_WEIGHT, max_iter=10, tol=0.0))
        (pa, ia, sa, ca), (pb, ib, sb, cb) = out
        assert ia == ib == 10
        _pose_close(pa, pb)
        assert abs(ca - cb) <= 1e-9 * abs(cb)
        # an early stop takes the same number of iterations either way
        ea = res.gn_refine_joint(terms, p0, flags=L.USE_MASK | L.USE_WEIGHT, max_iter=40, tol=1e-7)
        eb = per.gn_refine_joint(terms, p0, flags=L.USE_MASK | L.USE_WEIGHT, max_iter=40, tol=1e-7)
        assert ea[1] == eb[1] and ea[1] < 40
        _pose_close(ea[0], eb[0])
    finally:
        res.close(); per.close()


@pytest.mark.parametrize("what", ["p2p", "bearing", "joint"])
def test_lost_grid_is_finished_with_one_launch_per_iteration(what):
    """rpe_debug_inject_resident_fault(k): the last workgroup withholds its sums of iteration k.  Its collecting workgroup gives up after its
    bounded wait and tells the host, which releases the grid and finishes the refinement with one launch per iteration: the call succeeds
    with the result an undisturbed run gives, and the context counts one lost grid."""
    n = 307200
    sc = util.scene_full(55, n, np.float32, n2d=2.0, n3d=0.03)
    p0 = api.pose12(*util.perturbed_pose(np.random.default_rng(5), sc.R, sc.t, 0.01, 0.03))
    c = _ctx().load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U)
    try:
        def run():
            if what == "p2p":
                return c.gn_refine([L.RES_P2P], p0, max_iter=8, tol=0.0)
            if what == "bearing":
                return c.gn_refine([L.RES_BEARING], p0, max_iter=8, tol=0.0)
            return c.gn_refine_joint([(L.RES_P2P, 1.0, 0, 1.0), (L.RES_BEARING, 4.0, 0, 1.0)], p0, max_iter=8, tol=0.0)
        good = run()
        # the workgroups' wait for the next pose is set to 9 s here: the call must come back after the collecting workgroup's 2 s, i.e.
        # the host's stop tag RELEASES the workgroups that already wait for the pose after the unfinished iteration
        c.inject_resident_fault(4, 9.0)
        try:
            t0 = time.perf_counter()
            hit = run()
            c.synchronize()   # the grid has left the GPU too
            dt = time.perf_counter() - t0
        finally:
            c.inject_resident_fault(0, 0.0)
        assert hit[1] == good[1] == 8
        _pose_close(hit[0], good[0])
        st = c.resident_state()
        assert st["lost"] == 1 and st["enabled"]
        assert 1.5 < dt < 5.0   # the collecting workgroup's bounded wait (2 s), not the pose wait (9 s) of every other workgroup on top
        again = run()           # and the context goes on with resident loops
        _pose_close(again[0], good[0])
        assert c.resident_state()["lost"] == 1
    finally:
        c.close()


def test_early_stops_back_to_back_do_not_stall():
    """Every call stops after a few iterations (step < tol) and the next call follows at once: a workgroup that has not yet seen the stop
    tag when the next call's first tag overwrites it must take the larger tag for what it is (the advisor's 2 s stall)."""
    n = 307200
    sc = util.scene_full(56, n, np.float32, n3d=0.03)
    p0 = api.pose12(*util.perturbed_pose(np.random.default_rng(6), sc.R, sc.t, 0.01, 0.03))
    c = _ctx().load(L.F32, xw=sc.Q, xc=sc.P)
    try:
        c.gn_refine([L.RES_P2P], p0, max_iter=30, tol=1e-6)
        t0 = time.perf_counter()
        worst = 0.0
        for _ in range(3000):
            t1 = time.perf_counter()
            p, it, _, _ = c.gn_refine([L.RES_P2P], p0, max_iter=30, tol=1e-6)
            worst = max(worst, time.perf_counter() - t1)
            assert 1 <= it < 30
        assert worst < 0.5, f"a call took {worst:.2f} s"
        assert c.resident_state()["lost"] == 0
    finally:
        c.close()


def test_smaller_device_same_answer():
    """RPE_RESIDENT_CAP = 40 (a partition with 40 compute units): the resident grid shrinks to what is resident at once, every workgroup
    sweeps several groups per iteration, and the refinement gives the answer of the full device."""
    import subprocess, sys, json
    code = r'''
import json, numpy as np, sys
sys.path.insert(0, "tests")
from rgbd_pose_estimation_amd import _lib as L, api
import util
n = 307200
sc = util.scene_full(57, n, np.float32, n2d=2.0, n3d=0.03)
p0 = api.pose12(*util.perturbed_pose(np.random.default_rng(7), sc.R, sc.t, 0.01, 0.03))
c = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U)
a = c.gn_refine([L.RES_P2P], p0, max_iter=8, tol=0.0)
b = c.gn_refine_joint([(L.RES_P2P, 1.0, 0, 1.0), (L.RES_BEARING, 4.0, 0, 1.0)], p0, max_iter=8, tol=0.0)
d = c.gn_refine_device([(L.RES_P2P, 1.0)], p0, 0, 8, 0.0)
print(json.dumps({"state": c.resident_state(), "p2p": a[0].tolist(), "joint": b[0].tolist(), "device": d[0].tolist()}))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for cap in ("", "40"):
        env = dict(os.environ)
        env.pop("RPE_RESIDENT_CAP", None)
        if cap:
            env["RPE_RESIDENT_CAP"] = cap
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    full, small = outs
    assert small["state"]["cap"] == 40 and full["state"]["cap"] > 40
    assert small["state"]["lost"] == 0 and small["state"]["enabled"]
    for k in ("p2p", "joint", "device"):
        _pose_close(np.array(full[k]), np.array(small[k]))


def test_host_wait_variants_agree_bitwise():
    """The host adds the run records in run order whichever way it waits for them: all tags in branch-free sweeps (default) or pair by
    pair (RPE_HOST_SWEEP = 0, read once per process)."""
    import subprocess, sys, json
    code = r'''
import json, numpy as np, sys
sys.path.insert(0, "tests")
from rgbd_pose_estimation_amd import _lib as L, api
import util
out = {}
for n in (3000, 307200):
    sc = util.scene_full(61, n, np.float32, n2d=2.0, n3d=0.03, nan_frac=0.02)
    p0 = api.pose12(*util.perturbed_pose(np.random.default_rng(9), sc.R, sc.t, 0.01, 0.03))
    c = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nc=sc.N, nw=sc.M)
    out[f"p2p_{n}"] = c.gn_refine([L.RES_P2P], p0, max_iter=12, tol=0.0)[0].tolist()
    out[f"plane_{n}"] = c.gn_refine([L.RES_P2PLANE], p0, max_iter=12, tol=0.0)[0].tolist()
    out[f"joint_{n}"] = c.gn_refine_joint([(L.RES_P2P, 1.0, 0, 1.0), (L.RES_BEARING, 4.0, 0, 1.0)], p0, max_iter=12, tol=0.0)[0].tolist()
    out[f"ne_{n}"] = c.normal_eq(L.RES_P2PLANE, p0)[0].tolist()
    c.close()
print(json.dumps(out))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for sweep in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, RPE_HOST_SWEEP=sweep), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0].keys() == outs[1].keys()
    for k in outs[0]:
        assert np.array_equal(np.array(outs[0][k]), np.array(outs[1][k])), k
