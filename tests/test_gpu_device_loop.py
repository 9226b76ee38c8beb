"""-m gpu: rpe_gn_refine_device in its one-launch form (the grid iterates by itself: granule hand-off, run records, solve + exp-map in every
workgroup; rpe_residuals.hpp resident_auto_stage) against the host-driven loop (rpe_gn_refine: same kernels' sums added in the same order,
solve on the host) and against its own one-launch-per-iteration form (RPE_DEVICE_LOOP_RESIDENT=0, a subprocess: the switch is read once)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from rgbd_pose_estimation_amd import _lib as L, api
import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene(n, dt, seed=700):
    sc = util.scene_full(seed + n % 97, n, dt, n2d=2.0, n3d=0.03, outliers=0.1, nan_frac=0.0)
    rng = np.random.default_rng(n)
    mask = (rng.uniform(size=n) < 0.8).astype(np.int16)
    w = rng.uniform(0.2, 2.0, n).astype(dt)
    p0 = api.pose12(*util.perturbed_pose(np.random.default_rng(2), sc.R, sc.t, 0.02, 0.05))
    return sc, mask, w, p0


@pytest.mark.parametrize("n", [1, 3, 4, 700, 4099, 61000, 307200, 1000003])
@pytest.mark.parametrize("kind", [L.RES_P2P, L.RES_P2PLANE])
@pytest.mark.parametrize("f64", [False, True])
def test_one_launch_loop_follows_the_host_driven_loop(gpu_ctx_factory, n, kind, f64):
    """every grid geometry: one workgroup, every workgroup its own run (few correspondences), runs of collecting workgroups (frame size),
    the slice re-read per iteration (1 M); iterations equal, poses equal to rounding of the device's sin / cos against libm's"""
    if n < 4 and kind == L.RES_P2PLANE:
        pytest.skip("fewer than 4 point-to-plane rows are rank-deficient")
    dt = np.float64 if f64 else np.float32
    sc, mask, w, p0 = _scene(n, dt)
    ctx = gpu_ctx_factory().load(L.F64 if f64 else L.F32, xw=sc.Q, xc=sc.P, nc=sc.N, nw=sc.M)
    ctx.upload_mask(L.MOD_33, mask); ctx.upload_weight(L.MOD_33, w)
    for flags in ((0, L.USE_MASK, L.USE_WEIGHT, L.USE_MASK | L.USE_WEIGHT) if n >= 700 else (0,)):
        try:
            ph, ith, steph, costh = ctx.gn_refine([kind], p0, None, flags, 25, 1e-9)
        except L.RpeError:
            with pytest.raises(L.RpeError):   # n = 1, 3: refused by both
                ctx.gn_refine_device([(kind, 1.0)], p0, flags, 25, 1e-9)
            continue
        pd, itd, stepd, costd = ctx.gn_refine_device([(kind, 1.0)], p0, flags, 25, 1e-9)
        assert itd == ith and 0 < itd <= 25
        # same records; the device solve rounds differently (one reciprocal per pivot, sincos).  Beyond one group per thread the two
        # loops run on different grids (the autonomous one leaves a compute unit to its solving workgroup: 255 workgroups against 256), so
        # their fp32 group sums are widened in different company: equal to the rounding of fp32 products, not of fp64 sums
        ptol = 1e-9 if n <= 500000 else 5e-8
        assert np.max(np.abs(pd - ph)) < ptol
        assert abs(costd - costh) <= (1e-9 if n <= 500000 else 1e-7) * max(abs(costh), 1e-30) and abs(stepd - steph) <= 1e-9 * steph + (1e-11 if n <= 500000 else 1e-9)   # the last step is at rounding level
        p2, it2, *_ = ctx.gn_refine_device([(kind, 1.0)], p0, flags, 2, 0.0)     # the iteration cap
        h2, *_ = ctx.gn_refine([kind], p0, None, flags, 2, 0.0)
        assert it2 == 2 and np.max(np.abs(p2 - h2)) < ptol


def test_back_to_back_loops_are_bitwise_reproducible(gpu_ctx_factory):
    """tags, granules and the double-buffered run records across thousands of launches of different lengths on one context, with other
    launches (collecting reductions, scoring) in between: the same inputs give the same bits"""
    n = 307200
    sc, mask, w, p0 = _scene(n, np.float32)
    ctx = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P, nc=sc.N, nw=sc.M, bv=sc.U)
    ctx.upload_mask(L.MOD_33, mask)
    ref = {K: ctx.gn_refine_device([(L.RES_P2P, 1.0)], p0, L.USE_MASK, K, 0.0) for K in (2, 3, 7, 20)}
    refpl = ctx.gn_refine_device([(L.RES_P2PLANE, 1.0)], p0, L.USE_MASK, 6, 0.0)
    q7 = api.pose7_from_Rt(sc.R, sc.t)
    for rep in range(1500):
        K = (2, 3, 7, 20)[rep % 4]
        out = ctx.gn_refine_device([(L.RES_P2P, 1.0)], p0, L.USE_MASK, K, 0.0)
        assert out[1] == K and np.array_equal(out[0], ref[K][0]) and out[2] == ref[K][2] and out[3] == ref[K][3]
        if rep % 50 == 0:
            ctx.normal_eq(L.RES_P2P, p0, L.USE_MASK)
            ctx.score(L.VOTE_33, q7[None], 0.2)
            ctx.gn_refine([L.RES_P2P], p0, None, L.USE_MASK, 5, 0.0)
            o2 = ctx.gn_refine_device([(L.RES_P2PLANE, 1.0)], p0, L.USE_MASK, 6, 0.0)
            assert np.array_equal(o2[0], refpl[0])


_CHILD = r'''
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from rgbd_pose_estimation_amd import _lib as L, api
import util
out = []
for n, kind in ((4099, 0), (307200, 0), (307200, 1)):
    sc = util.scene_full(700 + n % 97, n, np.float32, n2d=2.0, n3d=0.03, outliers=0.1, nan_frac=0.0)
    p0 = api.pose12(*util.perturbed_pose(np.random.default_rng(2), sc.R, sc.t, 0.02, 0.05))
    ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P, nc=sc.N, nw=sc.M)
    p, it, step, cost = ctx.gn_refine_device([(kind, 1.0)], p0, 0, 25, 1e-6)   # well above the rounding level of fp32 arrays
    out.append(dict(n=n, kind=kind, pose=[float(x) for x in p], it=it, cost=cost))
print(json.dumps(out))
'''


def test_one_launch_form_equals_one_launch_per_iteration_form():
    res = {}
    for sw in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", _CHILD, ROOT], env=dict(os.environ, RPE_DEVICE_LOOP_RESIDENT=sw), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        res[sw] = json.loads(r.stdout.strip().splitlines()[-1])
    for a, b in zip(res["1"], res["0"]):
        assert a["it"] == b["it"] and 0 < a["it"] < 25
        assert np.max(np.abs(np.array(a["pose"]) - np.array(b["pose"]))) < 1e-7    # different summation trees over the workgroups, stopped at |delta| < 1e-6
        assert abs(a["cost"] - b["cost"]) <= 1e-9 * abs(b["cost"])


def test_two_threads_two_contexts_one_gpu(gpu_ctx_factory):
    """resident loops of one process are serialised per GPU inside the library: two threads with a context each (ctypes releases the GIL
    during the calls) refine concurrently -- host-driven and one-launch device loops mixed -- and every result equals the single-threaded one"""
    import threading
    n = 307200
    sc, mask, w, p0 = _scene(n, np.float32)
    ctxs = [gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P, nc=sc.N, nw=sc.M) for _ in range(2)]
    ref_h = ctxs[0].gn_refine([L.RES_P2P], p0, None, 0, 12, 0.0)[0]
    ref_d = ctxs[0].gn_refine_device([(L.RES_P2P, 1.0)], p0, 0, 12, 0.0)[0]
    errors = []

    def work(ctx, first_device):
        try:
            for k in range(150):
                if (k + first_device) % 2:
                    out = ctx.gn_refine_device([(L.RES_P2P, 1.0)], p0, 0, 12, 0.0)[0]
                    if not np.array_equal(out, ref_d):
                        errors.append(("device", k))
                else:
                    out = ctx.gn_refine([L.RES_P2P], p0, None, 0, 12, 0.0)[0]
                    if not np.array_equal(out, ref_h):
                        errors.append(("host", k))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(ctxs[i], i)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ts)
    assert not errors, errors[:5]


@pytest.mark.parametrize("kind", [L.RES_P2PLANE, L.RES_P2P])
def test_the_solving_workgroup_and_its_workers_always_meet(gpu_ctx_factory, kind):
    """Beyond a frame the autonomous loop's grid is capped so that every shader engine keeps a compute unit free for the solving
    workgroup (rpe_normal_eq.hip auto_solver_cap).  With 225-255 heavy workers one of them started only when the others' bounded wait
    had run out: a lost loop (0.25 s, finished with one launch per iteration) within ten refinements, after which the context stops
    asking for a solving workgroup.  Forty refinements over 1.5 M correspondences with masks and weights: none may lose the loop."""
    import time
    n = 1_500_000
    sc, mask, w, p0 = _scene(n, np.float32)
    ctx = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P, nc=sc.N, nw=sc.M)
    ctx.upload_mask(L.MOD_33, mask); ctx.upload_weight(L.MOD_33, w)
    ctx.gn_refine([kind], p0, None, L.USE_MASK | L.USE_WEIGHT, 3, 0.0)           # (the arrays are verified finite: CLEAN flavour from here on)
    if not ctx.resident_state().get("solver"):
        pytest.skip("no solving workgroup on this device")
    first, slow = None, []
    for flags in (0, L.USE_MASK | L.USE_WEIGHT):
        first = None
        for i in range(20):
            t0 = time.perf_counter()
            p, it, *_ = ctx.gn_refine_device([(kind, 1.0)], p0, flags, 100, 0.0)
            dt = time.perf_counter() - t0
            assert it == 100
            if dt > 0.2:   # (a lost loop costs 0.25 s: the meeting timeout)
                slow.append((flags, i, round(dt, 3), L.lib().rpe_last_error().decode()[:160]))
            first = p if first is None else first
            assert np.array_equal(p, first)
    st = ctx.resident_state()
    assert not slow and st["solver"] and st["lost"] == 0, (slow, st)
