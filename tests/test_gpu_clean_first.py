"""-m gpu: the CLEAN-first protocol of the normal-equation kernels (rpe_receive.hip clean-first protocol): the flavour without NaN guards is
launched first, a non-finite record sends the launch through the guarded flavour, and from then on the arrays are known to need
the guards.  RPE_GUARD_ALWAYS=1 (read at rpe_create) pins a context to the guarded flavour: on finite arrays the two flavours add the
same bits, on NaN-marked arrays (the reference's "invalid measurement" columns, AOPoseAdapter.hpp:147-152) the protocol must hand back
exactly what the guarded flavour computes -- first call (CLEAN tried, repeated), later calls (guarded at once), one launch per
iteration, the resident loop and the device-resident loop."""
import os

import numpy as np
import pytest

from rgbd_pose_estimation_amd import _lib as L, api
import util

pytestmark = pytest.mark.gpu
KINDS = [L.RES_P2P, L.RES_P2PLANE, L.RES_BEARING, L.RES_REPROJ]


def _ctx(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return api.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("nan_frac", [0.0, 0.03])
@pytest.mark.parametrize("n", [7, 4099, 307200, 1000003])
@pytest.mark.parametrize("kind", KINDS)
def test_records_equal_the_guarded_flavours(kind, n, nan_frac):
    sc = util.scene_full(60 + n % 97, n, np.float32, n2d=2.0, n3d=0.03, nan_frac=nan_frac)
    if nan_frac:
        sc.U[::37] = np.nan          # bearings without a measurement too
        sc.Q[5 % n] = np.inf         # and an infinity in a world point that an inlier mask would have switched off
    rng = np.random.default_rng(n)
    pose = api.pose12(*util.perturbed_pose(rng, sc.R, sc.t))
    mod = L.MOD_23 if kind in (L.RES_BEARING, L.RES_REPROJ) else L.MOD_33
    mask = (rng.uniform(size=n) < 0.7).astype(np.int16)
    mask[5 % n] = 0
    w = rng.uniform(0.1, 2.0, n).astype(np.float32)
    out = {}
    for name, env in (("clean_first", {}), ("guarded", {"RPE_GUARD_ALWAYS": "1"})):
        ctx = _ctx(env).load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nc=sc.N)
        try:
            ctx.upload_mask(mod, mask); ctx.upload_weight(mod, w)
            recs = [ctx.normal_eq(kind, pose, flags=f)[0] for f in (L.USE_MASK, L.USE_MASK, L.USE_MASK | L.USE_WEIGHT)]   # first call, later call
            out[name] = recs
        finally:
            ctx.close()
    for a, b in zip(out["clean_first"], out["guarded"]):
        assert np.all(np.isfinite(b[:29]))
        assert np.array_equal(a, b)


@pytest.mark.parametrize("kind", KINDS)
def test_refinements_on_nan_marked_arrays(kind, oracle):
    """Host-driven resident loop, one launch per iteration and the device-resident loop on NaN-marked arrays: the same pose as a context
    pinned to the guarded flavour, and (point-to-point) the oracle's closed form over the valid correspondences."""
    n = 307200
    sc = util.scene_full(77, n, np.float32, n2d=2.0, n3d=0.03, outliers=0.0, nan_frac=0.04)
    sc.U[::41] = np.nan
    p0 = api.pose12(*util.perturbed_pose(np.random.default_rng(3), sc.R, sc.t, 0.01, 0.03))
    res = {}
    for name, env in (("clean_first", {}), ("guarded", {"RPE_GUARD_ALWAYS": "1"}), ("per_launch", {"RPE_RESIDENT": "0"})):
        ctx = _ctx(env).load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nc=sc.N)
        try:
            a = ctx.gn_refine([kind], p0, max_iter=12, tol=1e-10)
            b = ctx.gn_refine([kind], p0, max_iter=12, tol=1e-10)            # the arrays are known to need the guards by now
            d = ctx.gn_refine_device([(kind, 1.0)], p0, max_iter=12, tol=1e-10)
            res[name] = (a, b, d)
        finally:
            ctx.close()
    g = res["guarded"][0]
    assert g[1] < 12 and np.all(np.isfinite(g[0]))
    for name, runs in res.items():
        for r in runs:
            assert r[1] == g[1], (name, r[1], g[1])
            assert np.max(np.abs(r[0] - g[0])) < 1e-9, name
    assert np.array_equal(res["clean_first"][0][0], g[0]) and np.array_equal(res["clean_first"][1][0], g[0])
    if kind == L.RES_P2P:
        ok = ~np.isnan(sc.P).all(axis=1)
        Ro, to, _ = oracle.shinji_f32in_f64(sc.Q[ok], sc.P[ok])
        assert util.rot_err(g[0][:9].reshape(3, 3), Ro) < 1e-7 and np.linalg.norm(g[0][9:] - to) / np.linalg.norm(to) < 1e-7


def test_reprojection_launches_do_not_vouch_for_nan_marked_bearings():
    """(advisor, round 4) The reprojection term switches a correspondence whose bearing is NaN off by its own validity test, so a CLEAN
    reprojection record over NaN-marked bearing columns could come back finite and promote XW / BV to "verified finite" -- after which
    the device-consumed BEARING launches (no record inspected on the host) would run unguarded over the NaN columns.  The CLEAN
    reprojection flavour now multiplies its inputs into the cost slot; the sequence REPROJ refinement -> device-resident BEARING loop
    must give what a context pinned to the guarded flavour gives."""
    n = 50000
    sc = util.scene_full(91, n, np.float32, n2d=1.0, n3d=0.02, outliers=0.0)
    sc.U[::29] = np.nan
    p0 = api.pose12(*util.perturbed_pose(np.random.default_rng(5), sc.R, sc.t, 0.01, 0.03))
    res = {}
    for name, env in (("clean_first", {}), ("guarded", {"RPE_GUARD_ALWAYS": "1"})):
        ctx = _ctx(env).load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nc=sc.N)
        try:
            rec = ctx.normal_eq(L.RES_REPROJ, p0)[0]
            a = ctx.gn_refine([L.RES_REPROJ], p0, max_iter=12, tol=1e-10)
            d = ctx.gn_refine_device([(L.RES_BEARING, 1.0)], p0, max_iter=12, tol=1e-10)
            res[name] = (rec, a, d)
        finally:
            ctx.close()
    g, c = res["guarded"], res["clean_first"]
    assert np.all(np.isfinite(g[0][:29])) and np.array_equal(c[0], g[0])
    assert np.all(np.isfinite(g[1][0])) and np.array_equal(c[1][0], g[1][0]) and c[1][1] == g[1][1]
    assert np.all(np.isfinite(c[2][0])) and c[2][1] == g[2][1] and np.max(np.abs(c[2][0] - g[2][0])) < 1e-9
