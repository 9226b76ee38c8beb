"""-m gpu: the two-float fp32 transform of the streaming Gauss-Newton kernels (rpe_residuals.hpp: p = Rh x + th, p_lo = Rl x + tl as
packed fp32 FMAs on pairs of correspondences, residual (p_hi - Xc) + p_lo, fp64 accumulators) against the oracle's fp64 restatement.
The library uses it for fp32 launches of >= 400 000 correspondences; RPE_F32_TRANSFORM=1 (read at rpe_create) forces it at any size so
that the small cases below exercise the same code.  Gate of the round-3 review: converged pose within 1e-7 rad / 1e-7 relative t of
the fp64 loop at 1 M, and the |delta| < 1e-9 stop still reached at N >= 100 000."""
import os

import numpy as np
import pytest

from rgbd_pose_estimation_amd import _lib as L, api
import util

pytestmark = pytest.mark.gpu
KIND_ARR = {L.RES_P2P: ("Q", "P", None), L.RES_P2PLANE: ("Q", "P", "N"), L.RES_BEARING: ("Q", "U", None)}


def _ctx(env):
    """a context created under a temporary environment (RPE_F32_TRANSFORM / RPE_RESIDENT are read at creation)"""
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


@pytest.mark.parametrize("flags", [0, L.USE_MASK | L.USE_WEIGHT])
@pytest.mark.parametrize("n", [1, 6, 1000, 4099, 400001, 1000000])
@pytest.mark.parametrize("kind", [L.RES_P2P, L.RES_P2PLANE, L.RES_BEARING])
def test_normal_equations_with_the_fp32_transform(oracle, kind, n, flags):
    """Forced fp32 transform (every size, NaN-marked columns and clean waves, masks + weights) and the library's own choice side by side
    against the oracle's fp64 normal equations at the same pose."""
    sc = util.scene_full(70 + n, n, np.float32, n2d=2.0, n3d=0.03, nan_frac=0.02 if n >= 100 else 0.0)
    rng = np.random.default_rng(n)
    pose = api.pose12(*util.perturbed_pose(rng, sc.R, sc.t))
    mod = L.MOD_23 if kind == L.RES_BEARING else L.MOD_33
    mask = (rng.uniform(size=n) < 0.7).astype(np.int16) if flags else None
    w = rng.uniform(0.1, 2.0, n).astype(np.float32) if flags else None
    a, b, c = (getattr(sc, k) if k else None for k in KIND_ARR[kind])
    ref = oracle.gn_normal_eq(kind, a, b, c, mask=mask, weight=w, pose=pose)
    Ho, go, costo, cnto = util.unpack_ne(ref)
    do = oracle.gn_solve(ref)[0] if n >= 1000 else None
    for env in ({"RPE_F32_TRANSFORM": "1"}, {}, {"RPE_F32_TRANSFORM": "0"}):
        ctx = _ctx(env).load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nc=sc.N)
        try:
            if flags:
                ctx.upload_mask(mod, mask); ctx.upload_weight(mod, w)
            rec, used = ctx.normal_eq(kind, pose, flags=flags)
            assert np.array_equal(used, pose)
            H, g, cost, cnt = util.unpack_ne(rec)
            assert abs(cnt - cnto) <= 1e-6 * max(1.0, abs(cnto)), env
            assert np.max(np.abs(H - Ho)) <= 3e-6 * np.max(np.abs(Ho)), env
            # the residuals carry ~1e-6 m of rounding each in the fp32 form (3e-9 m in the fp64 form): the cost and the step see it
            # averaged over the correspondences
            assert abs(cost - costo) <= (2e-5 if n >= 1000 else 2e-4) * abs(costo) + 1e-12, env
            if do is not None:
                d = api.gn_solve(rec)
                assert np.linalg.norm(d - do) <= 5e-7 * max(1.0, np.linalg.norm(do)), (env, np.linalg.norm(d - do))
        finally:
            ctx.close()


@pytest.mark.parametrize("n", [100000, 1000000])
@pytest.mark.parametrize("kind", [L.RES_P2P, L.RES_P2PLANE, L.RES_BEARING])
def test_gate_converged_pose_and_the_1e9_stop(oracle, kind, n):
    """One launch per iteration (the streaming kernels) with the fp32 transform forced, against the oracle's fp64 Gauss-Newton on the
    same fp32 inputs: the |delta| < 1e-9 stop is reached (no stall at the fp32 resolution of the pose), in the oracle's number of
    iterations give or take one, and the converged pose is within 1e-7 rad / 1e-7 relative t of the oracle's."""
    sc = util.scene_full(90 + n // 1000, n, np.float32, n2d=2.0, n3d=0.03, outliers=0.0)
    p0 = api.pose12(*util.perturbed_pose(np.random.default_rng(n), sc.R, sc.t, 0.01, 0.03))
    a, b, c = (getattr(sc, k) if k else None for k in KIND_ARR[kind])
    term = dict(kind=kind, a=a, b=b)
    if c is not None:
        term["c"] = c
    po, ito, stepo, _ = oracle.gn_refine([term], n, p0, max_iter=30, tol=1e-9)
    assert ito < 30 and stepo < 1e-9
    ctx = _ctx({"RPE_F32_TRANSFORM": "1", "RPE_RESIDENT": "0"}).load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nc=sc.N)
    try:
        p, it, step, _ = ctx.gn_refine([kind], p0, max_iter=30, tol=1e-9)
        assert it < 30 and step < 1e-9, (it, step)
        assert abs(it - ito) <= 1, (it, ito)
        assert util.rot_err(p[:9].reshape(3, 3), po[:9].reshape(3, 3)) < 1e-7
        assert np.linalg.norm(p[9:] - po[9:]) / np.linalg.norm(po[9:]) < 1e-7
        # and it stays there: ten more iterations from the converged pose do not wander (the rounding inside the fp32 chain is a fixed
        # function of the hi part of the pose, which no longer changes)
        p2, it2, step2, _ = ctx.gn_refine([kind], p, max_iter=10, tol=0.0)
        assert step2 < 1e-9 and util.rot_err(p2[:9].reshape(3, 3), p[:9].reshape(3, 3)) < 1e-9
    finally:
        ctx.close()
