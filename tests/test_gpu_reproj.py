"""-m gpu: RPE_RES_REPROJ, the 2D-3D PIXEL reprojection residual r = (p_x/p_z - bv_x/bv_z, p_y/p_z - bv_y/bv_z) (SURVEY.md Appendix B
row 4; pixel conversion /root/reference/TestMain.cpp:35-36, principal point at the origin, normalised image coordinates) against the
oracle's fp64 statement (oracle/orc_gn.hpp GN_REPROJ, itself checked against numerical Jacobians: tests/test_oracle_golden.py): the
one-launch kernel, the resident loop, the device-resident loop and the joint kernel, fp32 and fp64, masks / weights / robust weights,
NaN-marked bearings, and correspondences that are not in front of the camera."""
import numpy as np
import pytest

from rgbd_pose_estimation_amd import _lib as L, api
import util

pytestmark = pytest.mark.gpu
F = 585.0


def _scene(seed, n, dt, n2d=2.0, outliers=0.0, nan_frac=0.0, behind=True, on_plane=False):
    sc = util.scene_full(seed, n, np.float64, n2d=n2d, n3d=0.03, outliers=outliers)
    U = sc.U.copy()
    rng = np.random.default_rng(seed)
    if nan_frac > 0:
        U[rng.permutation(n)[: max(1, int(nan_frac * n))]] = np.nan           # bearings without a measurement
    Q = sc.Q.copy()
    if behind and n >= 20:
        # a few world points that land BEHIND the camera at the true pose, one exactly on the z = 0 plane, and bearings without a
        # forward component: they must contribute nothing and must not be counted
        idx = rng.permutation(n)[:6]
        # (on_plane: the middle one exactly ON the plane z = 0 -- for single evaluations only: a perturbed camera would see it in front)
        pc = np.array([[0.3, -0.2, -1.5], [0.1, 0.1, 0.0 if on_plane else -0.7], [1.0, 2.0, -0.2]])
        Q[idx[:3]] = (pc - sc.t) @ sc.R
        U[idx[3]] = [0.6, 0.8, 0.0]
        U[idx[4]] = [0.0, 0.6, -0.8]
    sc.Q, sc.U = Q, U
    return sc.astype(dt)


@pytest.mark.parametrize("f64", [False, True])
@pytest.mark.parametrize("flags", [0, L.USE_MASK | L.USE_WEIGHT])
@pytest.mark.parametrize("n", [1, 5, 1000, 4099, 307200])
def test_normal_equations_match_the_oracle(gpu_ctx_factory, oracle, n, flags, f64):
    dt = np.float64 if f64 else np.float32
    sc = _scene(120 + n, n, dt, nan_frac=0.03 if n >= 100 else 0.0, on_plane=True)
    rng = np.random.default_rng(n)
    pose = api.pose12(*util.perturbed_pose(rng, sc.R, sc.t))
    mask = (rng.uniform(size=n) < 0.7).astype(np.int16) if flags else None
    w = rng.uniform(0.1, 2.0, n).astype(dt) if flags else None
    ctx = gpu_ctx_factory().load(L.F64 if f64 else L.F32, xw=sc.Q, bv=sc.U)
    if flags:
        ctx.upload_mask(L.MOD_23, mask); ctx.upload_weight(L.MOD_23, w)
    rec, used = ctx.normal_eq(L.RES_REPROJ, pose, flags=flags)
    ref = oracle.gn_normal_eq(oracle.GN_REPROJ, sc.Q, sc.U, None, mask=mask, weight=w, pose=used, in_f64=f64)
    H, g, cost, cnt = util.unpack_ne(rec)
    Ho, go, costo, cnto = util.unpack_ne(ref)
    assert abs(cnt - cnto) <= 1e-6 * max(1.0, abs(cnto))          # the same correspondences counted (weights sum to rounding)
    tol = 1e-11 if f64 else 3e-6
    assert np.max(np.abs(H - Ho)) <= tol * np.max(np.abs(Ho))
    assert abs(cost - costo) <= (tol if n >= 1000 or f64 else 1e-4) * abs(costo) + 1e-12
    if n >= 1000:
        d, do = api.gn_solve(rec), oracle.gn_solve(ref)[0]
        assert np.linalg.norm(d - do) <= (1e-11 if f64 else 3e-7) * max(1.0, np.linalg.norm(do))


@pytest.mark.parametrize("f64", [False, True])
@pytest.mark.parametrize("n", [1000, 307200])
def test_refinement_matches_the_oracles_on_every_path(oracle, n, f64):
    """Resident loop, one launch per iteration and the device-resident loop against the oracle's fp64 Gauss-Newton, noisy pixels; the
    cost comes back in pixels^2 through the scale f^2 while the iterates do not depend on f."""
    import os
    dt = np.float64 if f64 else np.float32
    sc = _scene(140 + n // 1000, n, dt, n2d=1.5)
    p0 = api.pose12(*util.perturbed_pose(np.random.default_rng(n), sc.R, sc.t, 0.01, 0.03))
    po, ito, stepo, costo = oracle.gn_refine([dict(kind=oracle.GN_REPROJ, a=sc.Q, b=sc.U, scale=F * F)], n, p0, max_iter=30, tol=1e-9, in_f64=f64)
    assert 0 < ito < 30
    for env in ({}, {"RPE_RESIDENT": "0"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ctx = api.Context(0)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        try:
            ctx.load(L.F64 if f64 else L.F32, xw=sc.Q, bv=sc.U)
            p, it, step, cost = ctx.gn_refine([L.RES_REPROJ], p0, scales=[F * F], max_iter=30, tol=1e-9)
            assert it == ito and step < 1e-9
            assert util.rot_err(p[:9].reshape(3, 3), po[:9].reshape(3, 3)) < 1e-7 and util.trans_rel_err(p[9:], po[9:]) < 1e-7
            assert abs(cost - costo) <= 1e-5 * costo
            assert 0.5 < np.sqrt(cost / (2 * n)) < 3.0                     # RMS reprojection error per coordinate: the 1.5 px of the scene
            p1, it1, _, _ = ctx.gn_refine([L.RES_REPROJ], p0, max_iter=30, tol=1e-9)   # f = 1: the same iterates
            assert it1 == it and np.max(np.abs(p1 - p)) < 1e-12
            pd, itd, stepd, _ = ctx.gn_refine_device([(L.RES_REPROJ, 1.0)], p0, max_iter=30, tol=1e-9)
            assert itd == ito and np.max(np.abs(pd - p)) < 1e-9
        finally:
            ctx.close()
    # noise-free pixels: the minimiser is the true pose
    sc0 = _scene(7, 2000, dt, n2d=0.0, behind=False)   # (a point ON the z = 0 plane would swing in front of a perturbed camera)
    ctx = api.Context(0).load(L.F64 if f64 else L.F32, xw=sc0.Q, bv=sc0.U)
    p, it, _, _ = ctx.gn_refine([L.RES_REPROJ], api.pose12(*util.perturbed_pose(np.random.default_rng(1), sc0.R, sc0.t, 0.03, 0.05)), max_iter=30, tol=1e-9)
    ctx.close()
    assert util.rot_err(p[:9].reshape(3, 3), sc0.R) < (1e-9 if f64 else 2e-6) and np.linalg.norm(p[9:] - sc0.t) < (1e-8 if f64 else 2e-5)


@pytest.mark.parametrize("robust", [L.ROBUST_NONE, L.ROBUST_HUBER])
@pytest.mark.parametrize("n", [4099, 307200])
def test_joint_kernel_with_the_reprojection_term(gpu_ctx_factory, oracle, n, robust):
    """3D-3D + pixel reprojection (+ normals) in ONE fused pass, masks and weights, robust weights on |r| in normalised coordinates."""
    sc = _scene(160 + n % 7, n, np.float32, outliers=0.1 if robust else 0.0, nan_frac=0.02)
    rng = np.random.default_rng(n)
    pose = api.pose12(*util.perturbed_pose(rng, sc.R, sc.t, 0.01, 0.03))
    masks = [(rng.uniform(size=n) < 0.8).astype(np.int16) for _ in range(3)]
    weights = [rng.uniform(0.2, 1.5, n).astype(np.float32) for _ in range(3)]
    ctx = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    for m in range(3):
        ctx.upload_mask(m, masks[m]); ctx.upload_weight(m, weights[m])
    arr = {L.RES_P2P: ("Q", "P", None, L.MOD_33), L.RES_P2PLANE: ("Q", "P", "N", L.MOD_33), L.RES_REPROJ: ("Q", "U", None, L.MOD_23),
           L.RES_NORMAL: ("M", "N", None, L.MOD_NN)}
    for kinds in ((L.RES_REPROJ,), (L.RES_P2P, L.RES_REPROJ), (L.RES_P2PLANE, L.RES_REPROJ), (L.RES_REPROJ, L.RES_NORMAL),
                  (L.RES_P2P, L.RES_REPROJ, L.RES_NORMAL), (L.RES_P2PLANE, L.RES_REPROJ, L.RES_NORMAL)):
        terms = [(k, {L.RES_REPROJ: F * F / 1e4, L.RES_NORMAL: 0.5}.get(k, 1.0), robust, {L.RES_REPROJ: 0.004, L.RES_NORMAL: 0.05}.get(k, 0.08)) for k in kinds]
        rec = ctx.normal_eq_joint(terms, pose, flags=L.USE_MASK | L.USE_WEIGHT)
        tot = np.zeros(29)
        for kind, scale, rb, rk in terms:
            a, b, c, mod = arr[kind]
            r = oracle.gn_normal_eq(kind, getattr(sc, a), getattr(sc, b), None if c is None else getattr(sc, c), mask=masks[mod], weight=weights[mod],
                                    pose=pose, robust=rb, robust_k=rk)
            tot[:28] += scale * r[:28]; tot[28] += r[28]
        assert np.max(np.abs(rec[:29] - tot)) <= 1e-5 * np.max(np.abs(tot)), kinds
    # and a refinement with it: resident joint loop vs the oracle's
    terms = [(L.RES_P2P, 1.0, robust, 0.08), (L.RES_REPROJ, F * F / 1e4, robust, 0.004)]
    p, it, _, _ = ctx.gn_refine_joint(terms, pose, flags=L.USE_MASK, max_iter=40, tol=1e-8)
    po, ito, _, _ = oracle.gn_refine([dict(kind=L.RES_P2P, a=sc.Q, b=sc.P, mask=masks[1], scale=1.0, robust=robust, robust_k=0.08),
                                      dict(kind=oracle.GN_REPROJ, a=sc.Q, b=sc.U, mask=masks[0], scale=F * F / 1e4, robust=robust, robust_k=0.004)],
                                     n, pose, max_iter=40, tol=1e-8)
    assert 0 < it < 40 and abs(it - ito) <= 1
    assert util.rot_err(p[:9].reshape(3, 3), po[:9].reshape(3, 3)) < util.ROT_TOL_RAD and util.trans_rel_err(p[9:], po[9:]) < util.TRANS_REL_TOL


def test_reprojection_and_bearing_are_alternatives(gpu_ctx_factory):
    sc = _scene(3, 200, np.float32, behind=False)
    ctx = gpu_ctx_factory().load(L.F32, xw=sc.Q, bv=sc.U)
    with pytest.raises(L.RpeError) as e:
        ctx.normal_eq_joint([(L.RES_BEARING, 1.0), (L.RES_REPROJ, 1.0)], api.pose12(sc.R, sc.t))
    assert e.value.code == L.RPE_ERR_ARG and "alternatives" in str(e.value)
    # every correspondence behind the camera: nothing counts, the system is refused rather than solved
    far = _scene(4, 500, np.float32, behind=False)
    back = api.pose12(np.diag([1.0, -1.0, -1.0]) @ far.R, np.diag([1.0, -1.0, -1.0]) @ far.t)
    c2 = gpu_ctx_factory().load(L.F32, xw=far.Q, bv=far.U)
    rec, _ = c2.normal_eq(L.RES_REPROJ, back)
    assert rec[28] == 0 and np.all(rec[:28] == 0)
    with pytest.raises(L.RpeError):
        c2.gn_refine([L.RES_REPROJ], back, max_iter=5, tol=1e-9)


def test_adapter_level_gn_refine_reproj(oracle):
    """pose/GaussNewton.hpp gn_refine_reproj<Tp>(PnPPoseAdapter&) through rpe_run (ls = 9): the drop-in header level of the same path."""
    n = 20000
    sc = _scene(9, n, np.float32, n2d=1.0, behind=False)
    R0, t0 = util.perturbed_pose(np.random.default_rng(2), sc.R, sc.t, 0.01, 0.03)
    got = api.run(api.M_NONE, L.F32, xw=sc.Q, bv=sc.U, ls=api.LS_GN_REPROJ, pose_in=(R0, t0))
    # the adapter holds its pose in Tp: the refinement starts from the float-rounded pose and hands back a float-rounded one
    q = oracle.pose7_from_Rt(R0, t0, False)
    from scipy.spatial.transform import Rotation
    Rf = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()
    po, ito, _, _ = oracle.gn_refine([dict(kind=oracle.GN_REPROJ, a=sc.Q, b=sc.U)], n, api.pose12(Rf, q[4:]), max_iter=20, tol=1e-9)
    assert got["iters"] == ito
    assert util.rot_err(got["R"], po[:9].reshape(3, 3)) < 1e-6 and util.trans_rel_err(got["t"], po[9:]) < 1e-6
