"""-m gpu: the degenerate inputs of the path -- no valid correspondence at all, nothing selected by the mask, thresholds that admit nothing or
everything, rank-deficient point sets, the longest hypothesis list -- through the C ABI, against the oracle where it defines a result and
against a loud error where the reference would have produced NaNs or aborted (AbsoluteOrientation.hpp:53 assert, SOPHUS_ENSURE)."""
import os

import numpy as np
import pytest

from rgbd_pose_estimation_amd import _lib as L, api
import util

pytestmark = pytest.mark.gpu

ORC_KIND = lambda o: {L.VOTE_33: o.V_33, L.VOTE_33_23: o.V_33_23, L.VOTE_NN_33: o.V_NN_33, L.VOTE_NN_33_23: o.V_NN_33_23, L.VOTE_23: o.V_23}  # noqa: E731


def _poses(sc, H, f64):
    rng = np.random.default_rng(5)
    out = []
    for h in range(H):
        w = rng.normal(size=3) * (0.0 if h == 0 else 0.02)
        th = np.linalg.norm(w)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        dR = np.eye(3) if th == 0 else np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
        out.append(api.pose7_from_Rt(dR @ sc.R, sc.t + (rng.normal(size=3) * 0.02 if h else 0), L.F64 if f64 else L.F32))
    return np.stack(out)


@pytest.mark.parametrize("n", [1, 7, 1000, 307200])
@pytest.mark.parametrize("f64", [False, True])
def test_no_valid_correspondence_at_all(gpu_ctx_factory, oracle, n, f64):
    """every Xc column is NaN (isValid false everywhere): the 3D and normal tests admit nothing, the 2D test -- which the reference runs
    without the validity check -- still votes; counts and masks equal the oracle's; the moments see zero correspondences."""
    dt = np.float64 if f64 else np.float32
    sc = util.scene_full(300 + n, n, dt, nan_frac=0.0)
    P = np.full_like(sc.P, np.nan)
    poses = _poses(sc, 5, f64)
    thr3, cthr, cnl = 0.2, oracle.cos_thr(f64, 8.0, 585.0), oracle.cos_nl(f64, 0.1)
    ctx = gpu_ctx_factory().load(L.F64 if f64 else L.F32, xw=sc.Q, xc=P, bv=sc.U, nw=sc.M, nc=sc.N)
    prob = oracle.Problem(f64, xw=sc.Q, xc=P, bv=sc.U, nw=sc.M, nc=sc.N)
    for kind in (L.VOTE_33, L.VOTE_33_23, L.VOTE_NN_33, L.VOTE_NN_33_23):
        v = ctx.score(kind, poses, thr3, cthr, cnl, mode=L.SCORE_EXACT)
        vo, mo = oracle.votes(prob, ORC_KIND(oracle)[kind], poses, thr3, cthr, cnl, mask_for=0)
        assert np.array_equal(v, vo)
        if kind in (L.VOTE_33, L.VOTE_NN_33):
            assert not v.any()
        assert ctx.inlier_mask(kind, poses[0], thr3, cthr, cnl, mode=L.SCORE_EXACT) == vo[0]
        assert not ctx.download_mask(L.MOD_33).any()
    m = ctx.p2p_moments(L.SKIP_INVALID)
    assert m[17] == 0 and not m[:17].any()
    ne, _ = ctx.normal_eq(L.RES_P2P, api.pose12(sc.R, sc.t), L.SKIP_INVALID)
    assert not ne[:29].any()
    with pytest.raises(L.RpeError):   # nothing to solve: refused, not a NaN pose
        ctx.gn_refine([L.RES_P2P], api.pose12(sc.R, sc.t), None, L.SKIP_INVALID, 5, 1e-9)


@pytest.mark.parametrize("n", [5, 4099, 307200])
def test_mask_selects_nothing(gpu_ctx_factory, n):
    sc = util.scene33(310 + n, n, np.float32)
    ctx = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P)
    ctx.upload_mask(L.MOD_33, np.zeros(n, np.int16))
    m = ctx.p2p_moments(L.USE_MASK)
    assert m[17] == 0 and not m[:17].any()
    with pytest.raises(L.RpeError):
        ctx.gn_refine([L.RES_P2P], api.pose12(sc.R, sc.t), None, L.USE_MASK, 5, 1e-9)
    ctx.upload_mask(L.MOD_33, np.ones(n, np.int16))   # the context stays usable after the refusal
    pose, its, *_ = ctx.gn_refine([L.RES_P2P], api.pose12(np.eye(3), np.zeros(3)), None, L.USE_MASK, 20, 1e-9)
    assert its >= 1 and np.all(np.isfinite(pose))


@pytest.mark.parametrize("f64", [False, True])
def test_thresholds_that_admit_nothing_or_everything(gpu_ctx_factory, oracle, f64):
    n = 4099
    dt = np.float64 if f64 else np.float32
    sc = util.scene_full(320, n, dt, nan_frac=0.1)
    poses = _poses(sc, 4, f64)
    valid = int((~np.isnan(sc.P).all(1)).sum())
    ctx = gpu_ctx_factory().load(L.F64 if f64 else L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    prob = oracle.Problem(f64, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    for thr3, cthr, cnl in ((0.0, 2.0, 2.0), (1e30, -2.0, -2.0), (np.inf, -np.inf, -np.inf)):
        for mode in (L.SCORE_EXACT, L.SCORE_FAST):
            v = ctx.score(L.VOTE_NN_33_23, poses, thr3, cthr, cnl, mode=mode)
            vo, _ = oracle.votes(prob, oracle.V_NN_33_23, poses, thr3, cthr, cnl, mask_for=0)
            assert np.array_equal(v, vo)
            assert np.all(v == (0 if thr3 == 0.0 else 2 * valid + n))


def test_rank_deficient_sets_are_refused_not_nan(gpu_ctx_factory):
    """all points identical / all on one line: J^T J is singular; the reference's closed form would hand back NaNs or abort in SOPHUS_ENSURE,
    the Gauss-Newton entry points return an error."""
    n = 1000
    one = np.tile(np.array([[0.3, -0.2, 2.0]], np.float32), (n, 1))
    line = (np.linspace(0, 1, n, dtype=np.float32)[:, None] * np.array([[1.0, 2.0, 0.5]], np.float32)) + np.float32(1.0)
    off = np.array([[0.01, -0.02, 0.005]], np.float32)
    # The cancelled pivots of these sets are the rounding noise of the fp32 products, whose sign depends on the order of the sums (the
    # resident kernel, the one-launch kernel and the device loop differ): the pivot floor sits above that noise -- 16 eps of the
    # product dtype x the diagonal (rpe::pivot_floor) -- so every path refuses both sets, whatever the order (round-3 review, item 6).
    paths = {
        "resident": (lambda ctx: ctx.gn_refine([L.RES_P2P], api.pose12(np.eye(3), np.zeros(3)), None, 0, 5, 1e-9), {}),
        "one launch per iteration": (lambda ctx: ctx.gn_refine([L.RES_P2P], api.pose12(np.eye(3), np.zeros(3)), None, 0, 5, 1e-9), {"RPE_RESIDENT": "0"}),
        "device loop": (lambda ctx: ctx.gn_refine_device([(L.RES_P2P, 1.0)], api.pose12(np.eye(3), np.zeros(3)), 0, 5, 1e-9), {}),
        "device loop, one launch per iteration": (lambda ctx: ctx.gn_refine_device([(L.RES_P2P, 1.0)], api.pose12(np.eye(3), np.zeros(3)), 0, 5, 1e-9),
                                                  {"RPE_DEVICE_LOOP_RESIDENT": "0"}),
    }
    for pts in (one, line):
        for name, (f, env) in paths.items():
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
                ctx.load(L.F32, xw=pts, xc=pts + off)
                with pytest.raises(L.RpeError) as e:
                    f(ctx)
                assert "positive definite" in str(e.value), (name, str(e.value))
                rec, _ = ctx.normal_eq(L.RES_P2P, api.pose12(np.eye(3), np.zeros(3)))
                with pytest.raises(L.RpeError):
                    api.gn_solve(rec)            # the record carries its floor (slot 29) for the context-free solve too
            finally:
                ctx.close()
    # fp64 arrays: the same sets, the floor of fp64 products (1e-12)
    for pts in (one, line):
        ctx = gpu_ctx_factory().load(L.F64, xw=pts.astype(np.float64), xc=(pts + off).astype(np.float64))
        with pytest.raises(L.RpeError):
            ctx.gn_refine([L.RES_P2P], api.pose12(np.eye(3), np.zeros(3)), None, 0, 5, 1e-9)
    # and a well-posed far-field problem is NOT refused: a 2 m object 40 m away (translation / rotation coupling (s / d)^2 = 2.5e-3)
    rng = np.random.default_rng(5)
    far = (rng.uniform(-1, 1, (n, 3)) + np.array([0.0, 0.0, 40.0])).astype(np.float32)
    ctx = gpu_ctx_factory().load(L.F32, xw=far, xc=far + off)
    p, it, step, _ = ctx.gn_refine([L.RES_P2P], api.pose12(np.eye(3), np.zeros(3)), None, 0, 10, 1e-9)
    assert np.isfinite(p).all() and np.abs(far.astype(np.float64) @ p[:9].reshape(3, 3).T + p[9:] - (far + off).astype(np.float64)).max() < 1e-4
    # one plane seen point-to-plane: three of the six directions are unobservable
    rng = np.random.default_rng(3)
    xy = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
    plane = np.concatenate([xy, np.full((n, 1), 2.0, np.float32)], 1)
    nrm = np.tile(np.array([[0, 0, 1.0]], np.float32), (n, 1))
    ctx = gpu_ctx_factory().load(L.F32, xw=plane, xc=plane + off, nc=nrm, nw=nrm)
    with pytest.raises(L.RpeError):
        ctx.gn_refine([L.RES_P2PLANE], api.pose12(np.eye(3), np.zeros(3)), None, 0, 5, 1e-9)


@pytest.mark.parametrize("H", [1, 2, 31, 32, 33, 8192])
def test_list_lengths_up_to_the_longest(gpu_ctx_factory, oracle, H):
    n = 1000
    sc = util.scene_full(330, n, np.float32, nan_frac=0.1)
    base = _poses(sc, 8, False)
    poses = base[np.arange(H) % 8]
    thr3, cthr, cnl = 0.2, oracle.cos_thr(False, 8.0, 585.0), oracle.cos_nl(False, 0.1)
    ctx = gpu_ctx_factory().load(L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    prob = oracle.Problem(False, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
    vo, _ = oracle.votes(prob, oracle.V_33_23, base, thr3, cthr, cnl, mask_for=0)
    v = ctx.score(L.VOTE_33_23, poses, thr3, cthr, cnl, mode=L.SCORE_EXACT)
    assert np.array_equal(v, vo[np.arange(H) % 8])


def test_empty_problem_is_an_error_and_an_empty_list_is_empty(gpu_ctx_factory):
    ctx = gpu_ctx_factory()
    with pytest.raises(L.RpeError):
        ctx.p2p_moments(0)                     # no problem set
    ctx.set_problem(0, L.F32)
    with pytest.raises(L.RpeError):
        ctx.p2p_moments(0)                     # a problem of zero correspondences
    sc = util.scene33(340, 10, np.float32)
    ctx.load(L.F32, xw=sc.Q, xc=sc.P)
    assert len(ctx.score(L.VOTE_33, np.zeros((0, 7)), 0.2)) == 0
    sc7 = api.pose7_from_Rt(sc.R, sc.t)
    assert ctx.score(L.VOTE_33, sc7[None], 0.2)[0] > 0   # and the counters are clean afterwards
