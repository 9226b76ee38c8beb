"""-m gpu: randomised shapes.  The parametrised tests pin a handful of sizes; here sizes are drawn around every boundary the kernels
have (1 .. a few groups, wave and workgroup multiples +- 1, one workgroup's worth of groups +- 1), with random NaN columns, masks
and weights, and every integer result must match the oracle exactly, every float record to the documented tolerance."""
import os

import numpy as np
import pytest

from rgbd_pose_estimation_amd import _lib as L, api
import util

pytestmark = pytest.mark.gpu
SEEDS = range(int(os.environ.get("RPE_FUZZ_SEEDS", "6")))   # a longer campaign: RPE_FUZZ_SEEDS=200 pytest tests/test_gpu_fuzz.py


def sizes(rng, count):
    edges = [1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 127, 129, 255, 256, 257, 511, 513, 1023, 1025, 2047, 2049, 4095, 4097, 8191, 8193]
    extra = [int(x) for x in rng.integers(10, 20000, count)]
    pick = list(rng.choice(edges, min(count, len(edges)), replace=False)) + extra
    return [int(p) for p in pick[:count]]


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_votes_and_masks_exact(gpu_ctx_factory, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    kinds = [L.VOTE_33, L.VOTE_23, L.VOTE_33_23, L.VOTE_NN_23, L.VOTE_NN_33, L.VOTE_NN_33_23, L.VOTE_23_MATRIX]
    omap = {L.VOTE_33: oracle.V_33, L.VOTE_23: oracle.V_23, L.VOTE_33_23: oracle.V_33_23, L.VOTE_NN_23: oracle.V_NN_23,
            L.VOTE_NN_33: oracle.V_NN_33, L.VOTE_NN_33_23: oracle.V_NN_33_23, L.VOTE_23_MATRIX: oracle.V_23_MATRIX}
    for n in sizes(rng, 7):
        f64 = bool(rng.integers(0, 2))
        dt = np.float64 if f64 else np.float32
        sc = util.scene_full(int(rng.integers(1, 10**6)), n, dt, nan_frac=float(rng.choice([0.0, 0.2])) if n > 3 else 0.0)
        H = int(rng.integers(1, 90))
        poses = np.array([oracle.pose7_from_Rt(*util.perturbed_pose(rng, sc.R, sc.t, ang=0.004 * (h % 5), dt=0.02 * (h % 4)), f64) for h in range(H)])
        thr3, cthr, cnl = 0.2, oracle.cos_thr(f64, 8.0, 585.0), oracle.cos_nl(f64, 0.1)
        ctx = gpu_ctx_factory().load(L.F64 if f64 else L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
        prob = oracle.Problem(f64, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
        kind = kinds[int(rng.integers(0, len(kinds)))]
        v = ctx.score(kind, poses, thr3, cthr, cnl, mode=L.SCORE_EXACT)
        vo, mo = oracle.votes(prob, omap[kind], poses, thr3, cthr, cnl, mask_for=H - 1)
        assert np.array_equal(v, vo), (n, f64, kind, H)
        assert ctx.inlier_mask(kind, poses[H - 1], thr3, cthr, cnl, mode=L.SCORE_EXACT) == vo[H - 1]
        for mod in (L.MOD_23, L.MOD_33, L.MOD_NN):
            try:
                got = ctx.download_mask(mod)
            except L.RpeError:
                continue                                  # modality not voted on by this kind: no device mask
            if mo[mod] is not None and len(mo[mod]) == n and (mo[mod].any() or got.any()):
                assert np.array_equal(got, mo[mod]) or not _kind_has(kind, mod), (n, kind, mod)
        ctx.close()


def _kind_has(kind, mod):
    has23 = kind in (L.VOTE_23, L.VOTE_33_23, L.VOTE_NN_23, L.VOTE_NN_33_23, L.VOTE_23_MATRIX)
    has33 = kind in (L.VOTE_33, L.VOTE_33_23, L.VOTE_NN_33, L.VOTE_NN_33_23)
    hasnn = kind in (L.VOTE_NN_23, L.VOTE_NN_33, L.VOTE_NN_33_23)
    return {L.MOD_23: has23, L.MOD_33: has33, L.MOD_NN: hasnn}[mod]


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_normal_equations_and_moments(gpu_ctx_factory, oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    for n in sizes(rng, 8):
        f64 = bool(rng.integers(0, 2))
        dt = np.float64 if f64 else np.float32
        sc = util.scene_full(int(rng.integers(1, 10**6)), n, dt, nan_frac=float(rng.choice([0.0, 0.15])) if n > 3 else 0.0)
        mask = (rng.random(n) < 0.7).astype(np.int16) if rng.integers(0, 2) else None
        weight = (rng.random(n) + 0.1).astype(dt) if rng.integers(0, 2) else None
        if mask is not None and not ((mask == 1) & ~np.isnan(np.asarray(sc.P)).all(1)).any():
            mask[:] = 1
        ctx = gpu_ctx_factory().load(L.F64 if f64 else L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
        flags = 0
        for mod in (L.MOD_23, L.MOD_33, L.MOD_NN):
            if mask is not None:
                ctx.upload_mask(mod, mask)
            if weight is not None:
                ctx.upload_weight(mod, weight)
        flags |= L.USE_MASK if mask is not None else 0
        flags |= L.USE_WEIGHT if weight is not None else 0
        p = api.pose12(*util.perturbed_pose(rng, sc.R, sc.t, ang=0.01, dt=0.03))
        for kind, b, c in ((L.RES_P2P, sc.P, None), (L.RES_P2PLANE, sc.P, sc.N), (L.RES_BEARING, sc.U, None)):
            got, _ = ctx.normal_eq(kind, p, flags)
            ref = oracle.gn_normal_eq(kind, sc.Q, b, c, mask=mask, weight=weight, pose=p, in_f64=f64)
            scale = np.abs(ref[:21]).max() + 1e-30
            tol = 1e-10 if f64 else 3e-6
            assert np.abs(got[:29] - ref).max() <= tol * max(scale, np.abs(ref).max()), (n, f64, kind, mask is not None, weight is not None)
            if weight is None:
                assert got[28] == ref[28]                 # number of contributing correspondences: exact
        m = ctx.p2p_moments(flags | L.SKIP_INVALID)
        valid = ~np.isnan(np.asarray(sc.P, np.float64)).all(1)
        sel = valid & ((mask == 1) if mask is not None else True)
        w = (np.asarray(weight, np.float64) if weight is not None else np.ones(n))[sel]
        Q, P = np.asarray(sc.Q, np.float64)[sel], np.asarray(sc.P, np.float64)[sel]
        ref = np.concatenate([[w.sum()], (w[:, None] * Q).sum(0), (w[:, None] * P).sum(0), ((w[:, None] * P).T @ Q).reshape(9), [(w * (P * P).sum(1)).sum()], [sel.sum()]])
        assert np.abs(m - ref).max() <= 1e-11 * (np.abs(ref).max() + 1) * max(1.0, np.sqrt(n)), (n, f64)
        assert m[17] == sel.sum()
        ctx.close()


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_whole_pipelines_equal(oracle, seed):
    """Whole RANSAC / PROSAC runs on random noisy scenes (random solver, size, dtype, noise, outlier share, missing camera points,
    thresholds, iteration cap, sampler seed): consensus size, adapted Iter, the three masks and the winning hypothesis equal the
    oracle's -- every integer exactly, the hypothesis bit for bit."""
    from test_gpu_pipelines import PIPELINES
    rng = np.random.default_rng(3000 + seed)
    for _ in range(3):
        name, method, arrays, ls, thr = PIPELINES[int(rng.integers(0, len(PIPELINES)))]
        f64 = bool(rng.integers(0, 2))
        dt = np.float64 if f64 else np.float32
        n = int(rng.choice([rng.integers(8, 64), rng.integers(64, 1000), rng.integers(1000, 6000)]))
        sc = util.scene_full(int(rng.integers(1, 10**6)), n, dt, n2d=float(rng.choice([0.5, 1.0, 3.0])), n3d=float(rng.choice([0.01, 0.05, 0.1])),
                             nnl_deg=float(rng.choice([1.0, 2.0, 5.0])), outliers=float(rng.choice([0.0, 0.1, 0.3])), nan_frac=float(rng.choice([0.0, 0.05, 0.2])))
        data = dict(xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
        sel = {k: data[k] for k in arrays}
        scale = float(rng.choice([0.5, 1.0, 2.0]))
        kw = dict(iters=int(rng.choice([20, 100, 300])), confidence=float(rng.choice([0.99, 0.9999])), seed=int(rng.integers(1, 1000)),
                  **{k: v * scale for k, v in thr.items()})
        got = api.run(getattr(api, method), L.F64 if f64 else L.F32, weights=sc.weights, ls=api.LS_NONE, score_mode=L.SCORE_EXACT, **sel, **kw)
        ref = oracle.run(oracle.Problem(f64, weights=sc.weights, **sel), getattr(oracle, method), ls=oracle.LS_NONE, **kw)
        ctx = (name, n, f64, kw)
        assert got["max_votes"] == ref["max_votes"], ctx
        assert got["iters"] == ref["iters"], ctx
        assert np.array_equal(got["masks"], ref["masks"]), ctx
        if ref["max_votes"] > 0:
            assert np.array_equal(got["R"], ref["R"]) and np.array_equal(got["t"], ref["t"]), ctx


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_resident_loops_equal_one_launch_per_iteration(seed):
    """The resident grids (host-driven and autonomous) at random sizes -- hence random grids, run lengths and ragged last groups --
    random residual kind(s), dtype, masks and weights, against the same refinement with one launch per iteration: same iteration
    counts, poses to the rounding of sums added in another order, and no lost grid."""
    from test_gpu_resident_paths import _ctx, _pose_close
    rng = np.random.default_rng(4000 + seed)
    res, per = _ctx(), _ctx({"RPE_RESIDENT": "0", "RPE_DEVICE_LOOP_RESIDENT": "0"})
    try:
        for _ in range(3):
            f64 = bool(rng.integers(0, 4) == 0)
            dt = np.float64 if f64 else np.float32
            n = int(rng.choice([rng.integers(40, 600), rng.integers(600, 40000), rng.integers(40000, 700000)]))
            sc = util.scene_full(int(rng.integers(1, 10**6)), n, dt, n2d=2.0, n3d=0.03, nan_frac=float(rng.choice([0.0, 0.03])))
            p0 = api.pose12(*util.perturbed_pose(rng, sc.R, sc.t, 0.01, 0.03))
            use_mask, use_weight = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            flags = (L.USE_MASK if use_mask else 0) | (L.USE_WEIGHT if use_weight else 0)
            mask = (rng.uniform(size=n) < 0.8).astype(np.int16)
            weight = rng.uniform(0.2, 2.0, n).astype(dt)
            single = int(rng.choice([L.RES_P2P, L.RES_P2PLANE, L.RES_BEARING]))
            combo = [(L.RES_P2P, 1.0, 0, 1.0), (L.RES_BEARING, 4.0, int(rng.choice([0, L.ROBUST_HUBER])), 0.01)] + ([(L.RES_NORMAL, 0.5, 0, 1.0)] if rng.integers(0, 2) else [])
            iters = int(rng.integers(2, 15))
            out = []
            for c in (res, per):
                c.load(L.F64 if f64 else L.F32, xw=sc.Q, xc=sc.P, bv=sc.U, nw=sc.M, nc=sc.N)
                for m in (L.MOD_23, L.MOD_33, L.MOD_NN):
                    if use_mask:
                        c.upload_mask(m, mask)
                    if use_weight:
                        c.upload_weight(m, weight)
                out.append((c.gn_refine([single], p0, None, flags, iters, 0.0), c.gn_refine_joint(combo, p0, flags=flags, max_iter=iters, tol=0.0),
                            c.gn_refine_device([(single, 1.0)], p0, flags, iters, 0.0), c.gn_refine_device([t[:2] for t in combo], p0, flags, iters, 0.0)))
            what = (n, f64, single, len(combo), use_mask, use_weight, iters)
            # fp32 products added in another order, amplified by the conditioning of small or heavily masked problems: 5e-7 here (the
            # fixed-size tests hold 1e-8), against the 1e-5 rad / 1e-4 of BASELINE.json
            for a, b in zip(*out):
                assert a[1] == b[1] == iters, what
                try:
                    _pose_close(a[0], b[0], 5e-7, 5e-7)
                except AssertionError as e:
                    raise AssertionError(f"{what}: {e}")
        assert res.resident_state()["lost"] == 0 and res.resident_state()["enabled"]
    finally:
        res.close(); per.close()
