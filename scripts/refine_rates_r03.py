"""Per-iteration time of the refinement loops this round moved into ONE resident launch (development aid / profiles evidence):
bearing-only refinement, and the configs[2] joint refinement (307 200 3D-3D + 2 000 bearings, inlier masks) -- resident against one
launch per iteration (RPE_RESIDENT=0), same problem, same iterations (tol = 0)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api
import config3_case, util


def ctx_env(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return api.Context(0)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def per_iter(f, K=400, reps=5):
    f(50)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(K); t1 = time.perf_counter(); f(2 * K); t2 = time.perf_counter()
        best = min(best, ((t2 - t1) - (t1 - t0)) / K)   # slope: the launch / closing cost drops out
    return best * 1e6


sc, U = config3_case.scene()
rng = np.random.default_rng(1)
p0 = api.pose12(*util.perturbed_pose(rng, sc.R, sc.t, 0.003, 0.01))
q0 = api.pose7_from_Rt(p0[:9].reshape(3, 3), p0[9:], L.F32)
thr = float(np.cos(np.arctan(config3_case.THRE_2D / config3_case.F)))
rows = []
for name, env in (("resident", {}), ("launch_per_iteration", {"RPE_RESIDENT": "0"})):
    c = ctx_env(env).load(L.F32, xw=sc.Q, xc=sc.P, bv=U)
    c.inlier_mask(L.VOTE_33_23, q0, config3_case.THRE_3D, thr)
    terms = [(L.RES_P2P, 1.0, 0, 1.0), (L.RES_BEARING, 1.0, 0, 1.0)]
    joint = per_iter(lambda k: c.gn_refine_joint(terms, p0, flags=L.USE_MASK, max_iter=k, tol=0.0))
    p2p = per_iter(lambda k: c.gn_refine([L.RES_P2P], p0, flags=L.USE_MASK, max_iter=k, tol=0.0))
    c.close()
    b = ctx_env(env).load(L.F32, xw=sc.Q[:2000], bv=U[:2000])
    bear2k = per_iter(lambda k: b.gn_refine([L.RES_BEARING], p0, max_iter=k, tol=0.0))
    b.close()
    full = util.scene_full(5, 307200, np.float32, n2d=2.0, n3d=0.03)
    pf = api.pose12(*util.perturbed_pose(rng, full.R, full.t, 0.003, 0.01))
    b = ctx_env(env).load(L.F32, xw=full.Q, bv=full.U)
    bear307k = per_iter(lambda k: b.gn_refine([L.RES_BEARING], pf, max_iter=k, tol=0.0))
    b.close()
    rows.append(dict(mode=name, config3_joint_p2p_bearing_us_per_iteration=joint, p2p_masked_307200_us_per_iteration=p2p, bearing_2000_us_per_iteration=bear2k,
                     bearing_307200_us_per_iteration=bear307k))
    print(json.dumps(rows[-1]), flush=True)
