import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api
import config3_case, util
sc, U = config3_case.scene()
rng = np.random.default_rng(1)
p0 = api.pose12(*util.perturbed_pose(rng, sc.R, sc.t, 0.003, 0.01))
q0 = api.pose7_from_Rt(p0[:9].reshape(3, 3), p0[9:], L.F32)
thr = float(np.cos(np.arctan(config3_case.THRE_2D / config3_case.F)))
c = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P, bv=U)
c.inlier_mask(L.VOTE_33_23, q0, config3_case.THRE_3D, thr)
def slope(f, K=400):
    f(50); best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); f(K); t1 = time.perf_counter(); f(2 * K); t2 = time.perf_counter()
        best = min(best, ((t2 - t1) - (t1 - t0)) / K)
    return best * 1e6
terms = [(L.RES_P2P, 1.0), (L.RES_BEARING, 1.0)]
print(json.dumps({"config3_joint_device_loop_us_per_iteration": slope(lambda k: c.gn_refine_device(terms, p0, L.USE_MASK, k, 0.0)),
                  "config3_joint_host_loop_us_per_iteration": slope(lambda k: c.gn_refine_joint([(a, b, 0, 1.0) for a, b in terms], p0, flags=L.USE_MASK, max_iter=k, tol=0.0))}))
a = c.gn_refine_device(terms, p0, L.USE_MASK, 12, 0.0); b = c.gn_refine_joint([(x, y, 0, 1.0) for x, y in terms], p0, flags=L.USE_MASK, max_iter=12, tol=0.0)
print("pose diff", float(np.abs(a[0] - b[0]).max()), a[1], b[1])
