import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S
rng = np.random.default_rng(2)
R, t = S.random_pose(rng)
n = 10000000
base = S.simulate_2d_3d_nl_correspondences(rng, R, t, 1000000, 1.0, 0.0, 0.02, 0.0, 0.03, 0.0).astype(np.float32)
tile = lambda a: np.ascontiguousarray(np.tile(a, (10, 1))[:n])
ctx = api.Context(0).load(L.F32, xw=tile(base.Q), xc=tile(base.P), bv=tile(base.U), nw=tile(base.M), nc=tile(base.N))
p = api.pose12(R, t)
for name, terms, bpc in (("p2p (joint kernel)", [(L.RES_P2P, 2.0)], 24), ("p2plane (joint)", [(L.RES_P2PLANE, 2.0)], 36), ("bearing (joint)", [(L.RES_BEARING, 2.0)], 24),
                         ("normal (joint)", [(L.RES_NORMAL, 1.0)], 24), ("p2p+bearing", [(L.RES_P2P, 1.0), (L.RES_BEARING, 1.0)], 36),
                         ("p2p+bearing+normal", [(L.RES_P2P, 1.0), (L.RES_BEARING, 1.0), (L.RES_NORMAL, 1.0)], 60)):
    f = lambda: ctx.normal_eq_joint(terms, p)
    for _ in range(3): f()
    K = 30
    t0 = time.perf_counter()
    for _ in range(K): f()
    dt = (time.perf_counter() - t0) / K
    print(json.dumps(dict(terms=name, wall_us=round(dt * 1e6, 1), wall_GBs=round(bpc * n / dt / 1e9))), flush=True)
for kind, name, bpc in ((L.RES_P2P, "p2p (dedicated K1)", 24), (L.RES_BEARING, "bearing (dedicated K3)", 24)):
    f = lambda: ctx.normal_eq(kind, p)
    for _ in range(3): f()
    t0 = time.perf_counter()
    for _ in range(30): f()
    dt = (time.perf_counter() - t0) / 30
    print(json.dumps(dict(terms=name, wall_us=round(dt * 1e6, 1), wall_GBs=round(bpc * n / dt / 1e9))), flush=True)
