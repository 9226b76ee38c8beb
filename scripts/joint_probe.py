"""Fused joint normal-equation kernel: HIP-event-free wall time per call at 307200 / 10 M correspondences (development aid)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S
rng = np.random.default_rng(2)
R, t = S.random_pose(rng)
for n in (307200, 10000000):
    base = S.simulate_2d_3d_nl_correspondences(rng, R, t, min(n, 1000000), 1.0, 0.0, 0.02, 0.0, 0.03, 0.0).astype(np.float32)
    reps = (n + len(base.Q) - 1) // len(base.Q)
    tile = lambda a: np.ascontiguousarray(np.tile(a, (reps, 1))[:n])
    ctx = api.Context(0).load(L.F32, xw=tile(base.Q), xc=tile(base.P), bv=tile(base.U), nw=tile(base.M), nc=tile(base.N))
    p = api.pose12(R, t)
    for name, terms, bpc in (("p2p+bearing+normal", [(L.RES_P2P, 1.0), (L.RES_BEARING, 1.0), (L.RES_NORMAL, 1.0)], 60),
                             ("p2plane+bearing+normal", [(L.RES_P2PLANE, 1.0), (L.RES_BEARING, 1.0), (L.RES_NORMAL, 1.0)], 60),
                             ("p2plane+bearing", [(L.RES_P2PLANE, 1.0), (L.RES_BEARING, 1.0)], 48)):
        f = lambda: ctx.normal_eq_joint(terms, p)
        for _ in range(3): f()
        K = 40
        t0 = time.perf_counter()
        for _ in range(K): f()
        dt = (time.perf_counter() - t0) / K
        print(json.dumps(dict(n=n, terms=name, wall_us=dt * 1e6, wall_GBs=bpc * n / dt / 1e9)), flush=True)
    ctx.close()
