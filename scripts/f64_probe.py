import json, os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S
rng = np.random.default_rng(1)
R, t = S.random_pose(rng)
for n in (307200, 10000000):
    base = S.simulate_2d_3d_nl_correspondences(rng, R, t, min(n, 1000000), 1.0, 0.0, 0.02, 0.0, 0.03, 0.0)
    reps = (n + len(base.Q) - 1) // len(base.Q)
    tile = lambda a, dt: np.ascontiguousarray(np.tile(a, (reps, 1))[:n].astype(dt))
    for dt, code, bpc in ((np.float32, L.F32, 24), (np.float64, L.F64, 48)):
        ctx = api.Context(0).load(code, xw=tile(base.Q, dt), xc=tile(base.P, dt), nc=tile(base.N, dt))
        p = api.pose12(R, t)
        for kind, name, mult in ((L.RES_P2P, "p2p", 1.0), (L.RES_P2PLANE, "p2plane", 1.5)):
            for _ in range(5): ctx.normal_eq(kind, p)
            ctx.timing_enable(60, 1)
            for _ in range(60): ctx.normal_eq(kind, p)
            cnt, tot, mn = ctx.timing_collect()
            avg = tot / cnt * 1e-3
            print(json.dumps(dict(n=n, dtype=str(np.dtype(dt)), kind=name, kernel_us=avg * 1e6, GBs=bpc * mult * n / avg / 1e9)), flush=True)
        ctx.close()
