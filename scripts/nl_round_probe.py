"""K5 (nl_round) wall time per call at the BASELINE sizes (development aid)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S
rng = np.random.default_rng(2)
R, t = S.random_pose(rng)
for n in (307200, 1000000, 10000000):
    base = S.simulate_2d_3d_nl_correspondences(rng, R, t, min(n, 1000000), 1.0, 0.0, 0.02, 0.0, 0.03, 0.0).astype(np.float32)
    reps = (n + len(base.Q) - 1) // len(base.Q)
    tile = lambda a: np.ascontiguousarray(np.tile(a, (reps, 1))[:n])
    ctx = api.Context(0).load(L.F32, xw=tile(base.Q), xc=tile(base.P), bv=tile(base.U), nw=tile(base.M), nc=tile(base.N))
    for weighted in (False, True):
        for mod in range(3):
            ctx.upload_weight(mod, np.ones(n, np.float32) if weighted else None)
        f = lambda: ctx.nl_round(np.zeros(3), np.zeros(3), np.zeros(3), np.eye(3))
        for _ in range(3): f()
        K = 30
        t0 = time.perf_counter()
        for _ in range(K): f()
        dt = (time.perf_counter() - t0) / K
        bpc = 66 + (12 if weighted else 0)
        print(json.dumps(dict(n=n, weighted=weighted, wall_us=dt * 1e6, wall_GBs=bpc * n / dt / 1e9)), flush=True)
    ctx.close()
