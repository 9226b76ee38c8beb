import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S
sc = S.dense_depth_scene(1, 307200)
ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P)
p0 = api.pose12(sc.R, sc.t)
for reset in (False, True):
    pp = p0.copy()
    out = []
    for chunk in range(16):
        if reset: pp = p0.copy(); pp[9] += 0.01
        t0 = time.perf_counter()
        for _ in range(500): ctx.gn_step(L.RES_P2P, pp)
        out.append((time.perf_counter() - t0) / 500 * 1e6)
    print("reset" if reset else "converged", " ".join("%.1f" % x for x in out))
# normal_eq only (no solve) over time
out = []
for chunk in range(16):
    t0 = time.perf_counter()
    for _ in range(500): ctx.normal_eq(L.RES_P2P, p0)
    out.append((time.perf_counter() - t0) / 500 * 1e6)
print("normal_eq only", " ".join("%.1f" % x for x in out))
