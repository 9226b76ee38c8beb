"""Sweep workgroup size / grid cap of the reduction kernels (run once per env setting; development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S
def timeit(f, K=600, W=100):
    for _ in range(W): f()
    t0 = time.perf_counter()
    for _ in range(K): f()
    return (time.perf_counter() - t0) / K * 1e6
out = []
for n in [int(x) for x in sys.argv[1:]]:
    sc = S.dense_depth_scene(1, min(n, 1000000))
    reps = (n + len(sc.Q) - 1) // len(sc.Q)
    Q = np.ascontiguousarray(np.tile(sc.Q, (reps, 1))[:n]); P = np.ascontiguousarray(np.tile(sc.P, (reps, 1))[:n])
    ctx = api.Context(0).load(L.F32, xw=Q, xc=P)
    p = api.pose12(sc.R, sc.t); pp = p.copy()
    K = 600 if n < 5000000 else 100
    wall = timeit(lambda: ctx.gn_step(L.RES_P2P, pp), K)
    ctx.timing_enable(200, 1)
    for _ in range(200): ctx.normal_eq(L.RES_P2P, p)
    cnt, tot, mn = ctx.timing_collect()
    out.append("n=%d step %.1f us kern(ev) %.1f us %.0f GB/s" % (n, wall, tot / cnt * 1e3, 24 * n / (tot / cnt * 1e-3) / 1e9))
    ctx.close()
print(os.environ.get("RPE_BLOCK", "-"), os.environ.get("RPE_MAX_BLOCKS", "-"), " | ".join(out))
