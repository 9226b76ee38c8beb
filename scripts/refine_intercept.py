"""Fixed cost of one rpe_gn_refine call (resident loop): wall time of refine(max_iter=K) for several K; the slope is the per-step time, the
intercept the launch + first-iteration cost that a short refinement (the driver's 20-step bench) pays.  Development aid."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api
from tail_timeline import scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 307200
R, t, arrs = scene(n)
ctx = api.Context(0).load(L.F32, **arrs)
p = api.pose12(R, t)
ctx.gn_refine([0], p, max_iter=300, tol=0.0)
rows = []
for K in (2, 5, 10, 20, 50, 100, 400):
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        ctx.gn_refine([0], p, max_iter=K, tol=0.0)
        ts.append(time.perf_counter() - t0)
    rows.append((K, float(np.median(ts)) * 1e6))
Ks = np.array([r[0] for r in rows], float); T = np.array([r[1] for r in rows])
slope, icpt = np.polyfit(Ks, T, 1)
print(json.dumps(dict(n=n, us_per_call={str(k): round(v, 2) for k, v in rows}, slope_us_per_step=slope, intercept_us=icpt)))
