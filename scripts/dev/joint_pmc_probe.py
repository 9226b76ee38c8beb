"""The joint normal-equation kernels by themselves (for rocprofv3 --pmc passes on the SQ counters): p2p + bearing at RPE_PROBE_N
correspondences, one launch per call (30 launches), then -- frame-sized problems -- the resident loop (one launch, 200 iterations).
Prints one JSON line of event-timed averages."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from rgbd_pose_estimation_amd import _lib as L, api

n = int(os.environ.get("RPE_PROBE_N", "10000000"))
R, t, Q, P, N = bench.cheap_scene(n)
U = (Q.astype(np.float64) @ R.T + t)
U = (U / np.linalg.norm(U, axis=1, keepdims=True)).astype(np.float32)
pose = api.pose12(R, t)
ctx = api.Context(0).load(L.F32, xw=Q, xc=P, bv=U)
terms = [(L.RES_P2P, 1.0), (L.RES_BEARING, 1.0)]
out = {"n": n}
for _ in range(5):
    ctx.normal_eq_joint(terms, pose)
ctx.timing_enable(30, 1)
for _ in range(30):
    ctx.normal_eq_joint(terms, pose)
cnt, tot, mn = ctx.timing_collect()
out["one_launch"] = {"launches": cnt, "avg_launch_us": tot / cnt * 1e3, "min_us": mn * 1e3}
if n <= 400000:
    import time
    ctx.gn_refine_joint(terms, pose, max_iter=200, tol=0.0)
    t0 = time.perf_counter(); ctx.gn_refine_joint(terms, pose, max_iter=200, tol=0.0); out["resident_us_per_iter"] = (time.perf_counter() - t0) / 200 * 1e6
ctx.close()
print(json.dumps(out))
