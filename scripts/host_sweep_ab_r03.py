import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import bench
from rgbd_pose_estimation_amd import _lib as L, api
os.sched_setaffinity(0, {16})
sc = bench.make_shard(0, 307200)
ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P)
R0, t0 = bench.initial_pose(sc)
ctx.inlier_mask(L.VOTE_33, api.pose7_from_Rt(R0, t0, L.F32), thre_3d=0.2)
p = api.pose12(R0, t0)
def f(k): ctx.gn_refine([L.RES_P2P], p, None, L.USE_MASK, k, 0.0)
f(2000)
best = []
for _ in range(9):
    a = time.perf_counter(); f(2000); b = time.perf_counter(); f(4000); c = time.perf_counter()
    best.append(((c - b) - (b - a)) / 2000 * 1e6)
best.sort()
print("sweep=%s" % os.environ.get("RPE_HOST_SWEEP", "0"), "us/step median %.3f min %.3f" % (best[len(best)//2], best[0]))
