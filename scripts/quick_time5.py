import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_pose_estimation_amd import _lib as L, api, simulator as S
L.lib(); print("devices", L.device_count())
mode = sys.argv[1]
def timeit(f, K=1000, W=100):
    for _ in range(W): f()
    t0 = time.perf_counter()
    for _ in range(K): f()
    return (time.perf_counter() - t0) / K * 1e6
if mode == "launch_first":
    c0 = api.Context(0); sc0 = S.dense_depth_scene(1, 1000); c0.load(L.F32, xw=sc0.Q, xc=sc0.P); c0.p2p_moments(); 
import torch
torch.cuda.set_device(0)
print("avail", torch.cuda.is_available())
sc = S.dense_depth_scene(1, 307200)
ctx = api.Context(0).load(L.F32, xw=sc.Q, xc=sc.P)
p = api.pose12(sc.R, sc.t); pp = p.copy()
print(mode, "gn_step %.1f us" % timeit(lambda: ctx.gn_step(L.RES_P2P, pp)))
torch.cuda.synchronize()
print(mode, "gn_step after sync %.1f us" % timeit(lambda: ctx.gn_step(L.RES_P2P, pp)))
ctx.inlier_mask(L.VOTE_33, api.pose7_from_Rt(sc.R, sc.t), thre_3d=0.2)
print(mode, "gn_step mask %.1f us" % timeit(lambda: ctx.gn_step(L.RES_P2P, pp, L.USE_MASK)))
ctx.timing_enable(1000, 8)
print(mode, "gn_step mask + ev/8 %.1f us" % timeit(lambda: ctx.gn_step(L.RES_P2P, pp, L.USE_MASK), 1000, 0)); print(ctx.timing_collect())
